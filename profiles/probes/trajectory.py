"""Training-trajectory check of the Winograd kernels: the same ResNet-50 fp32 run (same initial weights, same four synthetic batches cycled)
with MCN_WINOGRAD=1 and MCN_WINOGRAD=0 — the loss sequences must stay together (both are correct fp32 evaluations of the same step; they
differ by rounding only).  Usage: python profiles/probes/trajectory.py [steps] > profiles/round3_winograd_trajectory.txt"""
import json
import os
import subprocess
import sys

import numpy as np

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
B = 64


def child():
    import torch
    import myconvnet_amd as M
    model = M.ResNet50([224, 224, 3], 1000, batch_size=B, num_gpus=1, half_precision=False, seed=0, device='cuda:0')
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.05, momentum=0.9, steps_per_epoch=5000, num_epochs=90, learning_warmup_epochs=0.0)
    rng = np.random.default_rng(99)
    data = [(rng.random((B, 224, 224, 3), dtype=np.float32), rng.integers(0, 1000, B).astype(np.float32)) for _ in range(4)]
    losses = []
    for s in range(STEPS):
        x, y = data[s % 4]
        model.feed(x, y)
        opt._update_learning_rate()
        loss, _, _ = opt._step(None)
        opt.curr_step += 1
        losses.append(float(loss))
    w = np.concatenate([np.asarray(v, np.float64).ravel() for k, v in sorted(model.get_variables('data').items())]) if hasattr(model, 'get_variables') else np.zeros(1)
    torch.cuda.synchronize()
    print(json.dumps({'losses': losses, 'wnorm': float(np.linalg.norm(w)), 'whead': w[:200000:997].tolist()}))


if __name__ == '__main__':
    if os.environ.get('TRAJ_CHILD') == '1':
        child()
        sys.exit(0)
    sys.path.insert(0, os.getcwd())
    runs = {}
    for mode in ('1', '0'):
        env = dict(os.environ, MCN_WINOGRAD=mode, TRAJ_CHILD='1', PYTHONPATH=os.getcwd())
        out = subprocess.run([sys.executable, os.path.abspath(__file__), str(STEPS)], env=env, capture_output=True, text=True, check=True).stdout
        runs[mode] = json.loads(out.strip().splitlines()[-1])
    a, b = np.array(runs['1']['losses']), np.array(runs['0']['losses'])
    print('# ResNet-v1.5-50 fp32, B = %d, 224x224, four synthetic batches cycled, Nesterov momentum 0.9, lr 0.05 (no warm-up), %d steps, same seed' % (B, STEPS))
    print('# step | loss with the Winograd kernels (MCN_WINOGRAD=1) | loss with the direct kernels (MCN_WINOGRAD=0) | relative difference')
    for s in range(STEPS):
        print('%3d | %.6f | %.6f | %.2e' % (s, a[s], b[s], abs(a[s] - b[s]) / abs(b[s])))
    wa, wb = np.array(runs['1']['whead']), np.array(runs['0']['whead'])
    print('# largest relative loss difference over the run: %.2e; parameters after the run (every 997th of the first 200 000): rel-L2 difference %.2e'
          % (float(np.max(np.abs(a - b) / np.abs(b))), float(np.linalg.norm(wa - wb) / max(np.linalg.norm(wb), 1e-30))))
