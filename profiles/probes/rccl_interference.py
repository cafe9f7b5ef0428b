"""How much does a collective's reduction kernel cost the backward pass it overlaps?  (RCCL interference priced on ONE GPU.)

A one-GPU box cannot run an N > 1 RCCL ring, but what the ring costs the compute streams is local: RCCL's all-reduce is a kernel of a few
"channels" (workgroups of 256-512 threads: 16-64 of them on MI300-class parts) that streams the bucket through the CUs it occupies —
per rank of an 8-rank ring, each bucket element is read ~2(N-1)/N times from local memory, reduced, written and sent.  This probe launches a
stand-in with exactly that shape — `reduce_kernel`: G workgroups x 512 threads, grid-stride `a[i] += b[i]` over the bucket's range of the
flat gradient buffer, PASSES times — from the product's own bucket hooks (myconvnet_amd.dist.GradientReducer: same bucket plan, same
side-stream ordering as the RCCL calls it replaces) and reports the step time against the same step without it.

    python profiles/probes/rccl_interference.py [fp32|bf16] [steps]

Output: one line per configuration (G workgroups, passes), ms per step over `steps` steps after warm-up, and the difference to the
no-collective step.  MCN_PERS_CUS=<n> (read by the library) can be set to leave CUs free in the persistent forward grids.
"""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SRC = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(512) void reduce_kernel(float* __restrict__ a, const float* __restrict__ b, long n4, int passes) {
    float4* a4 = reinterpret_cast<float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (int p = 0; p < passes; ++p)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            float4 x = a4[i];
            const float4 y = b4[i];
            x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
            a4[i] = x;
        }
}
extern "C" int launch_reduce(float* a, const float* b, long n, int groups, int passes, void* stream) {
    hipLaunchKernelGGL(reduce_kernel, dim3(groups), dim3(512), 0, (hipStream_t)stream, a, b, n / 4, passes);
    return (int)hipGetLastError();
}
'''


def build_probe():
    d = tempfile.mkdtemp(prefix='mcn_probe_')
    src, so = os.path.join(d, 'probe.hip'), os.path.join(d, 'libprobe.so')
    open(src, 'w').write(SRC)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-o', so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.launch_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.launch_reduce.restype = ctypes.c_int
    return lib


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import myconvnet_amd as M            # (torch first, then libmcn_hip.so: one HIP runtime)
    from myconvnet_amd.dist import GradientReducer
    probe = build_probe()
    B = 256
    model = M.ResNet50([224, 224, 3], 1000, batch_size=B, num_gpus=1, half_precision=(dtype != 'fp32'), seed=0, device='cuda:0')
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, momentum=0.9, steps_per_epoch=5000, num_epochs=90)
    rng = np.random.default_rng(1234)
    model.feed(rng.random((B, 224, 224, 3), dtype=np.float32), rng.integers(0, 1000, B).astype(np.float32))
    st = model.store
    marks = model._train_low.bwd.marks
    ready = {}
    for label, idx in marks.items():
        if isinstance(label, tuple) and label[0] == 'grad_ready':
            for name in label[1]:
                ready[name] = idx
    variables = [(v.name, v.offset, (v.size + 3) // 4 * 4) for v in st.variables if v.trainable]
    side = model._train_low.bwd.side_stream
    red = GradientReducer(st.grad, variables, ready, 25.0, side_stream=side)
    recv = torch.zeros_like(st.grad)                       # stands in for the neighbour's chunk arriving over xGMI
    comm = side if side is not None else torch.cuda.current_stream()

    def hooks(groups, passes):
        def make(spans):
            def fire():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                comm.wait_event(ev)
                for s, e in spans:
                    rc = probe.launch_reduce(st.grad.data_ptr() + 4 * s, recv.data_ptr() + 4 * s, e - s, groups, passes, comm.cuda_stream)
                    assert rc == 0
            return fire
        return {idx: make(spans) for idx, spans in red._hooks.items()}

    def step(h):
        sp = model.stream_ptr()
        opt._set_hyper()
        opt._pre.run(sp)
        model.forward(train=True)
        model.backward(h)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        opt.optimization_operation.run(sp)

    def timed(h):
        for _ in range(5):
            step(h)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(h)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    nbytes = sum(v[2] for v in variables) * 4
    print('ResNet-50 %s B=%d: %d buckets of ~25 MB over %.1f MB of gradients; MCN_PERS_CUS=%s' % (dtype, B, len(red.plan), nbytes / 1e6, os.environ.get('MCN_PERS_CUS', '-')), flush=True)
    base = [timed(None) for _ in range(2)]
    print('no collective           : %.3f / %.3f ms per step' % tuple(base), flush=True)
    b0 = min(base)
    for groups in (16, 32, 64):
        for passes in (2, 4):
            t = timed(hooks(groups, passes))
            # bytes the stand-in moves per step: 3 accesses x 4 B x passes per gradient element
            print('reduce %2d WGs x %d passes: %.3f ms per step  (+%.3f ms, %+.1f %%; stand-in traffic %.2f GB / step)' % (groups, passes, t, t - b0, (t / b0 - 1) * 100, 12.0 * passes * nbytes / 4 / 1e9), flush=True)
    print('no collective (again)   : %.3f ms per step' % timed(None), flush=True)


if __name__ == '__main__':
    main()
