"""world_size-2 gloo tests (CPU) of the data-parallel exchange: bucketed all-reduce of the flat gradient buffer with
hooks fired from the backward launch list, BN running-statistic chain after all-gather, mean loss."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from myconvnet_amd import dist as D
from myconvnet_amd.graph import Program
from oracle import ops as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_process_group('cpu')
    assert dist.get_backend() == 'gloo'
    try:
        n_var, size = 6, 1000
        variables = [('v%d' % i, i * size, size) for i in range(n_var)]
        flat = torch.zeros(n_var * size)
        rng = np.random.default_rng(100 + rank)
        grads = [torch.from_numpy(rng.standard_normal(size).astype(np.float32)) for _ in range(n_var)]
        # a fake backward program: call i "computes" the gradient of variable n_var-1-i
        prog = Program()
        calls = []

        def make(i):
            def fn(*a):
                v = n_var - 1 - i
                flat[v * size:(v + 1) * size] = grads[v]
                calls.append(v)
                return 0
            return fn
        ready = {}
        for i in range(n_var):
            prog.add(make(i))
            ready['v%d' % (n_var - 1 - i)] = i + 1
        red = D.GradientReducer(flat, variables, ready, bucket_mb=2.5 * size * 4 / 1024 / 1024)
        assert red.covered_elements() == n_var * size
        prog.run(0, red.hooks())
        red.finish()
        # BN running-stat chain on gathered batch statistics
        bstats = torch.from_numpy(rng.standard_normal(50).astype(np.float32))
        gathered = torch.zeros(world, 50)
        dist.all_gather_into_tensor(gathered.view(-1), bstats)
        q.put((rank, flat.numpy().copy(), [g.numpy() for g in grads], gathered.numpy().copy(), bstats.numpy().copy(), len(red.plan)))
    finally:
        dist.destroy_process_group()


def _run_world(world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_bucketed_allreduce_and_bn_chain_world2():
    res = _run_world(2)
    total = sum(np.concatenate(r[2]) for r in res)
    for r in res:
        np.testing.assert_allclose(r[1], total, rtol=1e-6)              # every rank holds the SUM (1/N is applied by the optimizer kernel)
        assert r[5] >= 2                                                # more than one bucket => overlap points exist
    np.testing.assert_array_equal(res[0][3], res[1][3])                 # identical gathered statistics on every rank
    np.testing.assert_array_equal(res[0][3][1], res[1][4])
    run = np.zeros(50)
    mu, _ = O.bn_running_update_chain(run, run, [res[0][4], res[1][4]], [res[0][4], res[1][4]], 0.99)
    manual = 0.99 * (0.99 * run + 0.01 * res[0][4]) + 0.01 * res[1][4]
    np.testing.assert_allclose(mu, manual)


def test_bucketed_allreduce_and_bn_chain_world8():
    """The N = 8 leg of BASELINE configs[2] / [4] on CPU (gloo): eight ranks, every rank ends with the sum of the eight gradients in
    every bucket, identical gathered statistics in rank order, and the reference's chained running update over eight towers
    (convnet.py:1899-1909) — the arithmetic mcn_bn_running_chain implements."""
    world = 8
    res = _run_world(world)
    total = sum(np.concatenate(r[2]) for r in res)
    for r in res:
        np.testing.assert_allclose(r[1], total, rtol=1e-5, atol=1e-6)
        assert r[5] >= 2
        np.testing.assert_array_equal(r[3], res[0][3])
    for k in range(world):
        np.testing.assert_array_equal(res[0][3][k], res[k][4])          # row k of the gathered buffer is rank k's statistics
    run = np.zeros(50)
    stats = [r[4] for r in res]
    mu, _ = O.bn_running_update_chain(run, run, stats, stats, 0.99)
    manual = run
    for k in range(world):
        manual = 0.99 * manual + 0.01 * stats[k]
    np.testing.assert_allclose(mu, manual)
