import torch, time
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda'); c = torch.empty(n, device='cuda')
    a.normal_(); b.normal_()
    w = t(lambda: a.fill_(1.0)); print('%5d MB  fill  (write only)  %.0f GB/s' % (mb, n * 4 / w / 1e9))
    r = t(lambda: a.sum()); print('%5d MB  sum   (read only)   %.0f GB/s' % (mb, n * 4 / r / 1e9))
    cp = t(lambda: b.copy_(a)); print('%5d MB  copy  (1r + 1w)     %.0f GB/s total' % (mb, 2 * n * 4 / cp / 1e9))
    ad = t(lambda: torch.add(a, b, out=c)); print('%5d MB  add   (2r + 1w)     %.0f GB/s total' % (mb, 3 * n * 4 / ad / 1e9))
    del a, b, c
