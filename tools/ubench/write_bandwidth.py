import torch, time
for mb in (2, 8, 33, 128, 512):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda").normal_()
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: x.copy_(y))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)[25] * 1e3
        print(f"{mb} MB {name}: {t:.1f} us -> {mb*1.048576/t*1e3 * (2 if name=='copy' else 1):.0f} GB/s (r+w)" )
