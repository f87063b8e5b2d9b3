import torch
for mb in (64, 189, 512, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.randn(n, device="cuda")
    for name, fn in (("fill", lambda: a.fill_(1.0)), ("copy", lambda: a.copy_(b)), ("read(sum)", lambda: b.sum())):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        print(f"{mb} MB {name}: {us:.1f} us = {mb * 1.048576 / us:.2f} TB/s per direction")
