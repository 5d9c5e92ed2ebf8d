import torch, numpy as np
x = torch.randn(1024, 50257, device="cuda")
for f, name in ((lambda: x.t().contiguous(), "x.t().contiguous()"), (lambda: x.clone(), "clone"), (lambda: torch.exp(x), "exp")):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    t = float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))
    print(f"{name}: {t:.1f} us  {2 * x.numel() * 4 / t / 1e6:.2f} TB/s")
