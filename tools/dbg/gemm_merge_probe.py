"""Would one GEMM for gate_proj + up_proj (a derived [2 I, d] weight) beat two?  Llama-3.2-1B and Llama-3-8B shapes, bf16,
M = 512 rows (the one-token forward over KV rows) and M = 6000 (the re-encoding forward)."""
import torch

dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for d, I in ((2048, 8192), (4096, 14336)):
    for M in (512, 6000):
        x = torch.randn(M, d, device=dev, dtype=torch.bfloat16)
        wg = torch.randn(I, d, device=dev, dtype=torch.bfloat16)
        wu = torch.randn(I, d, device=dev, dtype=torch.bfloat16)
        wgu = torch.cat([wg, wu])
        two = t(lambda: (torch.nn.functional.linear(x, wg), torch.nn.functional.linear(x, wu)))
        one = t(lambda: torch.nn.functional.linear(x, wgu))
        act2 = t(lambda: torch.nn.functional.silu(torch.nn.functional.linear(x, wg)) * torch.nn.functional.linear(x, wu))
        def merged():
            h = torch.nn.functional.linear(x, wgu)
            return torch.nn.functional.silu(h[:, :I]) * h[:, I:]
        act1 = t(merged)
        print(f"d {d} I {I} M {M}: two GEMMs {two:.1f} us, one {one:.1f} us; with silu*up: {act2:.1f} vs {act1:.1f} us")
