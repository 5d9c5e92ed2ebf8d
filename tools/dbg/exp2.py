"""Why is chunk_stats_kernel 42.5 us inside the SIS step but 38.6 us alone?  Fused step timed (HIP events) right behind
an lm_head-sized fp32 GEMM that (a) writes the very logits the step reads, (b) writes another buffer, (c) no GEMM."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
eng = HipEngine("cuda:0"); dev = eng.device
B, V, D = 1024, 50257, 768
g = torch.Generator(device=dev); g.manual_seed(0)
h = torch.randn((B, D), device=dev, generator=g)
W = torch.randn((V, D), device=dev, generator=g) * 0.1
bufs = [torch.randn((B, V), device=dev, generator=g) * 3 for _ in range(4)]
other = torch.empty((B, V), device=dev)
maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1/3, float("-inf"), 0.0)
bits, _ = eng.mask_to_bits(maskf); pm = eng.prepare_masks(bits, V)
mid = (torch.arange(B, device=dev) % 2).to(torch.int32)
out = (torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
def run(mode, iters=40):
    ts = []
    for i in range(iters + 5):
        x = bufs[i % 4]
        if mode == "same": torch.mm(h, W.t(), out=x)
        elif mode == "other": torch.mm(h, W.t(), out=other)
        elif mode == "chain":  # a few GEMMs like a forward
            for _ in range(6): torch.mm(h, W.t(), out=other)
            torch.mm(h, W.t(), out=x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.step(x, rng_mode=1, seed=1, offset=i, out=out, mask=pm, row_mask_id=mid); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = np.array(ts[5:]); print(f"{mode:6s}: median {np.median(ts):.1f} us  mean {ts.mean():.1f}", flush=True)
for m in ("none", "same", "other", "chain", "none"):
    run(m)
