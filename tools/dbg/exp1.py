import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
eng = HipEngine("cuda:0")
dev = eng.device
def trial(B, V, dtype, nbuf, iters=12, rng_mode=1):
    g = torch.Generator(device=dev); g.manual_seed(0)
    bufs = [(torch.randn((B, V), device=dev, generator=g) * 3).to(dtype) for _ in range(nbuf)]
    maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1/3, float("-inf"), 0.0)
    bits, _ = eng.mask_to_bits(maskf)
    pm = eng.prepare_masks(bits, V, dtype)
    mid = (torch.arange(B, device=dev) % 2).to(torch.int32)
    out = (torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
    ts = []
    for i in range(iters + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.step(bufs[i % nbuf], rng_mode=rng_mode, seed=1, offset=i, out=out, mask=pm, row_mask_id=mid)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"B={B} V={V} {dtype} nbuf={nbuf} rng={rng_mode}: ptrs={[hex(x.data_ptr()) for x in bufs]}", " ".join(f"{t:.0f}" for t in ts[3:]), flush=True)
    del bufs
f32, bf16 = torch.float32, torch.bfloat16
trial(2048, 50257, f32, 1)
trial(2048, 50257, f32, 2)
trial(2048, 50257, f32, 2, rng_mode=0)
trial(1536, 50257, f32, 2)
trial(1280, 50257, f32, 3)
trial(1024, 50257, f32, 4)
trial(4096, 32000, bf16, 2)
# mask prepare alone
V = 50257
g = torch.Generator(device=dev); g.manual_seed(0)
maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1/3, float("-inf"), 0.0)
bits, _ = eng.mask_to_bits(maskf)
for i in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pm = eng.prepare_masks(bits, V, f32); b.record(); torch.cuda.synchronize()
    print("mask_prepare us", a.elapsed_time(b) * 1e3)
