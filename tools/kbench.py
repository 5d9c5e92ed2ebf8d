"""Kernel micro-benchmark for tuning (not the judged bench): fused step on rotating logits buffers."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd  # noqa: E402
from genlm_backend_amd import _lib  # noqa: E402

if os.environ.get("GLB_DBG_LIB"):  # diagnostic build of the library (make -C genlm-backend_amd/csrc dbg)
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", os.environ["GLB_DBG_LIB"])
from genlm_backend_amd.engine import HipEngine  # noqa: E402


def run(eng, B, V, dtype, mask_kind, rng_mode, nbuf, iters, ld=None, pforbid=1 / 3):
    dev = eng.device
    ld = ld or V
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    bufs = [(torch.randn((B, ld), device=dev, generator=g) * 3).to(dtype) for _ in range(nbuf)]
    K = 2
    maskf = torch.where(torch.rand((K, V), device=dev, generator=g) < pforbid, float("-inf"), 0.0)
    bits, _ = eng.mask_to_bits(maskf)
    mid = (torch.arange(B, device=dev) % K).to(torch.int32)
    kw = {}
    if mask_kind == 1:
        kw = dict(mask_kind=1, mask=bits, mask_id=mid)
    elif mask_kind == 3:  # prepared masks, ids per row
        kw = dict(mask=eng.prepare_masks(bits, V, dtype), row_mask_id=mid)
    elif mask_kind == 2:
        kw = dict(mask_kind=2, mask=maskf.contiguous(), mask_id=mid)
    noise = None
    if rng_mode == 2:
        noise = torch.empty((B, V), device=dev).exponential_(generator=g)
        kw["noise"] = noise
    out = (torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
    for i in range(5):
        eng.step(bufs[i % nbuf][:, :V] if ld != V else bufs[i % nbuf], vocab=V, rng_mode=rng_mode, seed=1, offset=i, out=out, **kw)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters):
        evs[i][0].record()
        eng.step(bufs[i % nbuf][:, :V] if ld != V else bufs[i % nbuf], vocab=V, rng_mode=rng_mode, seed=1, offset=i, out=out, **kw)
        evs[i][1].record()
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) * 1e3 for a, b in evs])
    es = bufs[0].element_size()
    byt = B * V * es + B * 8 + (K * ((V + 31) // 32) * 4 if mask_kind in (1, 3) else 0) + (K * V * 4 if mask_kind == 2 else 0) + (B * V * 4 if rng_mode == 2 else 0)
    med = np.median(ts)
    print(f"B={B} V={V} ld={ld} {str(dtype):15s} mask={mask_kind} rng={rng_mode}: median {med:8.1f} us  min {ts.min():8.1f} us  "
          f"{byt / med / 1e6:7.3f} TB/s ({byt / med / 1e6 / 8 * 100:5.1f}% of 8 TB/s)  bytes={byt / 1e6:.1f} MB", flush=True)
    return med


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--nbuf", type=int, default=4)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--pforbid", action="store_true", help="sweep the forbidden fraction of the masks")
    ap.add_argument("--stats", action="store_true", help="statistics only (no draw) on the two headline shapes")
    a = ap.parse_args()
    eng = HipEngine("cuda:0")
    f32, bf16 = torch.float32, torch.bfloat16
    if a.stats:
        for _ in range(2):
            run(eng, 1024, 50257, f32, 3, 0, a.nbuf, a.iters)
            run(eng, 512, 128256, bf16, 3, 0, a.nbuf, a.iters)
        sys.exit(0)
    if a.pforbid:
        for pf in (1 / 3, 0.05, 0.0001):
            print("pforbid", pf)
            run(eng, 1024, 50257, f32, 1, 1, a.nbuf, a.iters, pforbid=pf)
            run(eng, 1024, 50257, f32, 1, 0, a.nbuf, a.iters, pforbid=pf)
        sys.exit(0)
    run(eng, 1024, 50257, f32, 3, 1, a.nbuf, a.iters)
    run(eng, 512, 128256, bf16, 3, 1, a.nbuf, a.iters)
    run(eng, 1024, 50257, f32, 1, 1, a.nbuf, a.iters)
    run(eng, 1024, 50257, f32, 0, 1, a.nbuf, a.iters)
    run(eng, 1024, 50257, f32, 0, 0, a.nbuf, a.iters)
    run(eng, 1024, 50257, f32, 1, 1, a.nbuf, a.iters, ld=50304)
    if not a.quick:
        run(eng, 1024, 50257, f32, 1, 2, a.nbuf, a.iters)
        run(eng, 1024, 50257, f32, 2, 1, a.nbuf, a.iters)
        run(eng, 512, 128256, bf16, 1, 1, a.nbuf, a.iters)
        run(eng, 512, 128256, bf16, 0, 1, a.nbuf, a.iters)
        run(eng, 2048, 50257, f32, 1, 1, 2, a.iters)
        run(eng, 4096, 32000, bf16, 1, 1, a.nbuf, a.iters)
