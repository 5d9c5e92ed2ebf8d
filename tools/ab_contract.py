"""Same-box A/B of the fused step's two arithmetic contracts on 16-bit rows (GLB_STEP_HW_EXP, include/glb.h):
512 x 128256 bf16 (BASELINE config 5) and 1024 x 50257 bf16, prepared masks, Philox draws; the two contracts alternate
call by call on rotating buffers, timed by the events the launch itself carries.  GLB_DBG_LIB names another build of the
library under tools/dbg/ (e.g. one made with -DGLB_STATS_WAVES_16_HW=4)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd  # noqa: E402,F401
from genlm_backend_amd import _lib  # noqa: E402

if os.environ.get("GLB_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", os.environ["GLB_DBG_LIB"])
from genlm_backend_amd.engine import HipEngine  # noqa: E402


def run(eng, B, V, dtype, iters, nbuf=4, rng_mode=1):
    dev = eng.device
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    bufs = [(torch.randn((B, V), device=dev, generator=g) * 3).to(dtype) for _ in range(nbuf)]
    maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
    bits, _ = eng.mask_to_bits(maskf)
    mid = (torch.arange(B, device=dev) % 2).to(torch.int32)
    masks = eng.prepare_masks(bits, V, dtype)
    out = (torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
    plans = {c: [eng.step_plan(x, mask=masks, row_mask_id=mid, rng_mode=rng_mode, seed=1, offset=0, out=out, contract=c) for x in bufs]
             for c in ("poly", "hw")}
    evs = {c: eng.timing_events(iters) for c in plans}
    for i in range(8):
        for c in plans:
            plans[c][i % nbuf].run(offset=i)
    torch.cuda.synchronize()
    # (the two contracts never read the same buffer back to back: a 131 MB buffer the other call has just streamed is
    # partly served by the 256 MiB Infinity Cache - 2 us of a 27 us call, measured the hard way)
    for i in range(iters):
        for k, c in enumerate(plans):
            plans[c][(2 * i + k) % nbuf].run_timed(evs[c][i], offset=i)
    torch.cuda.synchronize()
    byt = B * V * bufs[0].element_size() + B * 8 + 2 * ((V + 31) // 32) * 4
    for c in plans:
        ts = np.array([a.elapsed_time(b) * 1e3 for a, b in evs[c]])
        print(f"B={B} V={V} {str(dtype):15s} rng={rng_mode} contract={c:4s}: median {np.median(ts):7.2f} us  mean {ts.mean():7.2f}  min {ts.min():7.2f}  "
              f"{byt / np.median(ts) / 1e6 / 8 * 100:5.1f}% of 8 TB/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    eng = HipEngine("cuda:0")
    print("lib:", _lib.LIB_PATH)
    for _ in range(2):
        run(eng, 512, 128256, torch.bfloat16, a.iters)
        run(eng, 1024, 50257, torch.bfloat16, a.iters)
    run(eng, 1024, 50257, torch.float32, a.iters)
    run(eng, 1024, 50257, torch.float32, a.iters)
    run(eng, 512, 128256, torch.float16, a.iters)
    run(eng, 512, 128256, torch.bfloat16, a.iters, rng_mode=0)
