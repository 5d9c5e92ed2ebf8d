"""Small calls of the step (one shared row: SIS step 0; a lone query): two launches (four waves a chunk, then the finishing
launch) against the one-launch form - GLB_DBG_LIB names a build with -DGLB_FUSED_MIN_ITEMS=0.  Timed by the launch's events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import genlm_backend_amd  # noqa: E402,F401
from genlm_backend_amd import _lib  # noqa: E402

if os.environ.get("GLB_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["GLB_DBG_LIB"])
from genlm_backend_amd.engine import HipEngine  # noqa: E402

eng = HipEngine("cuda:0")
dev = eng.device
print("lib:", _lib.LIB_PATH)
for U, N, V, dt in ((1, 1024, 50257, torch.float32), (1, 512, 128256, torch.bfloat16), (1, 1, 50257, torch.float32),
                    (8, 8, 50257, torch.float32), (16, 64, 128256, torch.bfloat16), (30, 1024, 50257, torch.float32)):
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    bufs = [(torch.randn((U, V), device=dev, generator=g) * 3).to(dt) for _ in range(4)]
    maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
    bits, _ = eng.mask_to_bits(maskf)
    masks = eng.prepare_masks(bits, V, dt)
    row_of = (torch.arange(N, device=dev) % U).to(torch.int32)
    mid = (torch.arange(U, device=dev) % 2).to(torch.int32)
    out = (torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, dtype=torch.int32, device=dev))
    plans = [eng.step_plan(x, row_of=row_of, mask=masks, row_mask_id=mid, rng_mode=1, seed=1, offset=0, out=out) for x in bufs]
    evs = eng.timing_events(200)
    for i in range(8):
        plans[i % 4].run(offset=i)
    torch.cuda.synchronize()
    ref = [t.clone() for t in out]
    for i in range(200):
        plans[i % 4].run_timed(evs[i], offset=i)
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) * 1e3 for a, b in evs])
    plans[7 % 4].run(offset=7)
    torch.cuda.synchronize()
    print(f"  {U:3d} rows x {V} {str(dt):15s} {N:5d} particles: median {np.median(ts):6.2f} us  mean {ts.mean():6.2f}  checksum {int(out[2].sum())} {float(out[0].sum()):.6f}", flush=True)
