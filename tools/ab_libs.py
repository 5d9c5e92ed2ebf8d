"""Same-box A/B of builds of the library on the fused step's headline shapes (launch events, default contracts):
tools/ab_libs.py <rounds> <lib under tools/dbg | ""> ...  - every (round, library) is a fresh process (tools/ab.sh's kbench
form times between stream markers; this one by the launch's own events, on prepared masks, as bench.py's kernel workloads do)."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd import _lib

    if os.environ.get("GLB_DBG_LIB"):
        _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", os.environ["GLB_DBG_LIB"])
    from genlm_backend_amd.engine import HipEngine

    eng = HipEngine("cuda:0")
    dev = eng.device
    for B, V, dt in ((1024, 50257, torch.float32), (512, 128256, torch.bfloat16)):
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        bufs = [(torch.randn((B, V), device=dev, generator=g) * 3).to(dt) for _ in range(4)]
        maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
        bits, _ = eng.mask_to_bits(maskf)
        masks = eng.prepare_masks(bits, V, dt)
        mid = (torch.arange(B, device=dev) % 2).to(torch.int32)
        out = (torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
        plans = [eng.step_plan(x, mask=masks, row_mask_id=mid, rng_mode=1, seed=1, offset=0, out=out) for x in bufs]
        evs = eng.timing_events(300)
        for i in range(8):
            plans[i % 4].run(offset=i)
        torch.cuda.synchronize()
        for i in range(300):
            plans[i % 4].run_timed(evs[i], offset=i)
        torch.cuda.synchronize()
        ts = np.array([a.elapsed_time(b) * 1e3 for a, b in evs])
        byt = B * V * bufs[0].element_size() + B * 8 + 2 * ((V + 31) // 32) * 4
        print(f"  B={B} V={V} {str(dt):15s}: median {np.median(ts):7.2f} us  mean {ts.mean():7.2f}  min {ts.min():7.2f}  {byt / np.median(ts) / 1e6 / 8 * 100:5.1f}% of 8 TB/s", flush=True)
    sys.exit(0)

rounds = int(sys.argv[1])
for r in range(rounds):
    for lib in sys.argv[2:]:
        print("--", lib or "(product)", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, GLB_DBG_LIB=lib))
