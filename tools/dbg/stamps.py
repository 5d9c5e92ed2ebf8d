"""Where the one-launch step spends its time (diagnostic build: `make -C genlm-backend_amd/csrc dbg`).
Every stats wave leaves its start / end time (s_memrealtime, 100 MHz) in the padding of its record, every finishing wave
[start, records complete, token written] in the buffer handed over as out_margin.  Usage: stamps.py [gpt2|llama] [iters]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import genlm_backend_amd  # noqa: E402
from genlm_backend_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("GLB_DBG_LIB", "libglb_hip_dbg.so"))
from genlm_backend_amd.engine import HipEngine  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B, V, dt = (1024, 50257, torch.float32) if shape == "gpt2" else (512, 128256, torch.bfloat16)
eng = HipEngine("cuda:0", contract=os.environ.get("GLB_CONTRACT", "poly"))
dev = eng.device
g = torch.Generator(device=dev)
g.manual_seed(0)
bufs = [(torch.randn((B, V), device=dev, generator=g) * 3).to(dt) for _ in range(4)]
maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
bits, _ = eng.mask_to_bits(maskf)
masks = eng.prepare_masks(bits, V, dt)
mid = (torch.arange(B, device=dev) % 2).to(torch.int32)
out = (torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
stamps = torch.zeros(B * 16, device=dev)
nch = (V + 4095) // 4096
rows = []
for i in range(iters):
    eng.step(bufs[i % 4], mask=masks, row_mask_id=mid, rng_mode=1, seed=1, offset=i, out=out, out_margin=stamps)
    torch.cuda.synchronize()
    if i < 4:
        continue
    ws = eng._step_ws[: B * nch * 128].view(torch.int64).view(B, nch, 16).cpu().numpy()
    st = stamps.view(torch.int64).view(B, 8).cpu().numpy()
    t0 = ws[:, :, 12].min()
    s_start, s_end = (ws[:, :, 12] - t0) / 100.0, (ws[:, :, 13] - t0) / 100.0  # microseconds
    f = (st[:, :3] - t0) / 100.0
    rows.append((s_start.max(), s_end.max(), np.percentile(s_end, 50), np.percentile(s_end, 90), f[:, 0].min(), f[:, 0].max(),
                 f[:, 2].max(), (f[:, 2] - f[:, 1]).mean(), (f[:, 2] - f[:, 1]).max(), (f[:, 2] - f[:, 1]).min(),
                 (f[:, 1] - f[:, 0]).mean(), (s_end - s_start).mean(), np.percentile(f[:, 2], 50),
                 ((st[:, 3] - st[:, 1]) / 100.0)[st[:, 3] > 0].mean(), ((st[:, 2] - st[:, 3]) / 100.0)[st[:, 3] > 0].mean()))
    if i == iters - 1:
        last = s_end.max(axis=1)  # row completion
        lag = f[:, 2] - last      # finisher end after its row's records
        print("per-particle finish lag after row completion: mean %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (lag.mean(), np.percentile(lag, 50), np.percentile(lag, 90), lag.max()))
        ts = np.arange(1.0, s_end.max() + 1, 2.0)
        act = [(int(((s_start <= t) & (s_end > t)).sum()), int(((f[:, 0] <= t) & (f[:, 2] > t)).sum())) for t in ts]
        print("resident waves (stats, finishing) at t us: " + "  ".join("%g:%d,%d" % (t, a, b) for t, (a, b) in zip(ts, act)))
        print("first stats wave start %.2f end %.2f; starts p1 %.2f p10 %.2f" % (s_start.min(), s_end.min(), np.percentile(s_start, 1), np.percentile(s_start, 10)))
        order = np.argsort(last)
        print("last 6 rows to complete: row, records done at, finisher start, records seen, token at")
        for r in order[-6:]:
            print("  %4d  %.2f  %.2f  %.2f  %.2f" % (r, last[r], f[r, 0], f[r, 1], f[r, 2]))
        order = np.argsort(f[:, 2])
        print("last 10 tokens written: particle, records done at, finisher start, records seen, token at")
        for r in order[-10:]:
            print("  %4d  %.2f  %.2f  %.2f  %.2f" % (r, last[r], f[r, 0], f[r, 1], f[r, 2]))
    if i == iters - 1 and (st[:, 4] > 0).all():
        seg = lambda a, b: ((st[:, b] - st[:, a]) / 100.0)
        print("finishing wave, per segment (us; mean / min): records complete -> folded %.2f / %.2f; -> chunk picked %.2f / %.2f; -> "
              "row picked, loads issued %.2f / %.2f; -> logarithms done %.2f / %.2f; -> quarter chunk arrived %.2f / %.2f; -> token %.2f / %.2f"
              % (seg(1, 4).mean(), seg(1, 4).min(), seg(4, 5).mean(), seg(4, 5).min(), seg(5, 6).mean(), seg(5, 6).min(),
                 seg(6, 7).mean(), seg(6, 7).min(), seg(7, 3).mean(), seg(7, 3).min(), seg(3, 2).mean(), seg(3, 2).min()))
names = ["last stats wave start", "last stats wave end", "stats end p50", "stats end p90", "first finisher start", "last finisher start",
         "last token written", "finish work mean (records seen -> token)", "finish work max", "finish work min", "wait for records mean",
         "stats wave lifetime mean", "token written p50", "records seen -> quarter chunk arrived (fold, picks, reload)",
         "quarter chunk arrived -> token (partials, picks)"]
a = np.array(rows)
for n, col in zip(names, a.T):
    print("%-46s mean %7.2f  min %7.2f  max %7.2f us" % (n, col.mean(), col.min(), col.max()))
