"""Where the log-softmax launch (glb_log_softmax_rows, one wave per chunk) spends its time (diagnostic build:
`make -C genlm-backend_amd/csrc dbg`): every wave leaves [start, record out, lse in, stores drained] (s_memrealtime,
100 MHz) in the padding of its record.  Usage: stamps_lsm.py [gpt2|llama] [same|f32] [iters]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import genlm_backend_amd  # noqa: E402,F401
from genlm_backend_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("GLB_DBG_LIB", "libglb_hip_dbg.so"))
from genlm_backend_amd.engine import HipEngine  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "llama"
same = (sys.argv[2] if len(sys.argv) > 2 else "same") == "same"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, V, dt = (1024, 50257, torch.bfloat16) if shape == "gpt2" else (512, 128256, torch.bfloat16)
eng = HipEngine("cuda:0")
dev = eng.device
bufs = [(torch.randn((B, V), device=dev) * 3).to(dt) for _ in range(3)]
outs = [torch.empty((B, V), device=dev, dtype=dt if same else torch.float32) for _ in range(3)]
nch = (V + 4095) // 4096
acc = []
for i in range(iters):
    eng.log_softmax_rows(bufs[i % 3], out=outs[i % 3])
    torch.cuda.synchronize()
    if i < 3:
        continue
    st = eng._step_ws[: B * nch * 128].view(torch.int64).view(B, nch, 16).cpu().numpy()[:, :, 12:16]
    t = (st - st[:, :, 0].min()) / 100.0  # microseconds
    start, pub, got, end = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
    row_pub = pub.max(axis=1, keepdims=True)  # the row's last record out
    acc.append((end.max(), start.max(), (pub - start).mean(), (got - pub).mean(), (got - pub)[:, 1:].max(), (end - got).mean(),
                (end - start).mean(), (got - row_pub)[:, 1:].mean(), (got[:, 0] - row_pub[:, 0]).mean(), (row_pub - pub).mean(),
                (pub.max(axis=1) - pub.min(axis=1)).mean(), (start.max(axis=1) - start.min(axis=1)).mean()))
    if i == iters - 1:
        ts = np.arange(2.0, end.max() + 2, 4.0)
        print("resident waves (loading, waiting, storing) at t us: " + "  ".join(
            "%g:%d,%d,%d" % (x, ((start <= x) & (pub > x)).sum(), ((pub <= x) & (got > x)).sum(), ((got <= x) & (end > x)).sum()) for x in ts))
names = ["last wave done", "last wave start", "start -> record out (load + sums)", "record out -> lse in (wait)", "wait max",
         "lse in -> stores drained", "wave lifetime", "row's last record out -> lse in, row-mates", "row's last record out -> lse folded, first wave",
         "record out -> row's last record out", "spread of record-out times within a row", "spread of start times within a row"]
for n, col in zip(names, np.array(acc).T):
    print("%-52s mean %7.2f  min %7.2f  max %7.2f us" % (n, col.mean(), col.min(), col.max()))
