"""Timing of the small kernels of a step (HIP events, median of 50): glb_group_contexts at SIS-like populations."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
eng = HipEngine("cuda:0"); dev = eng.device
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))
rs = np.random.default_rng(0)
for n in (1024, 4096, 16384):
    for distinct in (1, n):
        cap = 19
        mat = np.zeros((n, cap), np.int32)
        base = rs.integers(0, 50000, (distinct, 13))
        mat[:, :13] = base[rs.integers(0, distinct, n)] if distinct > 1 else base[0]
        if distinct == n: mat[:, :13] = rs.integers(0, 50000, (n, 13))
        tok = torch.from_numpy(mat.reshape(-1)).to(dev); st = (torch.arange(n, device=dev, dtype=torch.int64) * cap)
        ln = torch.full((n,), 13, dtype=torch.int32, device=dev)
        print(f"group_contexts n={n} distinct={distinct}: {t(lambda: eng.group_contexts(tok, st, ln)):.1f} us", flush=True)
