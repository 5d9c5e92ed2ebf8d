"""Host-side profile (cProfile) of the DeviceSIS step with shared KV rows: where the CPU time of a 3.6 ms step goes.
Usage: sis_host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd  # noqa: E402,F401
from genlm_backend_amd.engine import HipEngine  # noqa: E402
from genlm_backend_amd.sis import SisBenchWorkload  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
eng = HipEngine(dev)
w = SisBenchWorkload(eng, dev, 0, 1, None, particle_kv=True)
for i in range(20):
    w.step(i, False)
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
for i in range(steps):
    w.step(20 + i, False)
pr.disable()
torch.cuda.synchronize()
print("ms per step (wall):", (time.perf_counter() - t0) / steps * 1e3)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
