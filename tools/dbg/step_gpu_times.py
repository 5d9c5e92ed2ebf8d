"""GPU time of every step of a DeviceSIS run (events between steps): where the restart step (the prompts' encoding) stands."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from genlm_backend_amd.sis import SisBenchWorkload

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
dev = torch.device("cuda", 0)
eng = HipEngine("cuda:0")
wl = SisBenchWorkload(eng, dev, 0, 1, None, n_particles=512 if model != 'gpt2' else 1024, particle_kv=True, model=model)
for i in range(12): wl.step(i, False)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    wl.step(12 + i, False); ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(40)]
print("ms per step:", " ".join(f"{x:.2f}" for x in t))
print("mean", np.mean(t), "median", np.median(t))
