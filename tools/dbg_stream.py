import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from oracle import oracle as O
from tests import synth
eng = HipEngine("cuda:0"); dev = eng.device
for (B, V) in [(5, 70001), (6, 1000)]:
    x = synth.logits(V, B, V)
    lo, so, to = O.step(x, rng_mode=O.RNG_PHILOX, seed=5, offset=2)
    l, s_, t = eng.step(torch.from_numpy(x).to(dev), rng_mode=1, seed=5, offset=2, variant=99 if V < 65000 else 0)
    torch.cuda.synchronize()
    print(V, "logZ", l.cpu().numpy(), lo); print("  lse", s_.cpu().numpy(), so); print("  tok", t.cpu().numpy(), to)
