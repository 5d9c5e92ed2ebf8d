import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from oracle import oracle as O
from tests import synth
eng = HipEngine('cuda:0')
for (B,V,var) in [(8,1000,0),(8,1000,2),(4,50257,0)]:
    x = synth.logits(1, B, V)
    lo, so, to = O.step(x, rng_mode=O.RNG_PHILOX, seed=5, offset=1)
    l, s_, t = eng.step(torch.from_numpy(x).cuda(), rng_mode=1, seed=5, offset=1, variant=var)
    torch.cuda.synchronize()
    print(B,V,var,'tok', t.cpu().numpy(), to)
    print(' lse', s_.cpu().numpy()[:4], so[:4])
    print(' logZ', l.cpu().numpy()[:4], lo[:4])
