import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from oracle import oracle as O
from tests import synth
eng = HipEngine("cuda:0"); dev = eng.device
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 41
N, U, V = 700, 300, 50257
x = synth.logits(V + U, U, V)
row_of = (np.arange(N) * 7 % U).astype(np.int32)
lo0, so0, to0 = O.step(x, row_of=row_of, rng_mode=O.RNG_PHILOX, seed=77, offset=5, particle_base=3)
xd = torch.from_numpy(x).to(dev); rd = torch.from_numpy(row_of).to(dev)
for rep in range(3):
    l, s_, t = eng.step(xd, row_of=rd, rng_mode=1, seed=77, offset=5, particle_base=3, variant=VAR)
    torch.cuda.synchronize()
    t = t.cpu().numpy(); s_ = s_.cpu().numpy()
    bad = np.nonzero(t != to0)[0]; bads = np.nonzero(s_.view(np.uint32) != so0.view(np.uint32))[0]
    print("unmasked rep", rep, "tok bad", len(bad), bad[:12], "lse bad", len(bads), bads[:12])
    if len(bads): print("   got", s_[bads][:6], "want", so0[bads][:6])
K = 3
masks = synth.binary_masks(V + 1, K, V); masks[2, 50:] = -np.inf
mid = (np.arange(N) % K).astype(np.int32)
bits, _ = O.mask_f32_to_bits(masks)
lo, so, to = O.step(x, row_of=row_of, rng_mode=O.RNG_PHILOX, seed=77, offset=5, particle_base=3, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
for rep in range(3):
    l, s_, t = eng.step(xd, row_of=rd, rng_mode=1, seed=77, offset=5, particle_base=3, variant=VAR,
                        mask_kind=1, mask=torch.from_numpy(bits.view(np.int32)).to(dev), mask_id=torch.from_numpy(mid).to(dev))
    torch.cuda.synchronize()
    t = t.cpu().numpy(); l = l.cpu().numpy(); s_ = s_.cpu().numpy()
    bad = np.nonzero(t != to)[0]
    badz = np.nonzero(l.view(np.uint32) != lo.view(np.uint32))[0]
    bads = np.nonzero(s_.view(np.uint32) != so.view(np.uint32))[0]
    print("masked rep", rep, "tok bad", len(bad), bad[:12], "logZ bad", len(badz), badz[:12], "lse bad", len(bads), bads[:12])
    if len(badz): print("   mid of bad", mid[badz][:12], "got", l[badz][:6], "want", lo[badz][:6])
