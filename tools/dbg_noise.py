import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from oracle import oracle as O
from tests import synth
eng = HipEngine("cuda:0"); dev = eng.device
B, V, K = 4, 32000, 2
x = synth.logits(V, B, V)
xb = torch.from_numpy(x).to(torch.bfloat16)
x_np = xb.view(torch.int16).numpy().view(np.uint16)
masks = synth.binary_masks(V, K, V)
bits, _ = O.mask_f32_to_bits(masks)
mid = (np.arange(B) % K).astype(np.int32)
E, _ = O.mt_exponential(4321, B * V); E = E.reshape(B, V)
try:
    lo, so, to = O.step(x_np, dtype=O.BF16, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_NOISE, noise=E)
except TypeError:
    lo, so, to = O.step(x_np, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_NOISE, noise=E)
bd = torch.from_numpy(bits.view(np.int32)).to(dev); md = torch.from_numpy(mid).to(dev); Ed = torch.from_numpy(E).to(dev); xd = xb.to(dev)
nbad = 0
for rep in range(200):
    l, s_, t = eng.step(xd, mask_kind=1, mask=bd, mask_id=md, rng_mode=2, noise=Ed)
    torch.cuda.synchronize()
    t = t.cpu().numpy(); l = l.cpu().numpy(); s_ = s_.cpu().numpy()
    if not (np.array_equal(l.view(np.uint32), lo.view(np.uint32)) and np.array_equal(t, to) and np.array_equal(s_.view(np.uint32), so.view(np.uint32))):
        nbad += 1
        print(rep, "tok", t, "want", to, "logZ", l, "want", lo, "lse", s_, "want", so)
print("noise mode bad launches:", nbad, "of 200")
nbad = 0
lo2, so2, to2 = O.step(x_np, dtype=O.BF16, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_PHILOX, seed=3) if False else (None, None, None)
for rep in range(200):
    l, s_, t = eng.step(xd, mask_kind=1, mask=bd, mask_id=md, rng_mode=1, seed=3, variant=-1)
    torch.cuda.synchronize()
    if rep == 0: l0, s0, t0 = l.clone(), s_.clone(), t.clone()
    elif not (torch.equal(l, l0) and torch.equal(s_, s0) and torch.equal(t, t0)):
        nbad += 1; print("philox v1 rep", rep, l.cpu().numpy(), l0.cpu().numpy())
print("philox v1 mode inconsistent launches:", nbad, "of 200")
