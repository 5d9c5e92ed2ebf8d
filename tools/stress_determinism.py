import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from tests import synth
eng = HipEngine("cuda:0"); dev = eng.device
def trial(B, V, dt, rng_mode, mask_kind, reps=150, variant=-1):
    x = torch.from_numpy(synth.logits(V, B, V)).to(dt).to(dev)
    K = 2
    maskf = torch.from_numpy(synth.binary_masks(V, K, V)).to(dev)
    bits, _ = eng.mask_to_bits(maskf)
    mid = (torch.arange(B, device=dev) % K).to(torch.int32)
    E = torch.empty((B, V), device=dev).exponential_()
    kw = dict(mask_kind=1, mask=bits, mask_id=mid) if mask_kind == 1 else (dict(mask_kind=2, mask=maskf.contiguous(), mask_id=mid) if mask_kind == 2 else {})
    if rng_mode == 2: kw["noise"] = E
    bad = 0
    for rep in range(reps):
        l, s_, t = eng.step(x, rng_mode=rng_mode, seed=3, variant=variant, **kw)
        torch.cuda.synchronize()
        if rep == 0: l0, s0, t0 = l.clone(), s_.clone(), t.clone()
        elif not (torch.equal(l, l0) and torch.equal(s_, s0) and torch.equal(t, t0)): bad += 1
    print(f"B={B} V={V} {dt} rng={rng_mode} mask={mask_kind}: inconsistent launches {bad}/{reps}", flush=True)
f32, bf16, f16 = torch.float32, torch.bfloat16, torch.float16
trial(4, 32000, bf16, 2, 1)

trial(4, 32000, bf16, 2, 2)
trial(4, 32000, f16, 2, 1)
trial(4, 32000, f32, 2, 1)

trial(4, 32000, bf16, 1, 1)

trial(64, 32000, bf16, 2, 1)
trial(4, 128256, bf16, 2, 1)
trial(4, 8000, bf16, 2, 1)
# persistent kernel: repeated launches must agree bit for bit
trial(1024, 50257, f32, 1, 1, reps=60, variant=0)
trial(1024, 50257, f32, 1, 0, reps=60, variant=0)
trial(700, 128256, bf16, 1, 1, reps=40, variant=0)
trial(700, 151936, f16, 1, 1, reps=40, variant=0)
trial(2048, 32000, bf16, 1, 1, reps=40, variant=0)

