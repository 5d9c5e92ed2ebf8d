"""Repeated launches of the one-launch kernels must agree bit for bit with themselves (fused step, log-softmax rows, trie
masses): waves meet through tagged records / LDS in whatever order the dispatcher and the memory system produce."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from tests import synth
eng = HipEngine("cuda:0"); dev = eng.device
def trial(B, V, dt, rng_mode, mask_kind, reps=150):
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
        l, s_, t = eng.step(x, rng_mode=rng_mode, seed=3, **kw)
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
trial(1024, 50257, f32, 1, 1, reps=60)
trial(1024, 50257, f32, 1, 0, reps=60)
trial(700, 128256, bf16, 1, 1, reps=40)
trial(700, 151936, f16, 1, 1, reps=40)
trial(2048, 32000, bf16, 1, 1, reps=40)



def trial_lsm(B, V, dt, same, reps=60):
    x = torch.from_numpy(synth.logits(V + 1, B, V)).to(dt).to(dev)
    bad = 0
    for rep in range(reps):
        out, lse = eng.log_softmax_rows(x, want_lse=True, out_dtype=dt if same else torch.float32)
        torch.cuda.synchronize()
        if rep == 0: o0, l0 = out.clone(), lse.clone()
        elif not (torch.equal(out.view(torch.int16 if out.element_size() == 2 else torch.int32), o0.view(torch.int16 if out.element_size() == 2 else torch.int32)) and torch.equal(lse, l0)): bad += 1
    eng.check()
    print(f"log_softmax_rows B={B} V={V} {dt} -> {'same' if same else 'f32'}: inconsistent launches {bad}/{reps}", flush=True)
trial_lsm(1024, 50257, f32, False)
trial_lsm(1024, 50257, bf16, True)
trial_lsm(512, 128256, bf16, True)
trial_lsm(512, 128256, bf16, False)
trial_lsm(300, 200003, f16, True)

def trial_trie(B, n_words, reps=40):
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie
    rs = np.random.default_rng(n_words)
    words, seen = [], set()
    while len(words) < n_words:
        w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
        if w not in seen: seen.add(w); words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=eng)
    x = torch.randn((B, n_words), device=dev) * 3
    lse = eng.row_lse(x)
    bad = 0
    for rep in range(reps):
        rows = trie.masses_from_logits(x, lse)
        torch.cuda.synchronize()
        if rep == 0: r0 = rows.clone()
        elif not torch.equal(rows.view(torch.int32), r0.view(torch.int32)): bad += 1
    print(f"trie rows B={B} V={n_words}: inconsistent launches {bad}/{reps}", flush=True)
trial_trie(1024, 50257)
trial_trie(7, 50257)
