"""The fused step's second arithmetic contract on 16-bit rows (include/glb.h GLB_STEP_HW_EXP: terms by v_exp_f32) against
the oracle.  v_exp_f32 is within one ulp of 2^y but not correctly rounded, so this contract has no bit-exact CPU
restatement; the acceptance bars are
  (i)   logZ / lse within 1e-4 of the oracle - both of its contracts - on random cases and at 512 x 128256 (observed 1e-6);
  (ii)  parity-mode (GLB_RNG_NOISE) tokens identical to torch's on every 16-bit golden, margins as reported;
  (iii) Philox tokens equal to the oracle's restatement (exp2f terms) except where the draw lies within 2^-20 of a
        boundary of the inverse CDF it walks - the exceptions are counted and printed;
and, what makes it a contract at all: on the GPU the results do not depend on launch geometry (one-wave and four-wave
statistics kernels, one launch or two, rows shared or not, particle shards), bit for bit."""
import os

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
EDGE = 2.0 ** -20
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _np(t):
    return t.detach().cpu().numpy()


def _mk16(x, dtype):
    if dtype == "f32":
        return x, torch.from_numpy(x)
    if dtype == "bf16":
        t = torch.from_numpy(x).to(torch.bfloat16)
        return t.view(torch.int16).numpy().view(np.uint16), t
    t = torch.from_numpy(x).to(torch.float16)
    return t.numpy(), t


def _close(got, want, what):
    """1e-4 where finite, the same infinities / NaNs elsewhere; returns the largest finite difference"""
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), what
    assert np.array_equal(got[~fin].view(np.uint32), want[~fin].view(np.uint32)) or np.array_equal(np.isnan(got[~fin]), np.isnan(want[~fin])), what
    d = float(np.abs(got[fin] - want[fin]).max()) if fin.any() else 0.0
    assert d < 1e-4, (what, d)
    return d


def check_hw(res, want_hw, want_poly, edge, what=""):
    """res: (logZ, lse, tok) device tensors of a contract="hw" call; want_*: the oracle's (logZ, lse, tok) under its two
    contracts; edge: the oracle's distance of every draw from the nearest CDF boundary.  Returns (max |d logZ|, max |d lse|,
    number of tokens that differ)."""
    logZ, lse, tok = res
    torch.cuda.synchronize()
    dz = max(_close(_np(logZ), want_hw[0], what + " logZ"), _close(_np(logZ), want_poly[0], what + " logZ vs poly"))
    dl = max(_close(_np(lse), want_hw[1], what + " lse"), _close(_np(lse), want_poly[1], what + " lse vs poly"))
    n_diff = 0
    if tok is not None:
        differ = _np(tok) != want_hw[2]
        n_diff = int(differ.sum())
        assert (edge[differ] < EDGE).all(), (what, "a token differs from the oracle's away from every CDF boundary", edge[differ])
    return dz, dl, n_diff


CASES = [
    # (B, V, dtype, n_masks)
    (4, 32000, "bf16", 2),
    (4, 128256, "bf16", 2),
    (3, 50257, "f16", 2),
    (3, 777, "bf16", 3),
    (300, 4097, "bf16", 2),   # more than 512 chunks: the one-launch kernel
    (40, 70001, "f16", 1),
    (16, 50257, "f32", 2),    # float32 rows have the contract too (the host asks for it with contract="hw" only)
    (200, 8191, "f32", 3),
]


@pytest.mark.parametrize("B,V,dtype,K", CASES)
@pytest.mark.parametrize("mask_kind", ["none", "bits", "f32"])
def test_hw_contract_against_the_oracle(engine, oracle, B, V, dtype, K, mask_kind):
    O = oracle
    dev = engine.device
    x = synth.logits(V + B, B, V)
    x[np.random.default_rng(V).random((B, V)) < 0.005] = -np.inf
    x_np, x_t = _mk16(x, dtype)
    masks = synth.binary_masks(V, K, V)
    mid = (np.arange(B) % K).astype(np.int32)
    kw_o, kw_g = {}, {}
    if mask_kind == "bits":
        bits, _ = O.mask_f32_to_bits(masks)
        kw_o = dict(mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
        kw_g = dict(mask_kind=1, mask=torch.from_numpy(bits.view(np.int32)).to(dev), mask_id=torch.from_numpy(mid).to(dev))
    elif mask_kind == "f32":
        mf = masks.copy()
        mf[np.isfinite(mf)] = np.random.default_rng(3).standard_normal(np.isfinite(mf).sum()).astype(np.float32)
        kw_o = dict(mask_kind=O.MASK_F32, mask=mf, mask_id=mid)
        kw_g = dict(mask_kind=2, mask=torch.from_numpy(mf).to(dev), mask_id=torch.from_numpy(mid).to(dev))
    common = dict(rng_mode=O.RNG_PHILOX, seed=1234, offset=7, particle_base=11)
    want_poly = O.step(x_np, **common, **kw_o)
    *want_hw, edge = O.step(x_np, contract="hw", want_edge=True, **common, **kw_o)
    res = engine.step(x_t.to(dev), rng_mode=1, seed=1234, offset=7, particle_base=11, contract="hw", **kw_g)
    dz, dl, n_diff = check_hw(res, want_hw, want_poly, edge)
    assert dz < 2e-5 and dl < 2e-5  # (the bar is 1e-4; the two exponentials are an ulp apart)
    # "auto" leaves float32 rows on the polynomial, bit for bit, and puts 16-bit rows on the hardware exponential
    if B <= 4:
        x32 = torch.from_numpy(x).to(dev)
        a = engine.step(x32, rng_mode=1, seed=1, contract="auto")
        b = engine.step(x32, rng_mode=1, seed=1, contract="poly")
        assert all(torch.equal(p, q) for p, q in zip(a, b))
        if dtype != "f32":
            c = engine.step(x_t.to(dev), rng_mode=1, seed=1234, offset=7, particle_base=11, contract="auto", **kw_g)
            assert all(torch.equal(p, q) for p, q in zip(c, res))


def test_hw_contract_at_config5_full_size(engine, oracle):
    """512 x 128256 bf16, two shared prepared masks, Philox draws (BASELINE config 5): (i) and (iii) at full size."""
    O = oracle
    dev = engine.device
    B, V = 512, 128256
    x_np, x_t = _mk16(synth.logits(23, B, V), "bf16")
    bits, _ = O.mask_f32_to_bits(synth.binary_masks(23, 2, V))
    mid = (np.arange(B) % 2).astype(np.int32)
    common = dict(rng_mode=O.RNG_PHILOX, seed=99, offset=5, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
    want_poly = O.step(x_np, **common)
    *want_hw, edge = O.step(x_np, contract="hw", want_edge=True, **common)
    prep = engine.prepare_masks(torch.from_numpy(bits.view(np.int32)).to(dev), V, torch.bfloat16)
    res = engine.step(x_t.to(dev), mask=prep, row_mask_id=torch.from_numpy(mid).to(dev), rng_mode=1, seed=99, offset=5, contract="hw")
    dz, dl, n_diff = check_hw(res, want_hw, want_poly, edge, "config 5")
    print(f"\nconfig 5 under GLB_STEP_HW_EXP: max |logZ - oracle| {dz:.2e}, max |lse - oracle| {dl:.2e}, "
          f"{n_diff} of {B} Philox tokens differ from the oracle's exp2f restatement (each within 2^-20 of a CDF boundary)")
    assert dz < 1e-5 and dl < 1e-5 and n_diff <= 2
    # the polynomial contract on the same call: the same tokens except near a boundary (the polynomial is 2.7e-6 off)
    tok_poly = _np(engine.step(x_t.to(dev), mask=prep, row_mask_id=torch.from_numpy(mid).to(dev), rng_mode=1, seed=99, offset=5,
                               contract="poly")[2])
    assert np.array_equal(tok_poly, want_poly[2])
    assert (tok_poly != _np(res[2])).sum() <= 4


@pytest.mark.parametrize("tag", ["llama_bf16", "small_f16"])
def test_hw_contract_parity_tokens_are_torchs_on_the_16bit_goldens(engine, oracle, tag):
    """(ii) on tests/golden/torch_kernel_ops.npz: the race under the hardware exponential picks torch.multinomial's ids."""
    G = np.load(os.path.join(GOLD, "torch_kernel_ops.npz"))
    O = oracle
    dev = engine.device
    B, V = [int(v) for v in G[f"{tag}::shape"]]
    x_np, x_t = _mk16(synth.logits(11, B, V), "bf16" if "bf16" in tag else "f16")
    bits, _ = O.mask_f32_to_bits(synth.binary_masks(11, 2, V))
    mid = (np.arange(B) % 2).astype(np.int32)
    E, _ = O.mt_exponential(1234, B * V)
    margin = torch.empty(B, device=dev)
    logZ, lse, tok = engine.step(x_t.to(dev), mask_kind=1, mask=torch.from_numpy(bits.view(np.int32)).to(dev),
                                 mask_id=torch.from_numpy(mid).to(dev), rng_mode=2, noise=torch.from_numpy(E.reshape(B, V)).to(dev),
                                 out_margin=margin, contract="hw")
    torch.cuda.synchronize()
    # torch's ids on the upcast logits = the oracle's polynomial-contract ids (tests/test_oracle.py pins those)
    want = O.step(x_np, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_NOISE, noise=E.reshape(B, V), want_margin=True)
    assert np.array_equal(_np(tok), want[2])
    assert np.abs(_np(margin) - want[3]).max() < 1e-4 and np.abs(_np(lse) - G[f"{tag}::lse32"]).max() < 1e-4


def test_hw_contract_parity_at_config5_full_size_is_torchs(engine, oracle):
    """(ii) at 512 x 128256 bf16 against the torch-made golden (ref_round6.npz, oracle/make_goldens_r6.py), noise from the
    device's own MT19937 stream, under BOTH contracts: every id torch.multinomial's, logZ / lse within 1e-4, margins."""
    gold = np.load(os.path.join(GOLD, "ref_round6.npz"))
    dev = engine.device
    B, V = 512, 128256
    _, x_t = _mk16(synth.logits(23, B, V), "bf16")
    bits, _ = oracle.mask_f32_to_bits(synth.binary_masks(23, 2, V))
    mid = torch.from_numpy((np.arange(B) % 2).astype(np.int32)).to(dev)
    noise = engine.noise_rng(2025, V).rows(B)
    x_d = x_t.to(dev)
    for contract in ("poly", "hw"):
        margin = torch.empty(B, device=dev)
        logZ, lse, tok = engine.step(x_d, mask_kind=1, mask=torch.from_numpy(bits.view(np.int32)).to(dev), mask_id=mid, rng_mode=2,
                                     noise=noise, out_margin=margin, contract=contract)
        torch.cuda.synchronize()
        assert np.array_equal(_np(tok), gold["parity512_llama::token"]), contract
        assert np.abs(_np(logZ) - gold["parity512_llama::logZ"]).max() < 1e-4
        assert np.abs(_np(lse) - gold["parity512_llama::lse"]).max() < 1e-4
        assert np.abs(_np(margin) - gold["parity512_llama::margin"]).max() < 1e-3


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_hw_contract_does_not_depend_on_launch_geometry(engine, dtype):
    """Deterministic and shard-invariant on the GPU: one row alone (four waves per chunk, two launches), the same row
    among 300 (one-wave statistics, one launch), shared by 700 particles, and a population cut into shards with
    particle_base give the same bits for logZ / lse / tokens."""
    dev = engine.device
    V = 50257
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = (torch.randn((300, V), device=dev, generator=g) * 3).to(torch.bfloat16 if dtype == "bf16" else torch.float32)
    maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
    bits, _ = engine.mask_to_bits(maskf)
    prep = engine.prepare_masks(bits, V, x.dtype)
    rid = (torch.arange(300, device=dev) % 2).to(torch.int32)
    kw = dict(rng_mode=1, seed=77, offset=3, contract="hw")
    full = [t.clone() for t in engine.step(x, mask=prep, row_mask_id=rid, **kw)]
    # row 5 alone: n_rows = 1 takes chunk_stats_small_kernel + finish_kernel; particle_base keeps its Philox counter
    one = engine.step(x[5:6], mask=prep, row_mask_id=rid[5:6], particle_base=5, **kw)
    assert all(torch.equal(a[5:6], b) for a, b in zip(full, one))
    # shards of the population
    parts = [engine.step(x[s:e], mask=prep, row_mask_id=rid[s:e], particle_base=s, **kw) for s, e in ((0, 100), (100, 101), (101, 300))]
    for k in range(3):
        assert torch.equal(torch.cat([p[k] for p in parts]), full[k])
    # 700 particles on shared rows: statistics once per row, the draws per particle
    row_of = (torch.arange(700, device=dev) * 7 % 300).to(torch.int32)
    shared = engine.step(x, row_of=row_of, mask=prep, row_mask_id=rid, **kw)
    own = engine.step(x[row_of.long()].contiguous(), mask=prep, row_mask_id=rid[row_of.long()].contiguous(), **kw)
    assert all(torch.equal(a, b) for a, b in zip(shared, own))
    assert torch.equal(shared[0], full[0][row_of.long()]) and torch.equal(shared[1], full[1][row_of.long()])


def test_undefined_flag_bits_are_refused_through_the_c_abi(engine):
    """glb_step_args.flags: GLB_STEP_HW_EXP or nothing; any other bit is GLB_EINVAL (include/glb.h)."""
    import ctypes as C

    from genlm_backend_amd._lib import GLB_EINVAL, STEP_HW_EXP

    x = torch.zeros((2, 100), device=engine.device)
    plan = engine.step_plan(x, rng_mode=1, seed=1)
    for flags in (2, 4 | STEP_HW_EXP, -1):
        plan.args.flags = flags
        assert engine.lib.glb_logprob_mask_sample(C.byref(plan.args), engine._stream()) == GLB_EINVAL
    for flags in (0, STEP_HW_EXP):
        plan.args.flags = flags
        assert engine.lib.glb_logprob_mask_sample(C.byref(plan.args), engine._stream()) == 0


def test_hw_contract_parity_at_the_fp32_headline_size_is_torchs(engine, oracle):
    """(ii) for float32 rows: 1024 x 50257 in parity mode under the hardware exponential against the torch-made golden
    (ref_round2.npz, parity1024::*): every id torch.multinomial's, logZ within 1e-4, the race's margins as torch has them."""
    from genlm_backend_amd.engine import DeviceRng

    gold = np.load(os.path.join(GOLD, "ref_round2.npz"))
    B, V = 1024, 50257
    dev = engine.device
    bits, _ = oracle.mask_f32_to_bits(synth.binary_masks(21, 2, V))
    mid = torch.from_numpy((np.arange(B) % 2).astype(np.int32)).to(dev)
    margin = torch.empty(B, device=dev)
    logZ, lse, tok = engine.step(torch.from_numpy(synth.logits(21, B, V)).to(dev), mask_kind=1,
                                 mask=torch.from_numpy(bits.view(np.int32)).to(dev), mask_id=mid, rng_mode=2,
                                 noise=DeviceRng(engine, 2024, V).rows(B), out_margin=margin, contract="hw")
    torch.cuda.synchronize()
    assert np.array_equal(_np(tok), gold["parity1024::token"])
    assert np.abs(_np(logZ) - gold["parity1024::logZ"]).max() < 1e-4
    assert np.abs(_np(margin) - gold["parity1024::margin"]).max() < 1e-3
