"""Pins the CPU oracle: layer A (reference semantics) and layer B (GLB math contract) against
torch-CPU goldens produced by oracle/make_goldens.py (tests/golden/torch_kernel_ops.npz) and against
live torch-CPU ops (torch is a third-party dependency of the reference, not the reference)."""
import os

import numpy as np
import pytest
import torch

from tests import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_kernel_ops.npz"))
TOL = 1e-4


def _inputs(O, tag):
    B, V = [int(v) for v in G[f"{tag}::shape"]]
    x = synth.logits(11, B, V)
    if "bf16" in tag:
        xt = torch.from_numpy(x).to(torch.bfloat16)
        x_in = xt.view(torch.int16).numpy().view(np.uint16)
        x32 = xt.float().numpy()
    elif "f16" in tag:
        xt = torch.from_numpy(x).to(torch.float16)
        x_in, x32 = xt.numpy(), xt.float().numpy()
    else:
        x_in = x32 = x
    masks = synth.binary_masks(11, 2, V)
    mid = (np.arange(B) % 2).astype(np.int32)
    return B, V, x_in, x32, masks, mid


def test_mt19937_exponential_stream_is_torch_cpu(oracle):
    E, _ = oracle.mt_exponential(99, 4096)
    assert np.array_equal(E.view(np.uint32), G["mt::seed99_first4096"].view(np.uint32))
    g = torch.Generator()
    g.manual_seed(31337)
    want = torch.empty(3 * 50257).exponential_(1, generator=g).numpy()
    got, _ = oracle.mt_exponential(31337, 3 * 50257)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("tag", ["gpt2_f32", "llama_bf16", "small_f16"])
def test_layer_b_step_matches_torch_goldens(oracle, tag):
    O = oracle
    B, V, x_in, x32, masks, mid = _inputs(O, tag)
    bits, nonbin = O.mask_f32_to_bits(masks)
    assert not nonbin
    E, _ = O.mt_exponential(1234, B * V)
    logZ, lse, tok = O.step(x_in, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_NOISE,
                            noise=E.reshape(B, V))
    # bf16 goldens carry the reference's bf16-rounded log-probs (cache.py:96 keeps the logits dtype): looser bar
    tol = TOL if "f32" in tag else 3e-2
    assert np.abs(logZ - G[f"{tag}::logZ"]).max() < tol
    assert np.abs(lse - G[f"{tag}::lse32"]).max() < TOL
    if "f32" in tag:
        assert np.array_equal(tok, G[f"{tag}::token"])  # sampled ids identical to torch.multinomial
        assert G[f"{tag}::race_margin"].min() > 1e-4      # ... and none of them was a near tie
    lp, _ = O.log_softmax_rows(x_in)
    assert np.abs(lp[:, :64] - G[f"{tag}::lp32_head"]).max() < TOL
    assert np.abs(lp.astype(np.float64).sum(-1) - G[f"{tag}::lp32_rowsum"]).max() / V < 1e-5
    # same masks as additive float rows: the masked sums are taken on scales of their own (float masks can raise
    # values, bit masks only gate terms), so logZ agrees to rounding and the race picks the same tokens
    logZ2, _, tok2 = O.step(x_in, mask_kind=O.MASK_F32, mask=masks, mask_id=mid, rng_mode=O.RNG_NOISE,
                            noise=E.reshape(B, V))
    fin = np.isfinite(logZ)
    assert np.array_equal(fin, np.isfinite(logZ2)) and np.abs(logZ[fin] - logZ2[fin]).max() < 2e-6
    assert np.array_equal(tok, tok2)


def test_16bit_tokens_match_torch_on_upcast_logits(oracle):
    """bf16 / f16 logits: the build computes in fp32 on the upcast values; against torch on the same
    upcast values the draws are identical."""
    O = oracle
    for tag in ("llama_bf16", "small_f16"):
        B, V, x_in, x32, masks, mid = _inputs(O, tag)
        lp = torch.log_softmax(torch.from_numpy(x32), -1)
        masked = lp + torch.from_numpy(masks)[torch.from_numpy(mid).long()]
        logZ_t = masked.logsumexp(-1)
        g = torch.Generator()
        g.manual_seed(77)
        tok_t = torch.multinomial((masked - logZ_t[:, None]).exp(), 1, generator=g).flatten().numpy()
        E, _ = O.mt_exponential(77, B * V)
        bits, _ = O.mask_f32_to_bits(masks)
        logZ, _, tok = O.step(x_in, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_NOISE,
                              noise=E.reshape(B, V))
        assert np.array_equal(tok, tok_t)
        assert np.abs(logZ - logZ_t.numpy()).max() < TOL


def test_layer_a_reference_semantics_match_torch(oracle):
    O = oracle
    x = synth.logits(3, 4, 5000)
    mask = synth.binary_masks(3, 1, 5000)[0]
    for r in range(4):
        lp_t = torch.log_softmax(torch.from_numpy(x[r]), 0)
        assert np.abs(O.ref_log_softmax(x[r]) - lp_t.numpy()).max() < 1e-5
        masked = lp_t + torch.from_numpy(mask)
        logZ_t = masked.logsumexp(-1)
        g = torch.Generator()
        g.manual_seed(5 + r)
        tok_t = torch.multinomial((masked - logZ_t).exp(), 1, generator=g).item()
        E, _ = O.mt_exponential(5 + r, 5000)
        z, t = O.ref_particle(lp_t.numpy(), mask, E)
        assert abs(z - logZ_t.item()) < 1e-5 and t == tok_t


def test_philox_known_answers(oracle):
    # Random123 known-answer vectors for philox4x32-10
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_edge_cases(oracle):
    O = oracle
    V = 37
    x = synth.logits(1, 3, V, outliers=1)
    # everything forbidden -> logZ = -inf, token = -1 (the reference's multinomial raises on the NaNs)
    bits = np.zeros((1, 2), np.uint32)
    logZ, lse, tok = O.step(x, mask_kind=O.MASK_BITS, mask=bits, rng_mode=O.RNG_PHILOX, seed=1)
    assert np.all(np.isneginf(logZ)) and np.all(tok == -1) and np.all(np.isfinite(lse))
    # a single allowed token is always drawn; logZ is its log-prob
    bits = np.zeros((1, 2), np.uint32)
    bits[0, 0] = 1 << 5
    logZ, lse, tok = O.step(x, mask_kind=O.MASK_BITS, mask=bits, rng_mode=O.RNG_PHILOX, seed=1)
    assert np.all(tok == 5)
    assert np.abs(logZ - (x[:, 5] - lse)).max() < 1e-5
    # forbidden top token: masked maximum sits in a lower binade than the row maximum
    x2 = x.copy()
    x2[:, 7] += 60
    bits = np.full((1, 2), 0xffffffff, np.uint32)
    bits[0, 0] &= ~np.uint32(1 << 7)
    logZ, lse, tok = O.step(x2, mask_kind=O.MASK_BITS, mask=bits, rng_mode=O.RNG_PHILOX, seed=3)
    lp = torch.log_softmax(torch.from_numpy(x2), -1)
    m = torch.zeros(V)
    m[7] = float("-inf")
    want = (lp + m).logsumexp(-1).numpy()
    assert np.all(tok != 7) and np.abs(logZ - want).max() < 1e-3 * np.abs(want).max()
    # temperature scaling (base.py:136)
    logZ, lse, _ = O.step(x, logit_scale=0.5)
    assert np.abs(lse - torch.logsumexp(torch.from_numpy(x) * 0.5, -1).numpy()).max() < 1e-5
    # no mask: logZ == 0 up to rounding (SURVEY.md §8c observed -1.9e-6 in the reference)
    assert np.abs(logZ).max() < 1e-5


def test_host_ops(oracle):
    O = oracle
    ctxs = [[1, 2, 3, 4], [1, 2, 3, 5], [1, 2, 3, 4], [7, 8], [], [7, 8], [1, 2, 3]]
    g, rep, ng = O.group_contexts(ctxs)
    assert list(g) == [0, 1, 0, 2, 3, 2, 4] and list(rep) == [0, 1, 3, 4, 6] and ng == 5
    pre = [[1, 2, 3], [1, 2], [7, 8], [9]]
    p, b = O.match_prefixes(ctxs, pre)
    # deepest cached prefix that is a PROPER prefix (hf.py:334-342): [7, 8] itself never matches [7, 8]
    assert list(p) == [0, 0, 0, -1, -1, -1, 1] and list(b) == [3, 3, 3, 0, 0, 0, 2]
    ids, am, pos, last = O.gather_padded(ctxs, sel=[0, 3, 6], base=b, pad_id=99, p_max=3, l_max=2)
    assert ids.tolist() == [[4, 99], [7, 8], [3, 99]]
    assert pos.tolist() == [[3, 0], [0, 1], [2, 0]]
    assert am.tolist() == [[1, 1, 1, 1, 0], [0, 0, 0, 1, 1], [1, 1, 0, 1, 0]]  # hf.py:58-64 layout
    assert last.tolist() == [0, 1, 0]
    slabs = [np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4), -np.arange(2 * 2 * 4, dtype=np.float32).reshape(2, 2, 4)]
    kv = O.gather_kv_padded(slabs, [0, -1, 1], 3)
    assert np.array_equal(kv[0], slabs[0]) and not kv[1].any()
    assert np.array_equal(kv[2][:, :2], slabs[1]) and not kv[2][:, 2:].any()  # zero padded on the seq axis (hf.py:33-53)
    lw = np.array([-3.0, -1.0, -2.5, -60.0], np.float32)
    probs, stats = O.normalize_weights(lw)
    t = torch.from_numpy(lw)
    assert np.abs(probs - torch.exp(t - t.logsumexp(0)).numpy()).max() < 1e-6  # README.md:108-110
    assert abs(stats[0] - t.logsumexp(0).item()) < 1e-6
    w = torch.softmax(t, 0)
    assert abs(stats[1] - (1.0 / (w * w).sum()).item()) < 1e-4


def test_oracle_at_config5_full_size_against_the_torch_golden(oracle):
    """512 x 128256 bf16 (BASELINE config 5) in parity mode against torch-CPU's log_softmax + mask + logsumexp + multinomial
    on the upcast logits (tests/golden/ref_round6.npz, oracle/make_goldens_r6.py): every id identical, logZ / lse within
    1e-4, margins as torch's race has them - under BOTH contracts of the terms (the polynomial, and the hardware
    exponential as the oracle restates it with exp2f)."""
    O = oracle
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_round6.npz"))
    B, V = 512, 128256
    x = torch.from_numpy(synth.logits(23, B, V)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    bits, _ = O.mask_f32_to_bits(synth.binary_masks(23, 2, V))
    mid = (np.arange(B) % 2).astype(np.int32)
    E, _ = O.mt_exponential(2025, B * V)
    assert gold["parity512_llama::margin"].min() > 1e-3  # no golden draw is a near tie
    for contract in ("poly", "hw"):
        logZ, lse, tok, margin = O.step(x, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_NOISE,
                                        noise=E.reshape(B, V), want_margin=True, contract=contract)
        assert np.array_equal(tok, gold["parity512_llama::token"]), contract
        assert np.abs(logZ - gold["parity512_llama::logZ"]).max() < TOL
        assert np.abs(lse - gold["parity512_llama::lse"]).max() < TOL
        assert np.abs(margin - gold["parity512_llama::margin"]).max() < 1e-3


def test_hw_contract_of_the_oracle_is_the_polynomial_within_tolerance(oracle):
    """The oracle's restatement of GLB_STEP_HW_EXP (exp2f terms) against its polynomial contract: lse / logZ within 1e-5,
    Philox tokens identical except where a draw lies within 2^-16 of a boundary of the inverse CDF (the polynomial is
    2.7e-6 off 2^f); -inf and NaN logits, masks of every kind."""
    O = oracle
    rs = np.random.default_rng(7)
    for V, B in ((5000, 12), (40000, 6)):
        x = synth.logits(V, B, V)
        x[rs.random((B, V)) < 0.01] = -np.inf
        x[0, rs.integers(0, V, 5)] = np.nan
        xb = O.f32_to_bf16_bits(x)
        bits, _ = O.mask_f32_to_bits(synth.binary_masks(V, 2, V))
        mid = (np.arange(B) % 2).astype(np.int32)
        for kw in (dict(), dict(mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)):
            a = O.step(xb, rng_mode=O.RNG_PHILOX, seed=5, offset=3, **kw)
            b = O.step(xb, rng_mode=O.RNG_PHILOX, seed=5, offset=3, contract="hw", want_edge=True, **kw)
            assert np.abs(a[0] - b[0]).max() < 1e-5 and np.abs(a[1] - b[1]).max() < 1e-5
            differ = a[2] != b[2]
            assert (b[3][differ] < 2.0 ** -16).all() and differ.sum() <= 1
