"""GPU parity of the fused particle step against the CPU oracle (bit-exact: GLB math)."""
import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _bits_dev(bits_np, dev):
    return torch.from_numpy(bits_np.view(np.int32)).to(dev)


CASES = [
    # (B, V, dtype, n_masks)
    (8, 1000, "f32", 2),
    (5, 4097, "f32", 1),
    (16, 50257, "f32", 2),
    (4, 32000, "bf16", 2),
    (4, 128256, "bf16", 2),
    (3, 50257, "f16", 2),
    (3, 777, "bf16", 3),
]


def _mk(oracle, B, V, dtype, seed):
    x = synth.logits(seed, B, V)
    if dtype == "f32":
        return x, torch.from_numpy(x)
    if dtype == "bf16":
        t = torch.from_numpy(x).to(torch.bfloat16)
        return t.view(torch.int16).numpy().view(np.uint16), t
    t = torch.from_numpy(x).to(torch.float16)
    return t.numpy(), t


@pytest.mark.parametrize("B,V,dtype,K", CASES)
@pytest.mark.parametrize("mask_kind", ["none", "bits", "f32"])
def test_step_philox_bit_exact(engine, oracle, B, V, dtype, K, mask_kind):
    O = oracle
    x_np, x_t = _mk(O, B, V, dtype, seed=V + B)
    x_d = x_t.to(engine.device)
    masks = synth.binary_masks(V, K, V)
    mid = (np.arange(B) % K).astype(np.int32)
    kw_o, kw_g = {}, {}
    if mask_kind == "bits":
        bits, _ = O.mask_f32_to_bits(masks)
        kw_o = dict(mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
        kw_g = dict(mask_kind=1, mask=_bits_dev(bits, engine.device), mask_id=torch.from_numpy(mid).to(engine.device))
    elif mask_kind == "f32":
        mf = masks.copy()
        mf[np.isfinite(mf)] = np.random.default_rng(3).standard_normal(np.isfinite(mf).sum()).astype(np.float32)
        kw_o = dict(mask_kind=O.MASK_F32, mask=mf, mask_id=mid)
        kw_g = dict(mask_kind=2, mask=torch.from_numpy(mf).to(engine.device), mask_id=torch.from_numpy(mid).to(engine.device))
    logZ_o, lse_o, tok_o = O.step(x_np, rng_mode=O.RNG_PHILOX, seed=1234, offset=7, particle_base=11, **kw_o)
    logZ, lse, tok = engine.step(x_d, rng_mode=1, seed=1234, offset=7, particle_base=11, **kw_g)
    torch.cuda.synchronize()
    assert np.array_equal(_np(tok), tok_o)
    assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))
    assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))


@pytest.mark.parametrize("B,V,dtype,K", CASES[:4])
def test_step_noise_bit_exact(engine, oracle, B, V, dtype, K):
    O = oracle
    x_np, x_t = _mk(O, B, V, dtype, seed=V)
    masks = synth.binary_masks(V, K, V)
    bits, _ = O.mask_f32_to_bits(masks)
    mid = (np.arange(B) % K).astype(np.int32)
    E, _ = O.mt_exponential(4321, B * V)
    E = E.reshape(B, V)
    logZ_o, lse_o, tok_o = O.step(x_np, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid, rng_mode=O.RNG_NOISE, noise=E)
    dev = engine.device
    logZ, lse, tok = engine.step(x_t.to(dev), mask_kind=1, mask=_bits_dev(bits, dev),
                                 mask_id=torch.from_numpy(mid).to(dev), rng_mode=2, noise=torch.from_numpy(E).to(dev))
    torch.cuda.synchronize()
    assert np.array_equal(_np(tok), tok_o)
    assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))


def test_step_row_fanout_and_misaligned_rows(engine, oracle):
    """dedup fan-out (row_of) + rows whose start is not 16-byte aligned (ld = V odd)."""
    O = oracle
    U, N, V = 7, 40, 50257
    x = synth.logits(5, U, V)
    row_of = (np.arange(N) * 3 % U).astype(np.int32)
    masks = synth.binary_masks(1, 2, V)
    bits, _ = O.mask_f32_to_bits(masks)
    mid = (np.arange(N) % 2).astype(np.int32)
    logZ_o, lse_o, tok_o = O.step(x, row_of=row_of, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid,
                                  rng_mode=O.RNG_PHILOX, seed=99, offset=1)
    dev = engine.device
    logZ, lse, tok = engine.step(torch.from_numpy(x).to(dev), row_of=torch.from_numpy(row_of).to(dev), mask_kind=1,
                                 mask=_bits_dev(bits, dev), mask_id=torch.from_numpy(mid).to(dev), rng_mode=1,
                                 seed=99, offset=1)
    torch.cuda.synchronize()
    assert np.array_equal(_np(tok), tok_o)
    assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))
    # tokens differ across particles sharing a row (independent draws) but lse is shared
    assert np.array_equal(_np(lse)[row_of == 0], np.full((row_of == 0).sum(), _np(lse)[row_of == 0][0]))


def test_log_softmax_rows(engine, oracle):
    O = oracle
    for (B, V) in [(4, 1000), (6, 50257)]:
        x = synth.logits(B, B, V)
        want, lse_o = O.log_softmax_rows(x)
        got, lse = engine.log_softmax_rows(torch.from_numpy(x).to(engine.device), want_lse=True)
        torch.cuda.synchronize()
        assert np.array_equal(_np(got).view(np.uint32), want.view(np.uint32))
        assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))
        ref = torch.log_softmax(torch.from_numpy(x), -1).numpy()
        assert np.abs(_np(got) - ref).max() < 1e-4  # north_star tolerance vs the reference op (cache.py:96)


V2_CASES = [
    # (n_particles, n_rows, V, dtype, form)   form 0: mask ids per particle, bit rows prepared per call; 21+: prepared masks
    (700, 300, 50257, "f32", 21),
    (1024, 1024, 50257, "f32", 0),
    (130, 40, 65001, "f16", 23),
    (513, 513, 50257, "f32", 21),
    (300, 300, 30001, "f32", 22),
    (300, 100, 128256, "bf16", 23),
    (260, 260, 65001, "f16", 22),
    (9, 9, 1000, "f32", 22),
    (1024, 1024, 50257, "f32", 21),
    (512, 512, 128256, "bf16", 21),  # BASELINE config 5 at its full size, bit for bit
    (2300, 500, 50257, "f32", 21),   # 9 rows per workgroup: one wave per row in the tail
    (700, 300, 16001, "f32", 24),
    (700, 300, 32000, "bf16", 24),
    (400, 150, 151936, "bf16", 25),
    (400, 150, 70001, "f32", 25),
    (600, 600, 128256, "bf16", 0),   # auto: persistent geometry 23
    (400, 400, 100003, "f16", 0),
]


@pytest.mark.parametrize("N,U,V,dtype,variant", V2_CASES)
@pytest.mark.parametrize("mask_kind", ["none", "bits"])
def test_large_populations_bit_exact(engine, oracle, N, U, V, dtype, variant, mask_kind):
    """Large populations (more waves than the chip holds at once), shared rows, unaligned rows, low-mass and empty
    masks: the chunked reduction + per-particle finish give the same bits as the oracle, whichever way the masks
    are handed over."""
    O = oracle
    x_np, x_t = _mk(O, U, V, dtype, seed=V + U)
    dev = engine.device
    row_of = (np.arange(N) * 7 % U).astype(np.int32)
    K = 5
    masks = synth.binary_masks(V + 1, K, V)
    masks[2, 50:] = -np.inf  # allowed set far below the row maximum for most rows: masked sum on its own scale
    masks[3, :] = -np.inf    # nothing allowed: logZ = -inf, token -1
    masks[4, :] = -np.inf    # a few hundred allowed tokens: mass around the 2^-7 switch of the masked-sum scale
    masks[4, 7::V // 300] = 0.0
    mid = (np.arange(N) % K).astype(np.int32)
    kw_o, kw_g = {}, {}
    if mask_kind == "bits":
        bits, _ = O.mask_f32_to_bits(masks)
        kw_o = dict(mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
        kw_g = dict(mask_kind=1, mask=_bits_dev(bits, dev), mask_id=torch.from_numpy(mid).to(dev))
    logZ_o, lse_o, tok_o = O.step(x_np, row_of=row_of, rng_mode=O.RNG_PHILOX, seed=77, offset=5, particle_base=3, **kw_o)
    logZ, lse, tok = engine.step(x_t.to(dev), row_of=torch.from_numpy(row_of).to(dev), rng_mode=1, seed=77, offset=5,
                                 particle_base=3, **kw_g)
    torch.cuda.synchronize()
    assert np.array_equal(_np(tok), tok_o)
    assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))
    assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))
    if mask_kind == "bits" and variant:  # the same masks prepared once (GLB_MASK_PREPARED)
        tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
        kw_p = dict(mask=engine.prepare_masks(kw_g["mask"], V, tdt), mask_id=kw_g["mask_id"])
        logZ1, lse1, tok1 = engine.step(x_t.to(dev), row_of=torch.from_numpy(row_of).to(dev), rng_mode=1, seed=77,
                                        offset=5, particle_base=3, **kw_p)
        assert torch.equal(tok, tok1) and torch.equal(logZ, logZ1) and torch.equal(lse, lse1)


@pytest.mark.parametrize("B,V,dtype", [(5, 70001, "f32"), (3, 262144, "bf16"), (4, 151936, "f16"), (6, 1000, "f32")])
@pytest.mark.parametrize("mask_kind", ["none", "bits"])
def test_streaming_fallback_bit_exact(engine, oracle, B, V, dtype, mask_kind):
    """Very long rows (up to 262144 columns: more than 64 chunks per row, the finish kernel's second sweep) and short
    ones: same bits as the oracle."""
    O = oracle
    x_np, x_t = _mk(O, B, V, dtype, seed=V)
    dev = engine.device
    masks = synth.binary_masks(V, 2, V)
    mid = (np.arange(B) % 2).astype(np.int32)
    kw_o, kw_g = {}, {}
    if mask_kind == "bits":
        bits, _ = O.mask_f32_to_bits(masks)
        kw_o = dict(mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
        kw_g = dict(mask_kind=1, mask=_bits_dev(bits, dev), mask_id=torch.from_numpy(mid).to(dev))
    logZ_o, lse_o, tok_o = O.step(x_np, rng_mode=O.RNG_PHILOX, seed=5, offset=2, **kw_o)
    logZ, lse, tok = engine.step(x_t.to(dev), rng_mode=1, seed=5, offset=2, **kw_g)
    torch.cuda.synchronize()
    assert np.array_equal(_np(tok), tok_o)
    assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))
    assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))
    if dtype == "f32":
        want, _ = O.log_softmax_rows(x_np)
        got = engine.log_softmax_rows(x_t.to(dev))
        assert np.array_equal(_np(got).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("variant,N,U,V,dtype", [(21, 600, 200, 50257, "f32"), (-1, 40, 20, 50257, "f32"),
                                                  (23, 400, 100, 128256, "bf16"), (99, 6, 6, 9000, "f16")])
def test_logit_scale_and_stats_mode(engine, oracle, variant, N, U, V, dtype):
    """Temperature scaling (base.py:136-141: logits / T before the softmax) and the statistics-only mode, for launches
    on both sides of the four-waves-per-chunk threshold."""
    O = oracle
    x_np, x_t = _mk(O, U, V, dtype, seed=V + 7)
    dev = engine.device
    row_of = (np.arange(N) * 5 % U).astype(np.int32)
    masks = synth.binary_masks(V + 3, 2, V)
    bits, _ = O.mask_f32_to_bits(masks)
    mid = (np.arange(N) % 2).astype(np.int32)
    kw_o = dict(mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
    kw_g = dict(mask_kind=1, mask=_bits_dev(bits, dev), mask_id=torch.from_numpy(mid).to(dev))
    rd = torch.from_numpy(row_of).to(dev)
    for scale in (1.0, 0.7, 1.9):
        logZ_o, lse_o, tok_o = O.step(x_np, row_of=row_of, rng_mode=O.RNG_PHILOX, seed=9, offset=1, logit_scale=scale, **kw_o)
        logZ, lse, tok = engine.step(x_t.to(dev), row_of=rd, rng_mode=1, seed=9, offset=1, logit_scale=scale,
                                     **kw_g)
        assert np.array_equal(_np(tok), tok_o)
        assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))
        assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))
        logZ, lse, tok = engine.step(x_t.to(dev), row_of=rd, rng_mode=0, logit_scale=scale, **kw_g)
        assert tok is None
        assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))
        assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))


def test_many_rows_per_workgroup(engine, oracle):
    """20 000 particles over 300 shared rows with low-mass and empty masks (mask ids per particle: every particle is
    its own reduction unit)."""
    O = oracle
    N, U, V = 20000, 300, 1000
    x_np, x_t = _mk(O, U, V, "f32", seed=99)
    dev = engine.device
    row_of = (np.arange(N) * 11 % U).astype(np.int32)
    masks = synth.binary_masks(V + 5, 3, V)
    masks[1, 7:] = -np.inf   # low mass: own-scale redo in the tail
    masks[2, :] = -np.inf    # nothing allowed
    bits, _ = O.mask_f32_to_bits(masks)
    mid = (np.arange(N) % 3).astype(np.int32)
    logZ_o, lse_o, tok_o = O.step(x_np, row_of=row_of, rng_mode=O.RNG_PHILOX, seed=4, offset=2, mask_kind=O.MASK_BITS,
                                  mask=bits, mask_id=mid)
    logZ, lse, tok = engine.step(x_t.to(dev), row_of=torch.from_numpy(row_of).to(dev), rng_mode=1, seed=4, offset=2,
                                 mask_kind=1, mask=_bits_dev(bits, dev), mask_id=torch.from_numpy(mid).to(dev))
    assert np.array_equal(_np(tok), tok_o)
    assert np.array_equal(_np(logZ).view(np.uint32), logZ_o.view(np.uint32))
    assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))


@pytest.mark.parametrize("noise_from", ["device", "host"])
def test_parity_mode_at_the_headline_size(engine, oracle, noise_from):
    """1024 x 50257 fp32 in parity RNG mode against torch-CPU's log_softmax + mask + logsumexp + multinomial
    (tests/golden/ref_round2.npz, parity1024::*): every sampled id identical, logZ within 1e-4, and the kernel's own
    report of how close each draw was to a tie agrees with torch's race.  The noise comes from the DEVICE's MT19937
    stream (DeviceRng(2024, V).rows(1024): what the product runs - the whole chain seed -> jump-ahead -> exponentials ->
    race on the GPU box against a torch-made fixture), and once from the serial host stream (the form the reference runs)."""
    import os

    from genlm_backend_amd.engine import DeviceRng, HostRng

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_round2.npz"))
    B, V = 1024, 50257
    dev = engine.device
    x = synth.logits(21, B, V)
    masks = synth.binary_masks(21, 2, V)
    bits, _ = oracle.mask_f32_to_bits(masks)
    mid = (np.arange(B) % 2).astype(np.int32)
    if noise_from == "device":
        noise = DeviceRng(engine, 2024, V).rows(B)
    else:
        noise = HostRng(2024).exponential(B * V).view(B, V).to(dev)
    margin = torch.empty(B, device=dev)
    logZ, lse, tok = engine.step(torch.from_numpy(x).to(dev), mask_kind=1, mask=_bits_dev(bits, dev),
                                 mask_id=torch.from_numpy(mid).to(dev), rng_mode=2, noise=noise, out_margin=margin)
    torch.cuda.synchronize()
    assert np.array_equal(_np(tok), gold["parity1024::token"])
    assert np.abs(_np(logZ) - gold["parity1024::logZ"]).max() < 1e-4
    assert gold["parity1024::margin"].min() > 1e-4  # none of the golden draws was a near tie ...
    assert np.abs(_np(margin) - gold["parity1024::margin"]).max() < 1e-3  # ... and the kernel reports the same margins
    # first rows against the oracle, margins included, bit for bit
    z_o, l_o, t_o, m_o = oracle.step(x[:8], mask_kind=oracle.MASK_BITS, mask=bits, mask_id=mid[:8], rng_mode=oracle.RNG_NOISE,
                                     noise=noise[:8].cpu().numpy(), want_margin=True)
    assert np.array_equal(_np(tok)[:8], t_o) and np.array_equal(_np(margin)[:8].view(np.uint32), m_o.view(np.uint32))


def test_parity_and_float_masks_at_llama_width_fp32(engine, oracle):
    """fp32 rows of 128256 columns with parity-noise draws and additive float masks (round 1 refused both above 65528
    columns): bit-identical to the oracle."""
    O = oracle
    N, V = 6, 128256
    x_np, x_t = _mk(O, N, V, "f32", seed=31)
    dev = engine.device
    rs = np.random.default_rng(2)
    mf = synth.binary_masks(5, 2, V)
    fin = np.isfinite(mf)
    mf[fin] = rs.standard_normal(int(fin.sum())).astype(np.float32)
    mid = (np.arange(N) % 2).astype(np.int32)
    E, _ = O.mt_exponential(5, N * V)
    E = E.reshape(N, V)
    z_o, l_o, t_o = O.step(x_np, mask_kind=O.MASK_F32, mask=mf, mask_id=mid, rng_mode=O.RNG_NOISE, noise=E)
    z, l, t = engine.step(x_t.to(dev), mask_kind=2, mask=torch.from_numpy(mf).to(dev), mask_id=torch.from_numpy(mid).to(dev),
                          rng_mode=2, noise=torch.from_numpy(E).to(dev))
    assert np.array_equal(_np(t), t_o) and np.array_equal(_np(z).view(np.uint32), z_o.view(np.uint32))
    assert np.array_equal(_np(l).view(np.uint32), l_o.view(np.uint32))


@pytest.mark.parametrize("B,V,dtype,scale", [(160, 50257, "f32", 1.0), (130, 128256, "bf16", 0.7), (128, 4099, "f16", 1.0),
                                             (200, 70001, "f32", 1.3), (130, 270001, "bf16", 1.0),
                                             (1024, 50257, "f32", 1.0), (512, 128256, "bf16", 1.0)])  # the headline sizes
def test_log_softmax_rows_single_launch_path(engine, oracle, B, V, dtype, scale):
    """Rows of up to 64 chunks on the engine's workspace take the one-launch kernel of independent waves (each logit
    read once and kept in registers until its log-probability is written); longer rows and workspaces without tags the
    workgroup-per-row kernel (128 rows and more) or three launches: same bits as the oracle, all of them."""
    O = oracle
    x_np, x_t = _mk(O, B, V, dtype, seed=V + B)
    dev = engine.device
    ld = V + 3
    buf = torch.zeros((B, ld), dtype=x_t.dtype)
    buf[:, :V] = x_t
    want, lse_o = O.log_softmax_rows(x_np, scale)
    got, lse = engine.log_softmax_rows(buf.to(dev)[:, :V], vocab=V, logit_scale=scale, want_lse=True)
    torch.cuda.synchronize()
    assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))
    assert np.array_equal(_np(got).view(np.uint32), want.view(np.uint32))
    few, lse_few = engine.log_softmax_rows(buf.to(dev)[:5, :V], vocab=V, logit_scale=scale, want_lse=True)
    assert torch.equal(few, got[:5]) and torch.equal(lse_few, lse[:5])
    # a workspace the library has not initialised: the workgroup-per-row kernel / the three launches, same bits
    ws = torch.empty(engine.lib.glb_log_softmax_workspace_bytes(B, V) + 64, dtype=torch.uint8, device=dev)
    ws = ws[(-ws.data_ptr()) % 32:]
    got2, lse2 = engine.log_softmax_rows(buf.to(dev)[:, :V], vocab=V, logit_scale=scale, want_lse=True, workspace=ws)
    few2, lse_few2 = engine.log_softmax_rows(buf.to(dev)[:5, :V], vocab=V, logit_scale=scale, want_lse=True, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(got2, got) and torch.equal(lse2, lse) and torch.equal(few2, few) and torch.equal(lse_few2, lse_few)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_peaked_rows_with_the_likely_token_forbidden(engine, oracle, dtype):
    """Constrained decoding's everyday case: the model is sure of a token the mask forbids.  The allowed mass of that
    chunk is then far below its largest term - down to underflowing to zero on the chunk's scale - and the chunk sums
    its allowed values again on their own scale (per chunk, from registers); logZ must still match a float64
    log-softmax, for every mask hand-over form and shared rows."""
    O = oracle
    dev = engine.device
    V, U = 50257, 160
    rs = np.random.default_rng(5)
    x = (rs.standard_normal((U, V)) * 2).astype(np.float32)
    hot = rs.integers(0, V, U)
    gap = rs.choice([12.0, 25.0, 40.0, 70.0], U).astype(np.float32)  # 40 and 70 nats: terms below 2^-36 of the maximum
    x[np.arange(U), hot] += gap
    masks = np.zeros((U, V), np.float32)
    masks[np.arange(U), hot] = -np.inf                       # forbid exactly the likely token
    masks[:8, :] = -np.inf
    masks[np.arange(8), (hot[:8] + 1) % V] = 0.0             # a few rows allow one single (unlikely) token
    bits, _ = O.mask_f32_to_bits(masks)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16}[dtype]
    xt = torch.from_numpy(x).to(tdt)
    x_np = x if dtype == "f32" else xt.view(torch.int16).numpy().view(np.uint16)
    N = 2 * U
    row_of = np.concatenate([np.arange(U), rs.integers(0, U, N - U)]).astype(np.int32)
    logZ_o, lse_o, tok_o = O.step(x_np, row_of=row_of, mask_kind=O.MASK_BITS, mask=bits, mask_id=row_of,
                                  rng_mode=O.RNG_PHILOX, seed=9, offset=2)
    x_d, ro_d = xt.to(dev), torch.from_numpy(row_of).to(dev)
    bits_d = torch.from_numpy(bits.view(np.int32)).to(dev)
    rmid = torch.arange(U, dtype=torch.int32, device=dev)
    calls = {
        "per particle": dict(mask_kind=1, mask=bits_d, mask_id=ro_d),
        "per row": dict(mask_kind=1, mask=bits_d, row_mask_id=rmid),
        "per row, prepared": dict(mask=engine.prepare_masks(bits_d, V, tdt), row_mask_id=rmid),
    }
    for what, kw in calls.items():
        logZ, lse, tok = engine.step(x_d, row_of=ro_d, rng_mode=1, seed=9, offset=2, **kw)
        torch.cuda.synchronize()
        assert np.array_equal(logZ.cpu().numpy().view(np.uint32), logZ_o.view(np.uint32)), what
        assert np.array_equal(lse.cpu().numpy().view(np.uint32), lse_o.view(np.uint32)), what
        assert np.array_equal(tok.cpu().numpy(), tok_o), what
    # against float64: masked log-softmax mass (the reference's logsumexp(log_softmax(x) + mask), README.md:84-85)
    xf = xt.to(torch.float64)[torch.from_numpy(row_of).long()]
    lp = torch.log_softmax(xf, -1) + torch.from_numpy(masks).to(torch.float64)[torch.from_numpy(row_of).long()]
    want = torch.logsumexp(lp, -1).numpy()
    assert np.all(np.isfinite(logZ_o)) and np.abs(logZ_o - want).max() < 2e-4 * max(1.0, np.abs(want).max() / 10), \
        np.abs(logZ_o - want).max()
    assert np.all(masks[row_of, tok_o] == 0.0)


def test_philox_draws_follow_the_masked_distribution(engine):
    """Statistics of the two-stage draw (chunk, then element inside the chunk in register order): token frequencies over
    many particles match the masked softmax - through the per-particle reload path (200 000 particles on ONE shared
    row) and through the draws made by the reducing waves (20 000 identical rows, one particle each)."""
    dev = engine.device
    V = 9000  # three chunks, the last one partial
    g = torch.Generator().manual_seed(11)
    x = torch.randn(V, generator=g) * 2.5
    maskf = torch.where(torch.rand(V, generator=g) < 0.3, float("-inf"), 0.0)
    p = torch.softmax((x + maskf).double(), 0).numpy()
    bits, _ = engine.mask_to_bits(maskf[None].to(dev))
    top = np.argsort(-p)[:60]

    def check(tok, what):
        tok = tok.cpu().numpy()
        n = len(tok)
        assert np.all(np.isfinite(maskf.numpy()[tok])), what  # only allowed tokens
        cnt = np.bincount(tok, minlength=V)
        z = (cnt[top] - n * p[top]) / np.sqrt(n * p[top] * (1 - p[top]))
        assert np.abs(z).max() < 5.0, (what, np.abs(z).max())
        # per chunk as well (stage 1 alone)
        for c in range(3):
            pc = p[c * 4096:(c + 1) * 4096].sum()
            zc = (cnt[c * 4096:(c + 1) * 4096].sum() - n * pc) / np.sqrt(n * pc * (1 - pc))
            assert abs(zc) < 5.0, (what, c, zc)

    n1 = 200_000
    _, _, tok = engine.step(x[None].to(dev), row_of=torch.zeros(n1, dtype=torch.int32, device=dev), mask_kind=1, mask=bits,
                            rng_mode=1, seed=77, offset=1)
    check(tok, "shared row")
    n2 = 20_000
    _, _, tok = engine.step(x[None].repeat(n2, 1).to(dev), mask_kind=1, mask=bits, rng_mode=1, seed=78, offset=1)
    check(tok, "draws by the reducing waves")


@pytest.mark.parametrize("B,V,dtype", [(1024, 50257, torch.float32), (512, 128256, torch.bfloat16)])
def test_full_size_properties(engine, B, V, dtype):
    """BASELINE configs 2 and 5 at full size through properties that need no oracle: complementary masks split the mass
    (exp(logZ_A) + exp(logZ_B) = 1), an all-allowing mask gives logZ = 0 exactly, rows can be permuted and the call split
    into shards (particle_base) without changing a bit, duplicate rows agree, every drawn token is allowed."""
    dev = engine.device
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.randn((B, V), device=dev, generator=g) * 3).to(dtype)
    x[5] = x[900 % B]  # a duplicate row
    allow = torch.rand((1, V), device=dev, generator=g) < 0.4
    maskf = torch.cat([torch.where(allow, 0.0, float("-inf")), torch.where(allow, float("-inf"), 0.0),
                       torch.zeros((1, V), device=dev)])
    bits, _ = engine.mask_to_bits(maskf)
    prep = engine.prepare_masks(bits, V, dtype)
    ids = lambda k: torch.full((B,), k, dtype=torch.int32, device=dev)
    zA, lse, tokA = engine.step(x, mask=prep, row_mask_id=ids(0), rng_mode=1, seed=5, offset=9)
    zB, _, tokB = engine.step(x, mask=prep, row_mask_id=ids(1), rng_mode=1, seed=5, offset=9)
    z1, lse1, _ = engine.step(x, mask=prep, row_mask_id=ids(2), rng_mode=1, seed=5, offset=9)
    torch.cuda.synchronize()
    assert torch.all(z1 == 0.0) and torch.equal(lse1, lse)
    assert (torch.logaddexp(zA.double(), zB.double())).abs().max().item() < 1e-5
    assert torch.equal(lse[5], lse[900 % B]) and torch.equal(zA[5], zA[900 % B])
    am = allow[0]
    assert bool(am[tokA.long()].all()) and not bool(am[tokB.long()].any())
    ref = torch.logsumexp(x.double(), -1)
    assert (lse.double() - ref).abs().max().item() < 1e-4
    # shards: two calls with particle_base == one call
    h = B // 2
    outs = [engine.step(x[s:s + h], mask=prep, row_mask_id=ids(0)[s:s + h], rng_mode=1, seed=5, offset=9, particle_base=s)
            for s in (0, h)]
    assert torch.equal(torch.cat([o[0] for o in outs]), zA) and torch.equal(torch.cat([o[2] for o in outs]), tokA)
    # permutation of the rows: statistics follow bit for bit
    perm = torch.randperm(B, device=dev, generator=g)
    zP, lseP, _ = engine.step(x[perm].contiguous(), mask=prep, row_mask_id=ids(0), rng_mode=0)
    assert torch.equal(zP, zA[perm]) and torch.equal(lseP, lse[perm])


def test_step_plan_equals_step(engine):
    """A prepared call (argument block filled once) gives what step() gives, for every offset it is run with."""
    dev = engine.device
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((96, 9000), device=dev, generator=g)
    bits, _ = engine.mask_to_bits(torch.where(torch.rand((2, 9000), device=dev, generator=g) < 0.3, float("-inf"), 0.0))
    mid = (torch.arange(96, device=dev) % 2).to(torch.int32)
    masks = engine.prepare_masks(bits, 9000)
    out = (torch.empty(96, device=dev), torch.empty(96, device=dev), torch.empty(96, dtype=torch.int32, device=dev))
    plan = engine.step_plan(x, mask=masks, row_mask_id=mid, rng_mode=1, seed=7, offset=0, out=out)
    for off in (0, 3, 11):
        want = engine.step(x, mask=masks, row_mask_id=mid, rng_mode=1, seed=7, offset=off)
        got = plan.run(offset=off)
        torch.cuda.synchronize()
        for a, b in zip(want, got):
            assert torch.equal(a, b)


def _first_use_rows(n, distinct, seed):
    """row ids numbered by first use (what glb_group_contexts' out_group_of looks like): row_of[p] <= p"""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, distinct, n)
    seen, out = {}, np.empty(n, np.int32)
    for i, g in enumerate(raw):
        out[i] = seen.setdefault(int(g), len(seen))
    return out, len(seen)


@pytest.mark.parametrize("N,distinct,V,dtype", [(1024, 4000, 50257, "f32"), (512, 3000, 128256, "bf16"), (700, 300, 50257, "f32"),
                                                (300, 64, 4099, "f16")])
def test_deduplicated_rows_at_headline_sizes_bit_exact(engine, oracle, N, distinct, V, dtype):
    """The SIS loop's hand-over at its real sizes: row ids numbered by first use (glb_group_contexts' out_group_of), masks
    per logits row, shared rows reduced once - and the same rows in reverse numbering: same bits as the oracle."""
    O = oracle
    row_of, U = _first_use_rows(N, distinct, seed=N + V)
    assert (row_of <= np.arange(N)).all()
    x_np, x_t = _mk(O, U, V, dtype, seed=V + U)
    dev = engine.device
    masks = synth.binary_masks(V + 2, 2, V)
    bits, _ = O.mask_f32_to_bits(masks)
    mid_row = (np.arange(U) % 2).astype(np.int32)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    prep = engine.prepare_masks(_bits_dev(bits, dev), V, tdt)
    for ro in (row_of, (U - 1 - row_of).astype(np.int32)):
        want = O.step(x_np, row_of=ro, mask_kind=O.MASK_BITS, mask=bits, mask_id=mid_row[ro], rng_mode=O.RNG_PHILOX,
                      seed=9, offset=4, particle_base=5)
        got = engine.step(x_t.to(dev), row_of=torch.from_numpy(ro).to(dev), mask=prep,
                          row_mask_id=torch.from_numpy(mid_row).to(dev), rng_mode=1, seed=9, offset=4, particle_base=5)
        torch.cuda.synchronize()
        for w, g in zip(want, got):
            assert np.array_equal(_np(g).view(np.uint32), w.view(np.uint32))
    engine.check()


@pytest.mark.parametrize("B,V,dtype,scale", [(160, 50257, "bf16", 1.0), (130, 128256, "bf16", 0.7), (128, 4099, "f16", 1.0),
                                             (9, 270001, "bf16", 1.0), (512, 128256, "bf16", 1.0), (3, 50257, "f16", 1.3)])
def test_log_softmax_rows_in_the_logits_dtype(engine, oracle, B, V, dtype, scale):
    """cache.py:96 keeps the model's dtype: `out_dtype` = the logits' type gives the float32 log-probabilities rounded to
    nearest even (bit for bit the oracle's float32 rows cast by torch), within one 16-bit ulp of torch.log_softmax on the
    16-bit tensor itself; every form (one launch of independent waves, three launches) the same bits."""
    O = oracle
    x_np, x_t = _mk(O, B, V, dtype, seed=V + B)
    dev = engine.device
    ld = V + 5
    buf = torch.zeros((B, ld), dtype=x_t.dtype)
    buf[:, :V] = x_t
    want32, lse_o = O.log_softmax_rows(x_np, scale)
    want = torch.from_numpy(want32).to(x_t.dtype)
    got, lse = engine.log_softmax_rows(buf.to(dev)[:, :V], vocab=V, logit_scale=scale, want_lse=True, out_dtype=x_t.dtype)
    torch.cuda.synchronize()
    assert got.dtype == x_t.dtype
    assert np.array_equal(_np(lse).view(np.uint32), lse_o.view(np.uint32))
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))
    want_o = O.round_rows_16(want32, dtype)  # the oracle's own statement of the rounding
    assert np.array_equal(got.cpu().view(torch.int16).numpy().view(np.uint16), want_o.view(np.uint16))
    out = torch.zeros((B, ld), dtype=x_t.dtype, device=dev)  # a padded output: pitch in elements of the output type
    engine.log_softmax_rows(buf.to(dev)[:, :V], vocab=V, logit_scale=scale, out=out[:, :V])
    assert torch.equal(out[:, :V], got) and not bool(out[:, V:].any())
    ws = torch.empty(engine.lib.glb_log_softmax_workspace_bytes(B, V) + 64, dtype=torch.uint8, device=dev)
    ws = ws[(-ws.data_ptr()) % 32:]
    got2 = engine.log_softmax_rows(buf.to(dev)[:, :V], vocab=V, logit_scale=scale, workspace=ws, out_dtype=x_t.dtype)
    assert torch.equal(got2, got)
    if scale == 1.0 and B <= 160:
        ref = torch.log_softmax(x_t[:, :V], -1)  # the reference's op on the 16-bit tensor (CPU)
        a, b = got.cpu().float(), ref.float()
        ulp = (2.0 ** -7 if dtype == "bf16" else 2.0 ** -10) * b.abs().clamp_min(1.0) * 2
        assert bool(((a - b).abs() <= ulp).all())
    with pytest.raises(TypeError):
        engine.log_softmax_rows(buf.to(dev)[:, :V], vocab=V, out_dtype=torch.float16 if dtype == "bf16" else torch.bfloat16)


def test_failed_one_launch_calls_are_reported(engine, oracle):
    """A wave that gives up waiting for its row's records (GLB_SPIN_NONE forces it) writes token -2 / NaN and counts in
    the workspace's error word: glb_workspace_check returns GLB_EHIP, `raise_if_failed` raises, and the next healthy
    call is bit-exact again."""
    from genlm_backend_amd._lib import GlbError

    O = oracle
    B, V = 1024, 50257
    x_np, x_t = _mk(O, B, V, "f32", seed=5)
    dev = engine.device
    x_d = x_t.to(dev)
    want = O.step(x_np, rng_mode=O.RNG_PHILOX, seed=3, offset=1)
    engine.check()
    try:
        engine.set_spin_limit(None)
        logZ, lse, tok = engine.step(x_d, rng_mode=1, seed=3, offset=1)
        t = _np(tok)
        assert (t == -2).any(), "no finishing wave had to wait: the failure path was not exercised"
        bad = t == -2
        assert np.isnan(_np(logZ)[bad]).all() and np.isnan(_np(lse)[bad]).all()
        # waves that did not have to wait still deliver the oracle's bits
        assert np.array_equal(t[~bad], want[2][~bad])
        assert int(engine.error_word().item()) == int(bad.sum())
        with pytest.raises(GlbError, match="gave up waiting"):
            engine.check()
        assert int(engine.error_word().item()) == 0  # cleared by the check
        engine.step(x_d, rng_mode=1, seed=3, offset=1)
        with pytest.raises(GlbError):
            engine.raise_if_failed(int(engine.error_word().item()))
        with pytest.raises(GlbError):
            engine.raise_if_failed(tokens=t)
        # log-softmax rows: NaN rows + the error word
        lp = engine.log_softmax_rows(x_d)
        torch.cuda.synchronize()
        if int(engine.error_word().item()):
            assert bool(torch.isnan(lp).any())
            with pytest.raises(GlbError):
                engine.check()
    finally:
        engine.set_spin_limit(0)
    engine.check()
    got = engine.step(x_d, rng_mode=1, seed=3, offset=1)
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        assert np.array_equal(_np(g).view(np.uint32), w.view(np.uint32))
    engine.check()


def test_particles_advance_leaves_failed_particles_untouched(engine, oracle):
    dev = engine.device
    n, cap = 6, 8
    ctx = torch.zeros((n, cap), dtype=torch.int32, device=dev)
    ln = torch.full((n,), 2, dtype=torch.int32, device=dev)
    act = torch.ones(n, dtype=torch.int32, device=dev)
    lw = torch.zeros(n, device=dev)
    logZ = torch.tensor([1.0, float("nan"), 2.0, 3.0, float("nan"), 4.0], device=dev)
    tok = torch.tensor([5, -2, -1, 7, -2, 9], dtype=torch.int32, device=dev)
    engine.particles_advance(ctx, ln, act, lw, logZ, tok, 7, cap)
    torch.cuda.synchronize()
    assert act.tolist() == [1, 1, 0, 0, 1, 1] and ln.tolist() == [3, 2, 2, 2, 2, 3]
    assert lw.tolist() == [1.0, 0.0, 2.0, 3.0, 0.0, 4.0]
    c2, l2, a2, w2 = (np.zeros((n, cap), np.int32), np.full(n, 2, np.int32), np.ones(n, np.int32), np.zeros(n, np.float32))
    oracle.particles_advance(c2, l2, a2, w2, _np(logZ), _np(tok), 7, cap)
    assert np.array_equal(c2, _np(ctx)) and np.array_equal(l2, _np(ln)) and np.array_equal(a2, _np(act))


@pytest.mark.parametrize("B,V,dtype", [(1024, 50257, "f32"), (512, 128256, "bf16"), (70, 4099, "f16")])
def test_one_bit_mask_per_particle_bit_exact(engine, oracle, B, V, dtype):
    """What a grammar gives (README.md:57-70 generalised): one bit mask PER PARTICLE, handed over raw
    (GLB_MASK_BITS, n_masks == n_particles, no ids) so the call transposes them itself (mask_prepare_kernel: a 64 x 64
    bit-matrix transpose per chunk) - at the headline sizes, same bits as the oracle; and the same masks prepared ahead."""
    O = oracle
    x_np, x_t = _mk(O, B, V, dtype, seed=V + B + 1)
    dev = engine.device
    rng = np.random.default_rng(B)
    masks = np.where(rng.random((B, V)) < 1 / 3, -np.inf, 0.0).astype(np.float32)
    masks[1, :] = -np.inf          # nothing allowed
    masks[2, :] = -np.inf
    masks[2, V - 1] = 0.0          # only the row's last token
    masks[3, :] = 0.0              # everything allowed
    bits, _ = O.mask_f32_to_bits(masks)
    want = O.step(x_np, mask_kind=O.MASK_BITS, mask=bits, mask_id=np.arange(B, dtype=np.int32), rng_mode=O.RNG_PHILOX, seed=31,
                  offset=2, particle_base=7)
    bits_d = _bits_dev(bits, dev)
    got = engine.step(x_t.to(dev), mask_kind=1, mask=bits_d, rng_mode=1, seed=31, offset=2, particle_base=7)
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        assert np.array_equal(_np(g).view(np.uint32), w.view(np.uint32))
    assert _np(got[2])[1] == -1 and _np(got[2])[2] == V - 1
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    got_p = engine.step(x_t.to(dev), mask=engine.prepare_masks(bits_d, V, tdt), rng_mode=1, seed=31, offset=2, particle_base=7)
    torch.cuda.synchronize()
    for a, b in zip(got, got_p):
        assert torch.equal(a, b)
    engine.check()


@pytest.mark.parametrize("B,V", [(520, 1), (520, 31), (520, 33), (520, 4096), (300, 4097), (200, 8191), (180, 12289)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_raw_masks_read_by_the_fused_launch_at_row_ends(engine, oracle, B, V, dtype):
    """Raw per-particle bit masks inside the ONE-launch form (more than 512 (unit, chunk) items: the stats waves read the
    caller's bit rows themselves, kMaskRaw) on rows that end inside a mask word, inside a 16-byte vector, on a chunk's
    last element and one past it: words past a row's end are never read (the index is clamped), their bits count as
    forbidden - the oracle's bits."""
    O = oracle
    x_np, x_t = _mk(O, B, V, dtype, seed=7 * V + B)
    dev = engine.device
    rng = np.random.default_rng(V)
    masks = np.where(rng.random((B, V)) < 0.4, -np.inf, 0.0).astype(np.float32)
    masks[1, :] = -np.inf
    masks[2, :] = -np.inf
    masks[2, V - 1] = 0.0
    masks[3, :] = 0.0
    bits, _ = O.mask_f32_to_bits(masks)
    want = O.step(x_np, mask_kind=O.MASK_BITS, mask=bits, mask_id=np.arange(B, dtype=np.int32), rng_mode=O.RNG_PHILOX, seed=3,
                  offset=1)
    # the bit rows sit at the very end of an allocation: a read past a row's last word would leave it
    pad = torch.zeros(bits.size + 4096, dtype=torch.int32, device=dev)
    bits_d = pad[pad.numel() - bits.size:].view(B, -1)
    bits_d.copy_(_bits_dev(bits, dev))
    got = engine.step(x_t.to(dev), mask_kind=1, mask=bits_d, rng_mode=1, seed=3, offset=1)
    torch.cuda.synchronize()
    for w, g, name in zip(want, got, ("logZ", "lse", "token")):
        assert np.array_equal(_np(g).view(np.uint32), w.view(np.uint32)), name
    assert _np(got[2])[1] == -1 and _np(got[2])[2] == V - 1
    engine.check()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_masks_prepared_again_only_where_they_changed(engine, oracle, dtype):
    """glb_mask_prepare_rows (round 5): of 96 per-particle masks prepared once, 17 change; preparing only those again gives
    the same bytes as preparing all of them, and the step on the updated masks holds the oracle's bits.  Indices outside the
    table are skipped."""
    dev = engine.device
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    B, V = 96, 50257
    x = synth.logits(41, B, V)
    x_t = torch.from_numpy(x).to(tdt)
    x_o = x if dtype == "f32" else x_t.view(torch.int16).numpy().view(np.uint16)
    bits0, _ = oracle.mask_f32_to_bits(synth.binary_masks(41, B, V))
    bits1 = bits0.copy()
    rs = np.random.default_rng(4)
    rows = rs.choice(B, 17, replace=False).astype(np.int32)
    bits1[rows] = oracle.mask_f32_to_bits(synth.binary_masks(42, 17, V))[0]
    b0 = torch.from_numpy(bits0.view(np.int32)).to(dev)
    b1 = torch.from_numpy(bits1.view(np.int32)).to(dev)
    prep = engine.prepare_masks(b0, V, tdt)
    rows_d = torch.from_numpy(np.concatenate([rows, [-1, B + 5]]).astype(np.int32)).to(dev)
    engine.update_prepared_masks(prep, b1, rows_d)
    fresh = engine.prepare_masks(b1, V, tdt)
    torch.cuda.synchronize()
    assert torch.equal(prep.blob, fresh.blob)
    want = oracle.step(x_o, mask_kind=oracle.MASK_BITS, mask=bits1, rng_mode=oracle.RNG_PHILOX, seed=5, offset=9)
    got = engine.step(x_t.to(dev), mask=prep, rng_mode=1, seed=5, offset=9)
    torch.cuda.synchronize()
    for w, g, name in zip(want, got, ("logZ", "lse", "token")):
        assert np.array_equal(g.cpu().numpy().view(np.uint32), w.view(np.uint32)), name
