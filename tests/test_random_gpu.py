"""Randomised parity sweep of the fused step (HIP through the C ABI vs the oracle, bit for bit): odd vocabulary
sizes down to 1, padded and misaligned rows, every element type, mask kind, draw mode and mask hand-over form
(bit rows per particle, prepared masks, mask ids per logits row = shared rows reduced once)."""
import os

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu

V_POOL = [1, 2, 3, 5, 31, 32, 33, 127, 255, 1000, 4097, 8191, 16380, 16381, 32765, 32769, 50257, 65528, 65529, 70001,
          69633, 73729, 131071, 200003, 262144, 262145]  # (17 / 18 / 19 / 32 / 49 / 64 / 65 chunks: waves of the log-softmax kernel
                                                          # carry one to three chunks; 65 chunks take its three-launch form)


def _case(rng):
    V = int(rng.choice(V_POOL)) if rng.random() < 0.7 else int(rng.integers(1, 60000))
    dtype = str(rng.choice(["f32", "bf16", "f16"]))
    U = int(rng.integers(1, 24))
    budget = 3_000_000 // max(V, 1)
    N = int(np.clip(rng.integers(1, 3 * U + 2), 1, max(budget, 1)))
    if rng.random() < 0.25:
        N = int(min(max(budget, 1), rng.integers(300, 700)))  # enough particles for the persistent kernel's auto path
    pad = int(rng.integers(0, 10))
    off = int(rng.integers(0, 8))
    mask_kind = str(rng.choice(["none", "bits", "bits", "f32"]))
    rng_mode = str(rng.choice(["philox", "philox", "none", "noise"]))
    if rng_mode == "noise" and N * V > 400_000:
        rng_mode = "philox"
    form = str(rng.choice(["particle", "prepared", "by_row"]))
    scale = float(rng.choice([1.0, 1.0, 0.5, 1.7]))
    return dict(V=V, dtype=dtype, U=U, N=N, pad=pad, off=off, mask_kind=mask_kind, rng_mode=rng_mode, form=form,
                scale=scale, seed=int(rng.integers(0, 2**31)), K=int(rng.integers(1, 5)))


CASES = [_case(np.random.default_rng(1000 + i)) for i in range(int(os.environ.get("GLB_RANDOM_CASES", "96")))]


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"V{c['V']}-{c['dtype']}-N{c['N']}-{c['mask_kind']}-{c['rng_mode']}-{c['form']}")
def test_random_case(engine, oracle, c):
    O = oracle
    dev = engine.device
    V, U, N = c["V"], c["U"], c["N"]
    rs = np.random.default_rng(c["seed"])
    x = synth.logits(c["seed"] % 100000, U, V)
    x[rs.random((U, V)) < 0.01] = -np.inf  # a few -inf logits (banned upstream)
    if c["seed"] % 5 == 0:
        x[rs.random((U, V)) < 0.002] = np.nan  # NaN logits count as absent (term 0), never as a maximum
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[c["dtype"]]
    xt = torch.from_numpy(x).to(tdt)
    if c["dtype"] == "f32":
        x_np = x
    elif c["dtype"] == "bf16":
        x_np = xt.view(torch.int16).numpy().view(np.uint16)
    else:
        x_np = xt.numpy()
    ld = V + c["pad"]
    flat = torch.full((c["off"] + U * ld + 8,), 123.0, dtype=tdt)
    view = flat[c["off"]:c["off"] + U * ld].view(U, ld)
    view[:, :V] = xt
    x_d = flat.to(dev)[c["off"]:c["off"] + U * ld].view(U, ld)[:, :V]
    row_of = rs.integers(0, U, N).astype(np.int32) if not (N == U and rs.random() < 0.5) else None
    K = c["K"]
    kw_o, kw_g = {}, {}
    if c["mask_kind"] != "none":
        masks = synth.binary_masks(c["seed"] % 9999, K, V)
        if K > 1:
            masks[K - 1, :] = -np.inf
            masks[K - 1, rs.integers(0, V, max(1, V // 500))] = 0.0  # nearly everything forbidden: low allowed mass
        mid = rs.integers(0, K, N).astype(np.int32)
        by_row = c["form"] == "by_row"
        if by_row:  # the mask is a function of the row: ids handed over per row, shared rows reduced once
            mid_row = rs.integers(0, K, U).astype(np.int32)
            mid = mid_row[row_of] if row_of is not None else mid_row[:N]
            ids_g = dict(row_mask_id=torch.from_numpy(mid_row).to(dev))
        else:
            ids_g = dict(mask_id=torch.from_numpy(mid).to(dev))
        if c["mask_kind"] == "bits":
            bits, _ = O.mask_f32_to_bits(masks)
            kw_o = dict(mask_kind=O.MASK_BITS, mask=bits, mask_id=mid)
            bits_d = torch.from_numpy(bits.view(np.int32)).to(dev)
            if c["form"] == "prepared":
                kw_g = dict(mask=engine.prepare_masks(bits_d, V, tdt), **ids_g)
            else:
                kw_g = dict(mask_kind=1, mask=bits_d, **ids_g)
        else:
            mf = masks.copy()
            fin = np.isfinite(mf)
            mf[fin] = rs.standard_normal(int(fin.sum())).astype(np.float32) * 2.0  # additive, some positive
            kw_o = dict(mask_kind=O.MASK_F32, mask=mf, mask_id=mid)
            kw_g = dict(mask_kind=2, mask=torch.from_numpy(mf).to(dev), **ids_g)
    mode = {"none": O.RNG_NONE, "philox": O.RNG_PHILOX, "noise": O.RNG_NOISE}[c["rng_mode"]]
    if c["rng_mode"] == "noise":
        E, _ = O.mt_exponential(c["seed"] % 1000, N * V)
        E = E.reshape(N, V)
        kw_o["noise"] = E
        kw_g["noise"] = torch.from_numpy(E).to(dev)
    seed, offset, base = c["seed"], c["seed"] % 17, c["seed"] % 1000
    logZ_o, lse_o, tok_o = O.step(x_np, row_of=row_of, rng_mode=mode, seed=seed, offset=offset, particle_base=base,
                                  logit_scale=c["scale"], n_particles=N if row_of is None else None, **kw_o)
    call = lambda **kw: engine.step(x_d, vocab=V, row_of=None if row_of is None else torch.from_numpy(row_of).to(dev),
                                    rng_mode={"none": 0, "philox": 1, "noise": 2}[c["rng_mode"]], seed=seed,
                                    offset=offset, particle_base=base, logit_scale=c["scale"], **kw_g, **kw)
    def check(res, what):
        logZ, lse, tok = res
        torch.cuda.synchronize()
        got_logZ, got_lse = logZ.cpu().numpy(), lse.cpu().numpy()
        assert np.array_equal(got_lse.view(np.uint32), lse_o.view(np.uint32)), f"lse ({what})"
        assert np.array_equal(got_logZ.view(np.uint32), logZ_o.view(np.uint32)), f"logZ ({what})"
        if c["rng_mode"] != "none":
            assert np.array_equal(tok.cpu().numpy(), tok_o), f"token ({what})"

    check(call(), "auto")
    # ... and under the second arithmetic contract (GLB_STEP_HW_EXP, round 6: the hardware's v_exp_f32): the same case against
    # the oracle's exp2f restatement by tolerance - logZ / lse within 1e-4 of both restatements, NaN / infinity patterns equal,
    # tokens equal except where the draw lies within 2^-20 of a boundary of the inverse CDF (tests/test_step_hw_gpu.py)
    if c["rng_mode"] != "noise":
        from tests.test_step_hw_gpu import check_hw

        *want_hw, edge = O.step(x_np, row_of=row_of, rng_mode=mode, seed=seed, offset=offset, particle_base=base,
                                logit_scale=c["scale"], n_particles=N if row_of is None else None, contract="hw",
                                want_edge=True, **kw_o)
        res = call(contract="hw")
        check_hw(res if c["rng_mode"] != "none" else (res[0], res[1], None), want_hw, (logZ_o, lse_o, tok_o), edge, "hw contract")
    # the same rows through glb_log_softmax_rows (the one-launch kernel of independent waves; -inf and NaN logits, odd
    # sizes, misaligned and padded rows): lse and every finite log-probability bit for bit, NaN where the logit is NaN
    want, lse_w = O.log_softmax_rows(x_np, c["scale"])
    got, lse_g = engine.log_softmax_rows(x_d, vocab=V, logit_scale=c["scale"], want_lse=True)
    torch.cuda.synchronize()
    got, lse_g = got.cpu().numpy(), lse_g.cpu().numpy()
    assert np.array_equal(lse_g.view(np.uint32), lse_w.view(np.uint32)), "log_softmax lse"
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan), "log_softmax NaN pattern"
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan]), "log_softmax rows"
    if c["dtype"] != "f32":  # ... and in the logits' own 16-bit type: the float32 result rounded to nearest even
        got16 = engine.log_softmax_rows(x_d, vocab=V, logit_scale=c["scale"], out_dtype=tdt)
        torch.cuda.synchronize()
        want16 = O.round_rows_16(want, c["dtype"])
        g16 = got16.cpu().view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(g16[~nan], want16.view(np.uint16)[~nan]), "log_softmax rows in the logits' dtype"
        assert np.array_equal(np.isnan(got16.float().cpu().numpy()), nan), "log_softmax NaN pattern (16-bit)"


# ---- round 5: the attention of the padded batches and the trie selections, swept -----------------------------------------------
def _attn_case(rng):
    Dh = int(rng.choice([16, 32, 64, 128]))
    Hkv = int(rng.choice([1, 2, 3, 8]))
    G = int(rng.choice([1, 1, 2, 4]))
    Lq = int(rng.integers(1, 20))
    Lk = Lq + int(rng.choice([0, 0, 0, 1, 5, 13, 30]))  # (up to 32 keys: the (row, KV head) kernel; beyond: round 4's)
    return dict(Dh=Dh, Hkv=Hkv, H=Hkv * G, Lq=Lq, Lk=Lk, U=int(rng.integers(1, 40)), dtype=str(rng.choice(["f32", "bf16", "f16"])),
                seed=int(rng.integers(0, 2**31)), masked=bool(rng.random() < 0.7))


ATTN_CASES = [_attn_case(np.random.default_rng(7000 + i)) for i in range(max(8, int(os.environ.get("GLB_RANDOM_CASES", "96")) // 3))]


@pytest.mark.parametrize("c", ATTN_CASES, ids=lambda c: f"U{c['U']}-H{c['H']}/{c['Hkv']}-L{c['Lq']}/{c['Lk']}-D{c['Dh']}-{c['dtype']}-{'mask' if c['masked'] else 'causal'}")
def test_random_short_attention(engine, c):
    """glb_short_attention against scaled_dot_product_attention in float32 on random shapes: head widths, grouped heads, key
    ranges on both sides of the 16 / 32-key kernels, ragged padding masks with fully padded queries, causal without a mask."""
    dev = engine.device
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[c["dtype"]]
    U, H, Hkv, Lq, Lk, Dh = c["U"], c["H"], c["Hkv"], c["Lq"], c["Lk"], c["Dh"]
    g = torch.Generator(device=dev)
    g.manual_seed(c["seed"])
    proj = torch.randn((U, Lq, (H + 2 * Hkv) * Dh), device=dev, generator=g).to(tdt)
    q = proj[..., :H * Dh].view(U, Lq, H, Dh).transpose(1, 2)
    k_new = proj[..., H * Dh:(H + Hkv) * Dh].view(U, Lq, Hkv, Dh).transpose(1, 2)
    v_new = proj[..., (H + Hkv) * Dh:].view(U, Lq, Hkv, Dh).transpose(1, 2)
    P = Lk - Lq
    if P:
        k = torch.cat([torch.randn((U, Hkv, P, Dh), device=dev, generator=g).to(tdt), k_new], dim=2)
        v = torch.cat([torch.randn((U, Hkv, P, Dh), device=dev, generator=g).to(tdt), v_new], dim=2)
    else:
        k, v = k_new, v_new
    ar = torch.arange(Lk, device=dev)
    causal = ar[None, :] <= (torch.arange(Lq, device=dev)[:, None] + P)
    mask = None
    if c["masked"]:
        lens = torch.randint(1, Lq + 1, (U,), device=dev, generator=g)
        base = torch.randint(0, P + 1, (U,), device=dev, generator=g) if P else torch.zeros(U, dtype=torch.long, device=dev)
        key_ok = (ar[None, :] < base[:, None]) | ((ar[None, :] >= P) & (ar[None, :] < P + lens[:, None]))
        mask = (key_ok[:, None, None, :] & causal[None, None, :, :]).contiguous()
    scale = Dh ** -0.5
    out = engine.short_attention(q, k, v, mask, scale)
    torch.cuda.synchronize()
    ref_mask = mask if mask is not None else causal[None, None].expand(U, 1, Lq, Lk)
    G = H // Hkv
    want = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(G, 1), v.float().repeat_interleave(G, 1),
                                                            attn_mask=ref_mask, scale=scale).transpose(1, 2)
    live = ref_mask.any(-1)[:, 0]
    tol = 5e-5 if tdt == torch.float32 else (2e-2 if tdt == torch.bfloat16 else 3e-3)
    assert out.shape == (U, Lq, H, Dh) and out.dtype == tdt
    assert (out.float() - want)[live].abs().max().item() < tol
    assert not bool(out[~live].any())


TRIE_CASES = [dict(seed=9000 + i) for i in range(max(4, int(os.environ.get("GLB_RANDOM_CASES", "96")) // 16))]


@pytest.mark.parametrize("c", TRIE_CASES, ids=lambda c: f"trie{c['seed']}")
def test_random_trie_selections(engine, oracle, c):
    """Selections of trie nodes on random vocabularies and part sizes: one selection for every row (the sub-forest plan) and a
    selection per row (parts a row does not need are skipped) give the whole trie's values of those nodes, bit for bit, and
    the whole trie's values are the oracle's - through the gathered plan and through the sweep plan."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(c["seed"])
    n_words, alpha, max_len = int(rs.integers(50, 3000)), int(rs.integers(2, 9)), int(rs.integers(2, 8))
    words, seen = [], set()
    tries = 0
    while len(words) < n_words and tries < 50 * n_words:
        tries += 1
        w = bytes(rs.integers(97, 97 + alpha, int(rs.integers(1, max_len + 1))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    trie.PLAN_CAP = int(rs.choice([60, 250, 1000, 20000]))
    if trie.plan() is None:
        pytest.skip("a node with more children than a part holds: the level kernels serve this trie")
    dev = engine.device
    V, nn, B = len(words), len(trie), int(rs.integers(1, 30))
    w = rs.random((B, V)).astype(np.float32)
    wd = torch.from_numpy(w).to(dev)
    op = int(rs.integers(0, 2))
    full = engine.trie_rows(wd, trie.plan_device_arrays(), op, False)
    assert np.array_equal(full.cpu().numpy().view(np.uint32), oracle.trie_reduce(w, trie.flat(), op).view(np.uint32))
    # one selection for all rows, through masses_from_logits' sub-forest plan (weights in as "log-probabilities" of themselves)
    x = (rs.standard_normal((B, V)) * 2).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    _, lse, _ = engine.step(xd, vocab=V, rng_mode=0)
    trie.prune_selection = False
    rows = trie.masses_from_logits(xd, lse)
    trie.prune_selection = True
    sel = torch.from_numpy(rs.choice(nn, int(rs.integers(1, min(nn, 300) + 1)), replace=False).astype(np.int32)).to(dev)
    assert torch.equal(trie.masses_from_logits(xd, lse, nodes=sel), rows[:, sel.long()])
    # a selection per row
    K = int(rs.integers(1, 20))
    per = rs.integers(-1, nn, size=(B, K)).astype(np.int32)
    got = trie.masses_from_logits(xd, lse, nodes=torch.from_numpy(per).to(dev))
    idx = torch.from_numpy(np.where(per >= 0, per, 0).astype(np.int64)).to(dev)
    want = torch.where(torch.from_numpy(per >= 0).to(dev), torch.gather(rows, 1, idx), torch.zeros_like(got))
    assert torch.equal(got, want)
    # the sweep plan (persistent workgroups read the rows front to back) on a random cut: the same bits for all nodes, the
    # slots, the per-row selection
    sw = trie.plan_device_arrays(int(rs.choice([60, 250, 1000, 20000])), sweep=True)
    if sw is not None:
        assert torch.equal(engine.trie_rows(wd, sw, op, False), full)
        assert torch.equal(engine.trie_rows(xd, sw, 0, True, lse=lse, layout="slots")[:, sw["slot_of"].long()], rows)
        if sw["n_parts"] <= 62:
            assert torch.equal(engine.trie_rows(xd, sw, 0, True, lse=lse, nodes=torch.from_numpy(per).to(dev)), want)
