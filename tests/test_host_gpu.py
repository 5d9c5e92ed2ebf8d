"""GPU parity of the bookkeeping kernels against the oracle (bit-exact integer work) and of the whole
backend against the reference's goldens with the model on the MI355X."""
import ast
import asyncio
import os
import sys

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden", "ref_hotpath_tiny.npz")
TOL = 1e-4


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("n,distinct", [(1, 1), (7, 3), (1024, 1), (1024, 37), (5000, 4000), (3000, 3000), (2000, 1),
                                        (9000, 2), (20000, 15000)])
def test_group_contexts(engine, oracle, n, distinct):
    ctxs = synth.contexts(n + distinct, n, distinct, lo=0, hi=18)
    tok, st, ln = oracle.ragged(ctxs)
    g_o, rep_o, ng_o = oracle.group_contexts(ctxs)
    g, rep, ng = engine.group_contexts(_dev(tok, engine.device), _dev(st, engine.device), _dev(ln, engine.device))
    torch.cuda.synchronize()
    assert int(ng.item()) == ng_o
    assert np.array_equal(g.cpu().numpy(), g_o)
    assert np.array_equal(rep.cpu().numpy()[:ng_o], rep_o)


def test_group_contexts_padded_matrix_rows(engine, oracle):
    """(tokens, starts, lengths) addressing of a padded [n, cap] particle matrix (DeviceSIS layout)."""
    rng = np.random.default_rng(0)
    n, cap = 300, 20
    mat = rng.integers(0, 5, size=(n, cap)).astype(np.int32)
    ln = rng.integers(1, 4, size=n).astype(np.int32)
    ctxs = [list(mat[i, :ln[i]]) for i in range(n)]
    g_o, rep_o, ng_o = oracle.group_contexts(ctxs)
    dev = engine.device
    g, rep, ng = engine.group_contexts(_dev(mat.reshape(-1), dev), _dev(np.arange(n, dtype=np.int64) * cap, dev), _dev(ln, dev))
    assert int(ng.item()) == ng_o and np.array_equal(g.cpu().numpy(), g_o)


def test_match_prefixes_and_gather(engine, oracle):
    rng = np.random.default_rng(1)
    pre = [list(rng.integers(0, 50, size=k)) for k in (3, 5, 8, 5)]
    pre[3] = pre[1][:4] + [49]
    ctxs = []
    for i in range(500):
        p = pre[i % 4] if i % 5 else []
        ctxs.append(list(p) + list(rng.integers(0, 50, size=rng.integers(0, 6))))
    ctxs[7] = list(pre[2])  # equal to a cached prefix: must NOT match it (proper prefix rule)
    dev = engine.device
    tok, st, ln = oracle.ragged(ctxs)
    ptok, pst, pln = oracle.ragged(pre)
    p_o, b_o = oracle.match_prefixes(ctxs, pre)
    t_d, s_d, l_d = _dev(tok, dev), _dev(st, dev), _dev(ln, dev)
    p, b = engine.match_prefixes(t_d, s_d, l_d, _dev(ptok, dev), _dev(pst, dev), _dev(pln, dev))
    assert np.array_equal(p.cpu().numpy(), p_o) and np.array_equal(b.cpu().numpy(), b_o)
    sel = np.array([i for i in range(500) if len(ctxs[i]) - b_o[i] > 0][::3], np.int32)
    l_max = max(len(ctxs[i]) - b_o[i] for i in sel)
    want = oracle.gather_padded(ctxs, sel, b_o, pad_id=777, p_max=8, l_max=l_max)
    got = engine.gather_padded(t_d, s_d, l_d, _dev(sel, dev), len(sel), b, 777, 8, l_max)
    for w, g in zip(want, got):
        assert np.array_equal(g.cpu().numpy(), w)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_kv_padded(engine, oracle, dtype):
    dev = engine.device
    H, D = 4, 16
    slabs = [torch.randn(H, P, D).to(dtype) for P in (3, 7, 5)]
    pref = np.array([0, -1, 2, 1, 1, 0], np.int32)
    nps = [s.view(torch.int16).numpy() if dtype == torch.bfloat16 else s.numpy() for s in slabs]
    want = oracle.gather_kv_padded(nps, pref, 7)
    d_slabs = [s.to(dev) for s in slabs]
    ptrs = torch.tensor([s.data_ptr() for s in d_slabs], dtype=torch.int64, device=dev)
    lens = torch.tensor([3, 7, 5], dtype=torch.int32, device=dev)
    got = engine.gather_kv_padded(ptrs, lens, _dev(pref, dev), H, D, 7, dtype)
    got_np = got.cpu().view(torch.int16).numpy() if dtype == torch.bfloat16 else got.cpu().numpy()
    assert np.array_equal(got_np, want)


def test_particles_advance_and_normalize(engine, oracle):
    rng = np.random.default_rng(2)
    n, cap = 777, 12
    ctx = rng.integers(0, 100, size=(n, cap)).astype(np.int32)
    ln = rng.integers(1, cap, size=n).astype(np.int32)
    act = (rng.random(n) < 0.8).astype(np.int32)
    lw = rng.standard_normal(n).astype(np.float32)
    logZ = rng.standard_normal(n).astype(np.float32)
    tok = rng.integers(-1, 5, size=n).astype(np.int32)
    dev = engine.device
    d = [_dev(a.copy(), dev) for a in (ctx, ln, act, lw)]
    engine.particles_advance(d[0], d[1], d[2], d[3], _dev(logZ, dev), _dev(tok, dev), 0, cap)
    oracle.particles_advance(ctx, ln, act, lw, logZ, tok, 0, cap)
    for a, b in zip(d, (ctx, ln, act, lw)):
        assert np.array_equal(a.cpu().numpy(), b)
    for m in (5, 1024, 4096, 100000):
        w = (rng.standard_normal(m) * 5).astype(np.float32)
        w[rng.integers(0, m)] = -np.inf
        p_o, s_o = oracle.normalize_weights(w)
        p, s = engine.normalize_weights(_dev(w, dev))
        assert np.array_equal(p.cpu().numpy().view(np.uint32), p_o.view(np.uint32))
        assert np.array_equal(s.cpu().numpy().view(np.uint32), s_o.view(np.uint32))


def test_mask_to_bits(engine, oracle):
    m = synth.binary_masks(4, 5, 50257)
    want, nb = oracle.mask_f32_to_bits(m)
    bits, flag = engine.mask_to_bits(torch.from_numpy(m))
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), want) and int(flag.item()) == 0
    m[2, 17] = -1.5
    _, flag = engine.mask_to_bits(torch.from_numpy(m))
    assert int(flag.item()) == 1


# ---- whole backend on the GPU against the reference's goldens -------------------------------------
class Tok:
    pad_token_id = None
    eos_token_id = 0


@pytest.fixture()
def llm(engine):
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM

    gold = np.load(G)
    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    m = AsyncAmdLM(model.to(engine.device), None, batch_size=64, timeout=0.02, engine=engine)
    m.tokenizer = Tok()
    return m, gold


def _strip(row):
    return [int(t) for t in row if t >= 0]


def test_backend_logprobs_and_kv_on_gpu(llm):
    m, gold = llm
    prompts = [_strip(r) for r in gold["lp_prompts"]]
    got = asyncio.run(m.batch_next_token_logprobs(prompts))
    assert got.is_cuda
    assert np.abs(got.cpu().numpy() - gold["lp_values"]).max() < TOL
    assert m.stats["unique"] == 4
    m.clear_cache()
    pre = [int(t) for t in gold["kv_prefix"]]
    m.cache_kv(pre)
    qs = [_strip(r) for r in gold["kv_queries"]]
    got = asyncio.run(m.batch_next_token_logprobs(qs)).cpu().numpy()
    assert np.abs(got - gold["kv_values"]).max() < TOL
    for p, want in zip(prompts, gold["lp_values"]):
        assert np.abs(m.next_token_logprobs_uncached(p).cpu().numpy() - want).max() < TOL


def test_backend_sis_on_gpu_matches_reference(llm):
    from genlm_backend_amd.sis import DeviceSIS, autobatched_sis

    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    want_ctx = [_strip(r) for r in gold["sis_contexts"]]
    for kw in (dict(), dict(use_prefix_kv=True), dict(use_particle_kv=True)):
        # prompt KV cached (cache_kv semantics) / device-resident per-particle KV: same tokens, same weights
        sis = DeviceSIS(m, 16, prompt, max_tokens=10, eos_id=0, seed=1234, rng="torch", **kw)
        sis.run()
        ctx, lw = sis.results()
        assert [list(map(int, c)) for c in ctx] == want_ctx
        assert np.abs(lw - gold["sis_log_weights"]).max() < TOL
    m.set_rng("torch", 1234)
    parts = asyncio.run(autobatched_sis(16, m, lambda c: 1 if len(c) >= 10 else 0, prompt, eos_id=0))
    assert [p.context for p in parts] == want_ctx
    m.set_rng("torch", 1234)  # the same loop with the step's coroutines run by AsyncAmdLM.gather (no Task per particle)
    parts = asyncio.run(autobatched_sis(16, m, lambda c: 1 if len(c) >= 10 else 0, prompt, eos_id=0, gather=m.gather))
    assert [p.context for p in parts] == want_ctx
    assert np.abs(np.asarray([p.log_weight for p in parts], np.float32) - gold["sis_log_weights"]).max() < TOL
    ids = asyncio.run(m.sample([int(t) for t in gold["sample_prompt"]], max_tokens=12, eos_token_ids=[0],
                               temperature=0.5, seed=80808))
    assert ids == [int(t) for t in gold["sample_ids"]]


def test_device_sis_per_particle_masks_follow_in_place_edits_on_gpu(llm):
    """`DeviceSIS(particle_masks=...)` on the HIP engine: the bit rows are brought into the kernels' layout once and only the
    rows `update_particle_masks` names are prepared again - but a caller that writes `sis.particle_masks` in place, or rebinds
    it (the pattern the raw hand-over supported), must never sample under the stale prepared form: the tensor's version /
    identity is checked every step.  Golden run first: every particle under the README's `valid` mask."""
    from genlm_backend_amd.sis import DeviceSIS

    m, gold = llm
    masks = torch.from_numpy(gold["sis_masks"])
    m.register_masks(masks)
    prompt = [int(t) for t in gold["sis_prompt"]]
    dev = m.device
    bits, _ = m.engine.mask_to_bits(masks.to(dev))
    pm = torch.cat([bits[:1].expand(16, -1), bits[1:2]]).contiguous()
    sis = DeviceSIS(m, 16, prompt, max_tokens=10, eos_id=0, seed=1234, rng="torch", particle_masks=pm)
    sis.run()
    ctx, lw = sis.results()
    assert [list(map(int, c)) for c in ctx] == [_strip(r) for r in gold["sis_contexts"]]
    assert np.abs(lw - gold["sis_log_weights"]).max() < TOL
    V = masks.shape[1]
    only = torch.full((17, V), float("-inf"), device=dev)
    for i in range(16):
        only[i, 3 + i] = 0.0
    only[16, 0] = 0.0
    pm2, _ = m.engine.mask_to_bits(only)
    other = torch.full((2, V), float("-inf"), device=dev)
    other[0, 40], other[1, 41] = 0.0, 0.0
    ob, _ = m.engine.mask_to_bits(other)
    sis = DeviceSIS(m, 16, prompt, max_tokens=4, eos_id=0, seed=1, particle_masks=pm2.clone())
    sis.step()
    sis.update_particle_masks(torch.tensor([2], dtype=torch.int32, device=dev), ob[:1])  # told: row 2 prepared again
    sis.step()
    sis.particle_masks[7] = ob[0]          # not told: written in place
    sis.step()
    fresh = sis.particle_masks.clone()
    fresh[9] = ob[1]
    sis.particle_masks = fresh             # not told: another tensor
    sis.step()
    ctx, _ = sis.results()
    want = [[3 + i] * 4 for i in range(16)]
    want[2], want[7], want[9] = [5, 40, 40, 40], [10, 10, 40, 40], [12, 12, 12, 41]
    assert [list(map(int, c)) for c in ctx] == want


def test_device_sis_per_particle_masks_raw_or_prepared_same_run_on_gpu(llm):
    """Round 6: a step hands the per-particle bit rows over RAW (the fused launch reads them itself) when the tensor is in a
    state not seen before or more than `pm_raw_above` of its rows changed since the prepared form was last brought up to date,
    and PREPARED otherwise (the rows named since are prepared again, all of them - the prepared form may lag several steps,
    until one passes in which no mask moved).
    Whatever the mix, the run is the run of the always-prepared form: same contexts, same weights (600 particles: more than
    512 (unit, chunk) items, so the raw steps do take the one-launch form)."""
    from genlm_backend_amd.sis import DeviceSIS

    m, gold = llm
    masks = torch.from_numpy(gold["sis_masks"])
    m.register_masks(masks)
    prompt = [int(t) for t in gold["sis_prompt"]]
    dev, V, N = m.device, masks.shape[1], 600
    g = torch.Generator(device="cpu")
    g.manual_seed(5)

    def rows(n):
        f = torch.where(torch.rand((n, V), generator=g) < 0.5, float("-inf"), 0.0)
        f[:, 1:4] = 0.0  # (something other than EOS is always allowed)
        return m.engine.mask_to_bits(f.to(dev))[0]

    eos_only, _ = m.engine.mask_to_bits(masks[1:2].to(dev))
    pm0 = torch.cat([rows(N), eos_only]).contiguous()
    script = [None, None, ("few", 40), ("many", 400), ("few", 30), None, ("all", N), ("few", 10)]
    edits = []
    for e in script:
        if e is None:
            edits.append(None)
        else:
            idx = torch.randperm(N, generator=g)[:e[1]].to(torch.int32)
            edits.append((idx.to(dev), rows(e[1])))
    runs = []
    for above in (None, 1.0):
        sis = DeviceSIS(m, N, prompt, max_tokens=len(script), eos_id=0, seed=77, particle_masks=pm0.clone())
        sis.pm_raw_above = above
        kinds = []
        for e in edits:
            if e is not None:
                sis.update_particle_masks(*e)
            before = sis.pm_raw_steps
            sis.step()
            kinds.append(sis.pm_raw_steps > before)
        ctx, lw = sis.results()
        runs.append(([list(map(int, c)) for c in ctx], lw, kinds))
    # first sight of the tensor raw, the second prepared; 40 of 600 changed: those rows prepared again; 400: raw (the prepared
    # form lags), 30 more: 430 behind - raw; nothing moved: the 430 prepared; all: raw; 10 more: raw (float32: above a quarter)
    assert runs[0][2] == [True, False, False, True, True, False, True, True]
    assert runs[1][2] == [False] * len(script)
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1])
    assert len({tuple(c) for c in runs[0][0]}) > 100  # (the masks do steer the particles apart)


def test_device_sis_philox_is_shard_invariant(llm):
    """Sharding the population (particle_base) does not change any particle's draws or weights."""
    from genlm_backend_amd.sis import DeviceSIS

    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    full = DeviceSIS(m, 32, prompt, max_tokens=6, eos_id=0, seed=99)
    full.run()
    c_full, w_full = full.results()
    halves = []
    for r in range(2):
        s = DeviceSIS(m, 16, prompt, max_tokens=6, eos_id=0, seed=99, rank=r, world=1)
        s.rank = r  # particle_base = rank * N
        s.run()
        halves.append(s.results())
    assert c_full == halves[0][0] + halves[1][0]
    # weights agree to rounding only: the PyTorch forward's GEMMs round differently for different batch
    # shapes (the fused kernel itself is bit-identical for identical logits, tests/test_step_gpu.py)
    assert np.abs(w_full - np.concatenate([halves[0][1], halves[1][1]])).max() < 1e-5


# ---- device-resident particle state: KV slabs, row gathers, resampling ------------------------------------------
@pytest.mark.parametrize("dtype,hd", [(torch.float32, 64), (torch.bfloat16, 64), (torch.float16, 20), (torch.float32, 6)])
def test_kv_slab_kernels(engine, dtype, hd):
    dev = engine.device
    g = torch.Generator(device="cpu")
    g.manual_seed(3)
    n, H, cap, u, lsrc = 37, 5, 19, 11, 9
    slab = torch.randn((n, H, cap, hd), generator=g).to(dtype).to(dev)
    want = slab.clone()
    # append: new rows come as a transposed (strided) view, like a projection output [n, 1, H, hd] -> [n, H, 1, hd]
    new = torch.randn((n, 1, H, hd), generator=g).to(dtype).to(dev).transpose(1, 2)
    pos = torch.randint(0, cap, (n,), generator=g).to(torch.int32).to(dev)
    engine.kv_append(slab, new, pos)
    want[torch.arange(n, device=dev), :, pos.long()] = new[:, :, 0]
    assert torch.equal(slab, want)
    # gather rows from a shorter source into two slabs at once; -1 rows stay
    srcs = [torch.randn((u, H, lsrc, hd), generator=g).to(dtype).to(dev) for _ in range(2)]
    dsts = [slab.clone(), slab.clone() + 1]
    wants = [d.clone() for d in dsts]
    row = torch.randint(-1, u, (n,), generator=g).to(torch.int32).to(dev)
    ln = torch.randint(0, lsrc + 3, (n,), generator=g).to(torch.int32).to(dev)
    engine.kv_gather_rows(srcs, dsts, row, ln)
    for s_, w_ in zip(srcs, wants):
        for i in range(n):
            if int(row[i]) >= 0:
                L = min(int(ln[i]), lsrc, cap)
                w_[i, :, :L] = s_[int(row[i]), :, :L]
    assert all(torch.equal(d, w) for d, w in zip(dsts, wants))


def test_gather_rows_i32(engine):
    dev = engine.device
    src = torch.arange(50 * 21, dtype=torch.int32, device=dev).view(50, 21)
    row = torch.tensor([3, 3, 49, 0, 17], dtype=torch.int32, device=dev)
    assert torch.equal(engine.gather_rows_i32(src, row), src[row.long()])
    assert torch.equal(engine.gather_rows_i32(src[:, :7], row), src[row.long(), :7])


@pytest.mark.parametrize("n", [1, 2, 65, 1024, 4096, 100000])
def test_resample_systematic(engine, oracle, n):
    rs = np.random.default_rng(n)
    lw = (rs.standard_normal(n) * 4).astype(np.float32)
    lw[rs.random(n) < 0.05] = -np.inf
    lw[0] = 1.0
    for seed, off in ((7, 0), (7, 3), (123456789012, 9)):
        anc_o, lse_o = oracle.resample_systematic(lw, seed, off)
        anc, lse = engine.resample_systematic(_dev(lw, engine.device), seed, off)
        assert np.array_equal(anc.cpu().numpy(), anc_o)
        assert np.float32(lse_o).view(np.uint32) == lse.cpu().numpy().view(np.uint32)[0]


def test_device_sis_resampling_on_gpu(llm):
    """Resampling after every step on the device: slab KV (rows follow their ancestors through one gather launch)
    and plain re-encoding agree token for token; ragged prompts."""
    from genlm_backend_amd.sis import DeviceSIS

    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    p = [int(t) for t in gold["sis_prompt"]]
    prompts = [p, p[:5], p[2:], p] * 8
    runs = []
    for kw in (dict(use_particle_kv=True), dict()):
        s = DeviceSIS(m, 32, prompts, max_tokens=6, eos_id=0, seed=5, resample_ess=1.0, **kw)
        s.run()
        assert s.n_resamples >= 2
        runs.append(s.results())
    assert runs[0][0] == runs[1][0]
    assert np.abs(runs[0][1] - runs[1][1]).max() < 1e-4


# ---- round-2 reference goldens on the GPU: trie masses, config 3, Llama-shaped model ------------------------------
G2 = os.path.join(os.path.dirname(__file__), "golden", "ref_round2.npz")


@pytest.mark.parametrize("tag", ["kat", "syn"])
def test_trie_masses_on_gpu(engine, oracle, tag):
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    gold = np.load(G2)
    words = bytes(gold[f"trie::{tag}::words"]).split(b"\x00")
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    ws = gold[f"trie::{tag}::ws"]
    for op, key, fn in ((0, "sum", trie.batch_weight_sum), (1, "max", trie.batch_weight_max)):
        got = fn(torch.from_numpy(ws))
        assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(ws, trie.flat(), op).view(np.uint32))
        assert np.abs(got - gold[f"trie::{tag}::{key}"]).max() < 1e-6  # the reference's own numbers
    assert np.array_equal(trie.weight_sum(ws[0]), trie.batch_weight_sum(ws)[0])
    lp = np.log(np.maximum(ws, 1e-30)).astype(np.float32)
    got = trie.batch_weight_sum_device(torch.from_numpy(lp), from_logprobs=True).cpu().numpy()
    assert np.abs(got - gold[f"trie::{tag}::sum"]).max() < 1e-5


def test_trie_masses_large_vocabulary(engine, oracle):
    """gpt2-sized vocabulary of synthetic byte strings: kernel == oracle bit for bit, root == row sum - 70 rows (node-major
    values, a row count that is not a multiple of the transpose tile), 40 and 8 rows for the maximum (node-major and
    row-major kernels)."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(0)
    words, seen = [], set()
    while len(words) < 50257:
        w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    ws = rs.random((70, len(words))).astype(np.float32)
    ws /= ws.sum(-1, keepdims=True)
    got = trie.batch_weight_sum(torch.from_numpy(ws))
    assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(ws, trie.flat(), 0).view(np.uint32))
    assert np.abs(got[:, trie.root] - 1.0).max() < 1e-6
    assert np.array_equal(trie.batch_weight_sum(torch.from_numpy(ws[:5])), got[:5])  # row-major kernels: same bits
    for nb in (40, 8):
        gm = trie.batch_weight_max(torch.from_numpy(ws[:nb]))
        assert np.array_equal(gm.view(np.uint32), oracle.trie_reduce(ws[:nb], trie.flat(), 1).view(np.uint32))


@pytest.mark.parametrize("K,mode", [(1, "prefix"), (8, "plain"), (8, "pkv"), (64, "prefix"), (64, "pkv")])
def test_config3_on_gpu_matches_reference(llm, K, mode):
    """BASELINE config 3: K distinct ragged shared prompts, dedup + prefix / per-particle KV, tokens == reference."""
    from genlm_backend_amd.sis import DeviceSIS

    m, _ = llm
    gold = np.load(G2)
    m.register_masks(torch.from_numpy(gold["c3::masks"]))
    prompts = [_strip(r) for r in gold[f"c3::K{K}::prompts"]]
    per = [prompts[i % K] for i in range(64)]
    sis = DeviceSIS(m, 64, per, max_tokens=6, eos_id=0, seed=4321 + K, rng="torch", use_prefix_kv=mode == "prefix",
                    use_particle_kv=mode == "pkv")
    sis.run()
    ctx, lw = sis.results()
    assert [list(map(int, c)) for c in ctx] == [_strip(r) for r in gold[f"c3::K{K}::contexts"]]
    assert np.abs(lw - gold[f"c3::K{K}::log_weights"]).max() < TOL


def test_llama_shaped_model_on_gpu_matches_reference(engine):
    from transformers import LlamaConfig, LlamaForCausalLM

    from genlm_backend_amd.llm import AsyncAmdLM
    from genlm_backend_amd.sis import DeviceSIS

    gold = np.load(G2)
    cfg = ast.literal_eval(bytes(gold["llama::config_json"]).decode())
    model = LlamaForCausalLM(LlamaConfig(**cfg)).eval()
    model.load_state_dict({k[len("llama::w::"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("llama::w::")})
    m = AsyncAmdLM(model.to(engine.device), None, batch_size=64, engine=engine)
    m.tokenizer = Tok()
    prompts = [_strip(r) for r in gold["llama::lp_prompts"]]
    got = asyncio.run(m.batch_next_token_logprobs(prompts)).cpu().numpy()
    assert np.abs(got - gold["llama::lp_values"]).max() < TOL
    m.register_masks(torch.from_numpy(gold["llama::sis_masks"]))
    p3 = [_strip(r) for r in gold["llama::sis_prompts"]]
    per = [p3[i % 3] for i in range(24)]
    for kw in (dict(), dict(use_prefix_kv=True), dict(use_particle_kv=True)):
        sis = DeviceSIS(m, 24, per, max_tokens=6, eos_id=0, seed=999, rng="torch", **kw)
        sis.run()
        ctx, lw = sis.results()
        assert [list(map(int, c)) for c in ctx] == [_strip(r) for r in gold["llama::sis_contexts"]]
        assert np.abs(lw - gold["llama::sis_log_weights"]).max() < TOL
    # the same loop through the stateless API with KV rows that follow the contexts (RoPE, grouped-query KV slabs)
    model2 = LlamaForCausalLM(LlamaConfig(**cfg)).eval()
    model2.load_state_dict(model.state_dict())
    a = AsyncAmdLM(model2.to(engine.device), None, batch_size=64, engine=engine, auto_kv_rows=30, auto_kv_cap=24)
    a.tokenizer = Tok()
    a.register_masks(torch.from_numpy(gold["llama::sis_masks"]))
    a.set_rng("torch", 999)
    gen, lw, active = [[] for _ in range(24)], np.zeros(24, np.float64), [True] * 24
    while any(active):
        idx = [i for i in range(24) if active[i]]
        logZ, tok = a.batch_next_token_step_sync([per[i] + gen[i] for i in idx], [1 if len(gen[i]) >= 6 else 0 for i in idx])
        for i, z, t in zip(idx, logZ, tok):
            lw[i] += z
            if t == 0 or t < 0:
                active[i] = False
            else:
                gen[i].append(int(t))
    assert gen == [_strip(r) for r in gold["llama::sis_contexts"]]
    assert np.abs(lw.astype(np.float32) - gold["llama::sis_log_weights"]).max() < TOL
    assert a._auto_kv.stats["encoded_rows"] == 3 and a._auto_kv.stats["in_place_calls"] >= 1


def test_refresh_weights_after_a_write_the_version_counters_do_not_see(engine):
    """the shadow's derived [q; k; v] / [gate; up] weights follow `param.mul_()` by themselves (Tensor._version)
    but not `param.data.mul_()` (EMA / weight-merging code): `AsyncAmdLM.refresh_weights()` drops the derived copies, the
    captured graphs and the caches, and the backend then computes what the caller's model computes."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from genlm_backend_amd.llm import AsyncAmdLM

    gold = np.load(G2)
    cfg = ast.literal_eval(bytes(gold["llama::config_json"]).decode())
    model = LlamaForCausalLM(LlamaConfig(**cfg)).eval()
    model.load_state_dict({k[len("llama::w::"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("llama::w::")})
    model = model.to(engine.device)
    m = AsyncAmdLM(model, None, batch_size=64, engine=engine)
    m.tokenizer = Tok()
    prompts = [_strip(r) for r in gold["llama::lp_prompts"]]
    ids = torch.tensor([prompts[0]], device=engine.device)

    def own():
        with torch.no_grad():
            return torch.log_softmax(model(ids).logits[0, -1].float(), -1)

    before = m.batch_next_token_logprobs_sync(prompts[:1])[0]
    assert (before - own()).abs().max().item() < TOL
    with torch.no_grad():  # a write PyTorch counts: followed without being told
        model.model.layers[0].self_attn.q_proj.weight.mul_(1.5)
    m.clear_cache()
    assert (m.batch_next_token_logprobs_sync(prompts[:1])[0] - own()).abs().max().item() < TOL
    model.model.layers[0].self_attn.k_proj.weight.data.mul_(1.7)  # a write it does not count
    model.model.layers[1].mlp.up_proj.weight.data.mul_(0.6)
    m.refresh_weights()
    after = m.batch_next_token_logprobs_sync(prompts[:1])[0]
    assert (after - own()).abs().max().item() < TOL and (after - before).abs().max().item() > 1e-3


def test_batched_submit_on_gpu_matches_reference(llm):
    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    m.set_rng("torch", 1234)
    m.cache_kv(prompt)
    ctxs, lw, active = [[] for _ in range(16)], np.zeros(16), [True] * 16
    while any(active):
        idx = [i for i in range(16) if active[i]]
        logZ, tok = m.batch_next_token_step_sync([prompt + ctxs[i] for i in idx], [1 if len(ctxs[i]) >= 10 else 0 for i in idx])
        for i, z, t in zip(idx, logZ, tok):
            lw[i] += z
            if t <= 0:
                active[i] = False
            else:
                ctxs[i].append(int(t))
    assert ctxs == [_strip(r) for r in gold["sis_contexts"]]
    assert np.abs(lw - gold["sis_log_weights"]).max() < TOL


# ---- the RCCL path on one GPU; the device guard ---------------------------------------------------------------
@pytest.fixture()
def nccl_single():
    """A one-rank "nccl" (= RCCL) process group on this GPU: the collectives of an 8-GPU run, executed."""
    import socket

    import torch.distributed as dist

    if dist.is_initialized():
        dist.destroy_process_group()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "pkv", "prefix"])
def test_rccl_collectives_on_one_rank_change_nothing(llm, nccl_single, mode):
    """DeviceSIS with the multi-rank branch forced on (all-gather of log-weights + active counts every step, of the
    token matrices and meta rows at every resampling step, the set-up reductions - all through RCCL) gives the tokens
    and weights of the run without collectives, through several resampling steps."""
    from genlm_backend_amd.sis import DeviceSIS

    dist = nccl_single
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    p = [int(t) for t in gold["sis_prompt"]]
    prompts = [p, p[:5], p[2:], p] * 8
    kw = dict(use_particle_kv=mode == "pkv", use_prefix_kv=mode == "prefix")
    runs = []
    for force in (True, False):
        s = DeviceSIS(m, 32, prompts, max_tokens=6, eos_id=0, seed=5, resample_ess=1.0, dist=dist if force else None,
                      force_collectives=force, **kw)
        assert s.collective == force
        s.run()
        assert s.n_resamples >= 2
        probs, stats = s.normalized_weights()
        runs.append((s.results(), probs.cpu().numpy(), stats.cpu().numpy()))
    assert runs[0][0][0] == runs[1][0][0]
    assert np.array_equal(runs[0][0][1].view(np.uint32), runs[1][0][1].view(np.uint32))
    assert np.array_equal(runs[0][1].view(np.uint32), runs[1][1].view(np.uint32))
    # the gathered vector really went through the collective
    probe = torch.arange(4, dtype=torch.float32, device=m.device)
    got = torch.empty(4, device=m.device)
    dist.all_gather_into_tensor(got, probe)
    assert got.cpu().tolist() == [0.0, 1.0, 2.0, 3.0]


def test_rccl_self_exchange_of_the_calls_a_multi_gpu_run_makes(nccl_single):
    """The RCCL calls of the N > 1 path that a one-rank run never reaches by itself (with one rank no particle changes
    ranks, so `rows_moved == 0`): forced here as self-exchanges on DEVICE tensors through the "nccl" backend -
    `all_to_all_single` with uneven split lists on int32 particle rows (sis._all_to_all: resample()), on bf16 5-D KV rows
    (sis._all_to_all_any: _migrate_kv()), an empty exchange, and the float64 all_reduce(MAX) of bench.py's clock."""
    from genlm_backend_amd.sis import _all_to_all, _all_to_all_any, _gather_all, _reduce_all

    dist = nccl_single
    assert dist.get_backend() == "nccl"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    for rows, width in ((37, 22), (1, 3), (512, 19 + 3)):
        send = torch.randint(-5, 1 << 30, (rows, width), dtype=torch.int32, device=dev, generator=g)
        recv = torch.full_like(send, -7)
        _all_to_all(dist, recv, send, [rows], [rows])
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
    # KV rows as _migrate_kv sends them: [rows, layers x {K, V}, heads, cap, head_dim], bf16 and fp32
    for dt in (torch.bfloat16, torch.float32):
        send = torch.randn((9, 2 * 3, 4, 14, 16), device=dev, generator=g).to(dt)
        recv = torch.zeros_like(send)
        _all_to_all_any(dist, recv, send, [9], [9])
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
    # nothing moves (what every step of a one-rank run would send): legal, a no-op
    empty_s = torch.empty((0, 22), dtype=torch.int32, device=dev)
    empty_r = torch.empty((0, 22), dtype=torch.int32, device=dev)
    _all_to_all(dist, empty_r, empty_s, [0], [0])
    # bench.py:366-368's max-over-ranks clock and the all-gather of log-weights (4 KiB at 1024 particles)
    t = torch.tensor([1.2345678901234567], dtype=torch.float64, device=dev)
    _reduce_all(dist, t, dist.ReduceOp.MAX)
    assert float(t.item()) == 1.2345678901234567
    mm = torch.tensor([8, -17], dtype=torch.int64, device=dev)
    _reduce_all(dist, mm, dist.ReduceOp.MIN)
    assert mm.cpu().tolist() == [8, -17]
    lw = torch.randn(1025, device=dev, generator=g)
    out = torch.empty(1025, device=dev)
    _gather_all(dist, out, lw)
    torch.cuda.synchronize()
    assert torch.equal(out, lw)


@pytest.mark.parametrize("dtype,head_dim", [(torch.float32, 64), (torch.bfloat16, 64), (torch.bfloat16, 128)])
def test_llama_shadow_equals_the_callers_model_on_gpu(engine, dtype, head_dim):
    """The shadow the backend runs (fuse.py: RMSNorm as rms_norm, q / k / v from one GEMM, the rotary embedding for queries
    and keys in one pass, glb attention) against the caller's untouched HuggingFace model on the device: a padded batch
    with a mask, and a prefill + one-token decode through transformers' own DynamicCache - float32 within rounding, bfloat16
    within a few bf16 ulps of the logits; the caller's model gives bit-identical results before and after."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from genlm_backend_amd.llm import AsyncAmdLM

    torch.manual_seed(11)
    dev = engine.device
    cfg = LlamaConfig(vocab_size=1000, hidden_size=4 * head_dim, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=head_dim, max_position_embeddings=64, bos_token_id=1, eos_token_id=2)
    model = LlamaForCausalLM(cfg).to(dtype).to(dev).eval()
    ids = torch.randint(3, 1000, (5, 9), device=dev)
    am = torch.ones_like(ids)
    am[1, 6:] = 0
    am[3, 4:] = 0
    with torch.no_grad():
        want = model(input_ids=ids, attention_mask=am).logits
        w1 = model(input_ids=ids[:, :6], use_cache=True)
        want2 = model(input_ids=ids[:, 6:7], past_key_values=w1.past_key_values).logits
    m = AsyncAmdLM(model, None, engine=engine)
    assert m._net is not model and set(m.fused) == {"rms_norm", "rope", "gate_up"} and m.glb_attention
    with torch.no_grad():
        got = m._lm_head(m._body(input_ids=ids, attention_mask=am).last_hidden_state)
        g1 = m._body(input_ids=ids[:, :6], use_cache=True)
        got2 = m._lm_head(m._body(input_ids=ids[:, 6:7], past_key_values=g1.past_key_values).last_hidden_state)
        again = model(input_ids=ids, attention_mask=am).logits
    tol = 2e-4 if dtype == torch.float32 else 6e-2
    real = am.bool()  # (padded positions of a row: whatever the model's own attention leaves there)
    assert (got.float() - want.float())[real].abs().max().item() < tol
    assert (got2.float() - want2.float()).abs().max().item() < tol
    assert torch.equal(again, want)


def test_c_abi_allgather_without_torch_distributed(engine):
    """glb_comm_* / glb_allgather_f32: the path's one collective for a caller that binds the C ABI without PyTorch - a
    one-rank RCCL communicator made from a unique id, the all-gather of 1025 floats (the shard's log-weights + its active
    count), destroyed again.  (N > 1 needs a GPU per rank: the driver's 8-GPU node.)"""
    import ctypes as C

    from genlm_backend_amd import _lib

    lib = engine.lib
    uid = (C.c_char * 128)()
    rc = lib.glb_comm_unique_id(C.cast(uid, C.c_void_p))
    if rc == _lib.GLB_EUNSUPPORTED:
        pytest.skip("no RCCL in this process: " + _lib.last_error())
    assert rc == 0, _lib.last_error()
    comm = C.c_void_p()
    assert lib.glb_comm_init(C.cast(uid, C.c_void_p), 0, 1, C.byref(comm)) == 0, _lib.last_error()
    dev = engine.device
    lw = torch.randn(1025, device=dev)
    out = torch.full((1025,), -1.0, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.glb_allgather_f32(comm, C.c_void_p(lw.data_ptr()), 1025, C.c_void_p(out.data_ptr()), stream) == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert torch.equal(out, lw)
    assert lib.glb_allgather_f32(None, C.c_void_p(lw.data_ptr()), 1025, C.c_void_p(out.data_ptr()), stream) == _lib.GLB_EINVAL
    assert lib.glb_comm_init(C.cast(uid, C.c_void_p), 1, 1, C.byref(comm)) == _lib.GLB_EINVAL
    assert lib.glb_comm_destroy(comm) == 0


def test_engine_launches_on_its_own_device(engine, oracle):
    """HipEngine runs every entry point with ITS device current, whatever the calling thread has set (one process
    driving several GPUs).  With one visible GPU the guard is exercised through its bookkeeping: the proxy resolves and
    the call lands on cuda:0; with two, a second engine on cuda:1 is driven while cuda:0 is current and vice versa."""
    from genlm_backend_amd.engine import HipEngine

    x = synth.logits(11, 3, 1000)
    want = oracle.step(x, rng_mode=oracle.RNG_PHILOX, seed=3, offset=1)
    n_dev = torch.cuda.device_count()
    engines = [engine] + ([HipEngine("cuda:1")] if n_dev > 1 else [])
    for cur in range(min(n_dev, 2)):
        torch.cuda.set_device(cur)
        for eng in engines:
            got = eng.step(torch.from_numpy(x).to(eng.device), rng_mode=1, seed=3, offset=1)
            torch.cuda.synchronize(eng.device)
            for w, g in zip(want, got):
                assert g.device == eng.device
                assert np.array_equal(g.cpu().numpy().view(np.uint32), w.view(np.uint32))
    torch.cuda.set_device(0)


# ---- batch_sample on the HIP path (base.py:148-179) ---------------------------------------------------------------
G3 = os.path.join(os.path.dirname(__file__), "golden", "ref_round3.npz")


@pytest.mark.parametrize("sync_every", [1, 4])
def test_batch_sample_on_gpu_matches_reference(llm, sync_every):
    """Ten ragged prompts, temperature 0.2, two stopping tokens, sequences ending at different steps: the device loop
    (per-sequence KV slabs, the fused step with logit_scale = 1/T and ONE Exp(1) row shared by every sequence - noise
    pitch 0 -, active count read back every `sync_every` steps) returns the reference's ids."""
    m, _ = llm
    g3 = np.load(G3)
    prompts = [_strip(r) for r in g3["bs_prompts"]]
    ids = m.batch_sample_sync(prompts, max_tokens=int(g3["bs_params"][0]), eos_token_ids=[int(t) for t in g3["bs_eos"]],
                              temperature=float(g3["bs_temperature"][0]), seed=int(g3["bs_params"][1]),
                              sync_every=sync_every)
    assert ids == [_strip(r) for r in g3["bs_ids"]]
    assert len({len(r) for r in ids}) >= 3
    free = m.batch_sample_sync(prompts, max_tokens=10, eos_token_ids=[], temperature=float(g3["bs_temperature"][0]),
                               seed=int(g3["bs_params"][1]), sync_every=sync_every)
    assert np.array_equal(np.array(free), g3["bs_ids_free"])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_shared_noise_row_step(engine, oracle, dtype):
    """engine.step with noise [1, V] and n > 1 particles (pitch 0: one row for everybody) == the oracle with the row
    repeated; with and without a temperature, over deduplicated rows."""
    n, U, V = 37, 9, 5003
    x = synth.logits(21, U, V)
    row_of = (np.arange(n) * 7 % U).astype(np.int32)
    E = oracle.mt_exponential(77, V)[0]
    if dtype == "bf16":
        xb = oracle.f32_to_bf16_bits(x)
        xin, xt = xb, torch.from_numpy(xb.view(np.int16)).view(torch.bfloat16)
    else:
        xin, xt = x, torch.from_numpy(x)
    dev = engine.device
    for scale in (1.0, 1.0 / 0.7):
        want = oracle.step(xin, row_of=row_of, rng_mode=oracle.RNG_NOISE, noise=np.tile(E, (n, 1)), logit_scale=scale,
                           want_margin=True)
        margin = torch.empty(n, device=dev)
        got = engine.step(xt.to(dev), row_of=torch.from_numpy(row_of).to(dev), rng_mode=2,
                          noise=torch.from_numpy(E).view(1, V).to(dev), logit_scale=scale, out_margin=margin)
        torch.cuda.synchronize()
        for w, g in zip(want[:3], got):
            assert np.array_equal(g.cpu().numpy().view(np.uint32), w.view(np.uint32))
        assert np.array_equal(margin.cpu().numpy().view(np.uint32), want[3].view(np.uint32))


def test_context_hashes_extend_token_by_token(engine, oracle):
    """glb_hash_contexts == the fold the CPU double states; glb_particles_advance extends a context's hash by the
    appended token; glb_group_contexts with the caller's hashes groups exactly like the oracle (and like itself
    without them)."""
    from tests.cpu_engine import CpuOracleEngine

    rng = np.random.default_rng(4)
    n, cap = 1024, 24
    mat = rng.integers(0, 7, size=(n, cap)).astype(np.int32)
    ln = rng.integers(1, 6, size=n).astype(np.int32)
    dev = engine.device
    ctx = _dev(mat, dev)
    st = _dev(np.arange(n, dtype=np.int64) * cap, dev)
    ln_d = _dev(ln, dev)
    h = engine.hash_contexts(ctx.view(-1), st, ln_d)
    want = np.array([CpuOracleEngine._hash(mat[i, :ln[i]]) for i in range(n)], np.uint64)
    assert np.array_equal(h.cpu().numpy().view(np.uint64), want)
    # three appended tokens (some particles stop on the way)
    active = torch.ones(n, dtype=torch.int32, device=dev)
    lw = torch.zeros(n, device=dev)
    for step in range(3):
        tok = _dev(rng.integers(0, 7, size=n).astype(np.int32), dev)
        engine.particles_advance(ctx, ln_d, active, lw, torch.zeros(n, device=dev), tok, 0, cap, hashes=h)
        mat2, ln2 = ctx.cpu().numpy(), ln_d.cpu().numpy()
        want = np.array([CpuOracleEngine._hash(mat2[i, :ln2[i]]) for i in range(n)], np.uint64)
        assert np.array_equal(h.cpu().numpy().view(np.uint64), want)
        g_o, rep_o, ng_o = oracle.group_contexts([list(mat2[i, :ln2[i]]) for i in range(n)])
        for hashes in (h, None):
            g, rep, ng = engine.group_contexts(ctx.view(-1), st, ln_d, hashes=hashes)
            assert int(ng.item()) == ng_o and np.array_equal(g.cpu().numpy(), g_o)
            assert np.array_equal(rep.cpu().numpy()[:ng_o], rep_o)


def test_logprob_rows_stay_under_the_byte_budget_on_gpu(llm):
    """50 steps of 1024 contexts through batch_next_token_logprobs with a 24 MB budget for the trie's log-prob rows:
    the store never holds more than the budget plus the newest slab, device memory does not grow with the steps, and
    an evicted row is recomputed on demand."""
    m, _ = llm
    V = m.model.config.vocab_size
    rng = np.random.default_rng(1)
    ctxs = [[int(t) for t in rng.integers(1, V, 4)] for _ in range(1024)]
    first = asyncio.run(m.batch_next_token_logprobs(ctxs)).clone()
    slab = m._rows.used
    assert slab >= 1024 * V * 4
    m._rows.budget = 24 << 20
    torch.cuda.synchronize()
    base_mem, peak_store, mems = torch.cuda.memory_allocated(), 0, []
    for step in range(50):
        ctxs2 = [c + [int(t)] for c, t in zip(ctxs, rng.integers(1, V, len(ctxs)))]
        rows = asyncio.run(m.batch_next_token_logprobs(ctxs2))
        assert rows.shape == (1024, V)
        del rows
        peak_store = max(peak_store, m._rows.used)
        torch.cuda.synchronize()
        mems.append(torch.cuda.memory_allocated())
    assert m._rows.evictions >= 40
    assert peak_store <= m._rows.budget + 3 * slab
    assert max(mems[10:]) - min(mems[10:]) < 8 * slab and max(mems) - base_mem < m._rows.budget + 12 * slab
    again = asyncio.run(m.batch_next_token_logprobs(ctxs[:64]))
    assert np.abs(again.cpu().numpy() - first[:64].cpu().numpy()).max() < 1e-5


def test_shared_kv_rows_on_gpu(llm):
    """Shared KV rows on the device: duplicates of an ancestor are forwarded once (forward rows < particles x steps),
    diverging particles get copies of the shared prefix, and a row budget that is spent (8 rows for 32 particles)
    gives the same tokens - the contexts that find no row are encoded from their tokens."""
    from genlm_backend_amd.sis import DeviceSIS

    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    p = [int(t) for t in gold["sis_prompt"]]
    prompts = [p, p[:5], p[2:], p] * 8
    ref = DeviceSIS(m, 32, prompts, max_tokens=6, eos_id=0, seed=5, resample_ess=1.0)
    ref.run()
    for rows in (None, 8):
        s = DeviceSIS(m, 32, prompts, max_tokens=6, eos_id=0, seed=5, resample_ess=1.0, use_particle_kv=True, kv_rows=rows)
        s.run()
        assert s.results()[0] == ref.results()[0]
        assert np.abs(s.results()[1] - ref.results()[1]).max() < 1e-4
        st = s.kv_stats
        assert st["forward_rows"] < 32 * st["steps"]
        if rows is None:
            assert st["encoded_rows"] == 3 and st["unkept_rows"] == 0 and st["copied_rows"] > 0
        else:
            assert st["unkept_rows"] > 0 and s.pkv.n == 8


def test_trie_masses_from_logits(engine, oracle):
    """glb_trie_masses: masses of softmax(logits) straight from the logits rows + the fused step's lse (no log-prob
    matrix), for fp32 and bf16 logits; the three output forms (row-major, node-major, selected nodes) hold the same
    bits; weights of a 16-bit type == the oracle on the upcast values."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(3)
    words, seen = [], set()
    while len(words) < 3000:
        w = bytes(rs.integers(97, 101, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    dev = engine.device
    V, nn = len(words), len(trie)
    for B in (70, 5):
        x = (rs.standard_normal((B, V)) * 3).astype(np.float32)
        for dt in (torch.float32, torch.bfloat16):
            xd = torch.from_numpy(x).to(dev).to(dt)
            _, lse, _ = engine.step(xd, rng_mode=0)
            rows = trie.masses_from_logits(xd, lse)
            p = torch.softmax(xd.float().double(), -1).cpu().numpy()
            want = oracle.trie_reduce(p.astype(np.float32), trie.flat(), 0)
            assert np.abs(rows.cpu().numpy() - want).max() < 2e-5
            assert np.abs(rows[:, trie.root].cpu().numpy() - 1.0).max() < 2e-5
            sel = torch.from_numpy(rs.choice(nn, 257, replace=False).astype(np.int32)).to(dev)
            got_sel = trie.masses_from_logits(xd, lse, nodes=sel)
            assert torch.equal(got_sel, rows[:, sel.long()])
            nm = trie.masses_from_logits(xd, lse, layout="nodes")
            assert torch.equal(nm[:, :B].t().contiguous(), rows)
            sl = trie.masses_from_logits(xd, lse, layout="slots")   # the folded trie: a node's value sits in its slot
            slot_of = trie.compact_device_arrays()["slot_of"].long()
            assert sl.shape[0] == trie.compact()["n_nodes"] < nn
            assert torch.equal(sl[slot_of][:, :B].t().contiguous(), rows)
    # weights (no exp) of a 16-bit type: exact against the oracle on the upcast values
    w = rs.random((40, V)).astype(np.float32)
    wb = torch.from_numpy(w).to(torch.bfloat16)
    got = engine.trie_masses(wb.to(dev), trie.device_arrays(), 0, False)
    want = oracle.trie_reduce(wb.float().numpy(), trie.flat(), 0)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("cap", [250, 700, 20000])
def test_trie_rows_in_lds_equal_the_level_kernels_and_the_oracle(engine, oracle, cap):
    """glb_trie_rows (one row of a part of the trie resident in LDS; trie.plan cuts the folded trie into parts of at most
    `cap` slots - several parts and a top for the small caps, one part for the big one): every node,
    selected nodes and the slot-major form hold the oracle's bits for weights (sum and max, fp32 and bf16, 1 / 9 / 70 rows)
    and the level-synchronous kernels' bits for masses straight from logits + lse."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(17)
    words, seen = [], set()
    while len(words) < 3000:
        w = bytes(rs.integers(97, 102, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    trie.PLAN_CAP = cap
    pl = trie.plan()
    assert pl is not None and pl["max_local"] <= cap and (pl["n_top"] > 0) == (cap < pl["n_slots"])
    old = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    old.resident = False
    dev = engine.device
    V, nn = len(words), len(trie)
    for B in (1, 9, 70):
        w = rs.random((B, V)).astype(np.float32)
        for op in (0, 1):
            got = trie._batch(torch.from_numpy(w), op, False).cpu().numpy()
            assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(w, trie.flat(), op).view(np.uint32))
        wb = torch.from_numpy(w).to(torch.bfloat16).to(dev)
        got = engine.trie_rows(wb, trie.plan_device_arrays(), 0, False).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(wb.float().cpu().numpy(), trie.flat(), 0).view(np.uint32))
        x = (rs.standard_normal((B, V + 5)) * 3).astype(np.float32)  # (a row pitch above the vocabulary)
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            xd = torch.from_numpy(x).to(dev).to(dt)
            _, lse, _ = engine.step(xd, vocab=V, rng_mode=0)
            rows = trie.masses_from_logits(xd, lse)
            assert torch.equal(rows, old.masses_from_logits(xd, lse))
            sel = torch.from_numpy(rs.choice(nn, 300, replace=True).astype(np.int32)).to(dev)
            assert torch.equal(trie.masses_from_logits(xd, lse, nodes=sel), rows[:, sel.long()])
            sl = trie.masses_from_logits(xd, lse, layout="slot_rows")
            assert sl.shape == (B, pl["n_slots"])
            assert torch.equal(sl[:, torch.from_numpy(trie.slot_plan()["slot_of"].astype(np.int64)).to(dev)], rows)
            half = trie.masses_from_logits(xd, lse, logit_scale=0.5)
            assert torch.equal(half, old.masses_from_logits(xd, lse, logit_scale=0.5))
            own = trie.masses_from_logits(xd[:, :V].contiguous(), logit_scale=0.5)  # lse computed by the call itself (row_lse)
            assert (own[:, trie.root] - 1.0).abs().max().item() < 1e-4


@pytest.mark.parametrize("cap", [250, 20000])
def test_selected_masses_read_only_the_selected_subtrees(engine, oracle, cap):
    """Masses of SELECTED nodes (round 5): `TokenByteTrie.selection_plan` plans only the sub-forest below the selection's
    maximal nodes, so glb_trie_rows reads those subtrees' tokens and reduces those nodes only - results bit-equal to the
    whole trie's (and through it to the oracle) for random nodes, leaves, a depth-1 node with some of its descendants, and
    - falling back to the whole plan - the root."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(23)
    words, seen = [], set()
    while len(words) < 4000:
        w = bytes(rs.integers(97, 104, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    trie.PLAN_CAP = cap
    dev = engine.device
    V, nn = len(words), len(trie)
    x = (rs.standard_normal((37, V)) * 3).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    _, lse, _ = engine.step(xd, vocab=V, rng_mode=0)
    trie.prune_selection = False
    rows = trie.masses_from_logits(xd, lse)
    want = oracle.trie_reduce(np.exp(x.astype(np.float64) - lse.cpu().numpy().astype(np.float64)[:, None]).astype(np.float32), trie.flat(), 0)
    assert np.abs(rows.cpu().numpy() - want).max() < 1e-5
    trie.prune_selection = True
    d1 = sorted(trie.children[trie.root].values())
    sels = {"random": rs.choice(nn, 256, replace=False), "leaves": trie.idx_to_leaf[rs.choice(V, 500, replace=False), 1],
            "depth1": np.asarray([d1[0]] + [int(c) for c in trie.jump[d1[0]]]), "root": np.asarray([trie.root, 3, 5])}
    for name, sel in sels.items():
        sel_d = torch.from_numpy(np.asarray(sel, np.int32)).to(dev)
        got = trie.masses_from_logits(xd, lse, nodes=sel_d)
        assert torch.equal(got, rows[:, sel_d.long()]), name
        pl = trie.selection_plan(sel_d)
        assert (pl is None) == (name == "root"), name
        if pl is not None:
            assert pl["n_slots"] < trie.plan()["n_slots"] // 2
        for dt in (torch.bfloat16,):
            xb = xd.to(dt)
            _, lb, _ = engine.step(xb, vocab=V, rng_mode=0)
            trie.prune_selection = False
            full = trie.masses_from_logits(xb, lb)
            trie.prune_selection = True
            assert torch.equal(trie.masses_from_logits(xb, lb, nodes=sel_d), full[:, sel_d.long()]), name


def test_selection_plans_are_made_when_asked_for_not_behind_a_call(engine):
    """a selection met for the first time used to cost a device-to-host copy of its ids, a Python plan of its
    sub-forest and twenty uploads inside `masses_from_logits`.  The default now serves an unprepared selection from the whole
    trie's plan (one launch, no synchronisation) and takes the pruned plan once `prepare_selection(nodes)` was called; ids
    outside the trie ("none" to the kernel: negative, or past the last node) select nothing on the host either."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(5)
    words = sorted({bytes(rs.integers(97, 104, int(rs.integers(1, 7))).astype(np.uint8)) for _ in range(3000)})
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    trie.PLAN_CAP = 250
    assert trie.prune_selection == "cached"
    dev = engine.device
    V, nn = len(words), len(trie)
    xd = torch.from_numpy((rs.standard_normal((9, V)) * 3).astype(np.float32)).to(dev)
    _, lse, _ = engine.step(xd, vocab=V, rng_mode=0)
    rows = trie.masses_from_logits(xd, lse)
    sel = rs.choice(nn, 200, replace=False).astype(np.int32)
    sel[::17] = -1  # "none"
    sel_d = torch.from_numpy(sel).to(dev)
    ok = sel_d >= 0  # (the column of an id outside the trie is left untouched by the kernel, whichever plan serves the call)
    want = rows[:, sel_d.clamp_min(0).long()][:, ok]
    assert trie.selection_plan(sel_d) is None  # nothing was planned ...
    assert torch.equal(trie.masses_from_logits(xd, lse, nodes=sel_d)[:, ok], want)
    assert trie.selection_plan(sel_d) is None and not trie._sel_plans  # ... by the call either
    assert trie.prepare_selection(sel_d) and trie.selection_plan(sel_d) is not None
    assert trie.selection_plan(sel_d)["n_slots"] < trie.plan()["n_slots"] // 2
    assert torch.equal(trie.masses_from_logits(xd, lse, nodes=sel_d)[:, ok], want)
    assert not trie.prepare_selection(torch.tensor([trie.root, 3], dtype=torch.int32, device=dev))  # the root: the whole plan


@pytest.mark.parametrize("cap", [250, 20000])
def test_per_row_selections_read_only_the_parts_a_row_needs(engine, cap):
    """A selection PER ROW (round 5: `nodes` int32 [B, K] - every particle's current node's children, what a byte-level
    sampler reads after trie/base.py:147-213): row r's values are the whole trie's values of row r's nodes, bit for bit;
    negative entries give 0; rows whose nodes sit above the cut (the root's children at a small cap) take every part."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(29)
    words, seen = [], set()
    while len(words) < 4000:
        w = bytes(rs.integers(97, 104, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    trie.PLAN_CAP = cap
    dev = engine.device
    V, nn, B, K = len(words), len(trie), 41, 12
    x = (rs.standard_normal((B, V)) * 3).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    _, lse, _ = engine.step(xd, vocab=V, rng_mode=0)
    rows = trie.masses_from_logits(xd, lse).cpu().numpy()
    internal = [n for n in range(nn) if len(trie.jump[n]) >= 2]
    sel = np.full((B, K), -1, np.int32)
    for r in range(B):
        node = trie.root if r % 7 == 0 else int(rs.choice(internal))
        kids = trie.jump[node][:K]
        sel[r, :len(kids)] = kids
        if r % 5 == 0:
            sel[r, 0] = -1  # a hole in the middle of a row
        if r == 3:
            sel[r] = -1     # a row that asks for nothing
    got = trie.masses_from_logits(xd, lse, nodes=torch.from_numpy(sel).to(dev)).cpu().numpy()
    for r in range(B):
        for k in range(K):
            want = rows[r, sel[r, k]] if sel[r, k] >= 0 else 0.0
            assert got[r, k].view(np.uint32) == np.float32(want).view(np.uint32), (r, k)
    xb = xd.to(torch.bfloat16)
    _, lb, _ = engine.step(xb, vocab=V, rng_mode=0)
    full = trie.masses_from_logits(xb, lb)
    gb = trie.masses_from_logits(xb, lb, nodes=torch.from_numpy(sel).to(dev))
    idx = torch.from_numpy(np.where(sel >= 0, sel, 0).astype(np.int64)).to(dev)
    assert torch.equal(gb, torch.where(torch.from_numpy(sel >= 0).to(dev), torch.gather(full, 1, idx), torch.zeros_like(gb)))
    with pytest.raises(Exception):
        trie.masses_from_logits(xd, lse, nodes=torch.from_numpy(sel[:5]).to(dev))


@pytest.mark.parametrize("cap", [300, 2000, None])
def test_trie_sweep_kernel_equals_the_gathered_one_and_the_oracle(engine, oracle, cap):
    """glb_trie_rows on a SWEEP plan (round 5: a persistent workgroup per part reads row after row front to back, the tokens'
    slots and the part's internal nodes in registers): the bits of the gathered plan and of the oracle - several parts and a
    top (small caps) or one part, 1 / 5 / 130 / 300 rows (more rows than persistent lanes: a workgroup takes several), every
    element type, sums and maxima, weights with an odd pitch, all nodes / slots / one selection / a selection per row (rows
    that ask nothing of a part skip it) / two outputs at once."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(23)
    words, seen = [], set()
    while len(words) < 3001:
        w = bytes(rs.integers(97, 102, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    V, nn, dev = len(words), len(trie), engine.device
    sw = trie.plan_device_arrays(cap, sweep=True)
    ga = trie.plan_device_arrays()
    host = trie.plan(cap, sweep=True)
    assert sw is not None and sw["sweep"] and (host["n_parts"] > 1) == (cap is not None) and (host["n_top"] > 0) == (cap is not None)
    slot_of = sw["slot_of"].long()
    for B in (1, 5, 130, 300):
        w = rs.random((B, V + 3)).astype(np.float32)
        wd = torch.from_numpy(w).to(dev)[:, :V]  # (pitch V + 3: rows start on any word)
        for op in (0, 1):
            got = engine.trie_rows(wd, sw, op, False)
            assert np.array_equal(got.cpu().numpy().view(np.uint32), oracle.trie_reduce(w[:, :V].copy(), trie.flat(), op).view(np.uint32)), (B, op)
        x = rs.standard_normal((B, V)).astype(np.float32) * 3
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            xd = torch.from_numpy(x).to(dev).to(dt)
            _, lse, _ = engine.step(xd, vocab=V, rng_mode=0)
            rows = engine.trie_rows(xd, ga, 0, True, lse=lse)
            assert torch.equal(engine.trie_rows(xd, sw, 0, True, lse=lse), rows)
            assert torch.equal(engine.trie_rows(xd, sw, 0, True, lse=lse, logit_scale=0.5), engine.trie_rows(xd, ga, 0, True, lse=lse, logit_scale=0.5))
            sl = engine.trie_rows(xd, sw, 0, True, lse=lse, layout="slots")
            assert sl.shape == (B, host["n_slots"]) and torch.equal(sl[:, slot_of], rows)
            sel = torch.from_numpy(rs.choice(nn, 77, replace=True).astype(np.int32)).to(dev)
            both = torch.full((B, host["n_slots"]), -1.0, device=dev)
            assert torch.equal(engine.trie_rows(xd, sw, 0, True, lse=lse, nodes=sel, out_slots=both), rows[:, sel.long()])
            assert torch.equal(both, sl)
            # a selection per row: the children of one of the root's children, or nothing
            d1 = sorted(trie.children[trie.root].values())
            K = max(len(trie.jump[c]) for c in d1)
            rowsel = np.full((B, K), -1, np.int32)
            for r in range(B):
                if r % 7 != 3:
                    c = d1[int(rs.integers(0, len(d1)))]
                    rowsel[r, :len(trie.jump[c])] = trie.jump[c]
            rsd = torch.from_numpy(rowsel).to(dev)
            got = engine.trie_rows(xd, sw, 0, True, lse=lse, nodes=rsd)
            want = torch.where(rsd >= 0, torch.gather(rows, 1, rsd.clamp(min=0).long()), torch.zeros((), device=dev))
            assert torch.equal(got, want)
            if cap is None:  # (the host entry point's hint that the selections are wide: the default sweep plan)
                assert torch.equal(trie.masses_from_logits(xd, lse, nodes=rsd, wide_selections=True), want)
                assert torch.equal(trie.masses_from_logits(xd, lse, nodes=rsd), want)


def test_trie_sweep_kernel_on_deep_and_tiny_tries(engine, oracle):
    """Words of up to 24 letters over two symbols: more depths than the reduction's work list holds trips in registers (the
    rest comes straight from global memory); and the degenerate vocabularies (one token, one chain, two leaves)."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(29)
    words, seen = [], set()
    while len(words) < 4000:
        w = bytes(rs.integers(97, 99, int(rs.integers(1, 25))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    for vocab in (words, [b"a"], [b"abc"], [b"a", b"b"], [b"a", b"ab", b"abc"]):
        trie = TokenByteTrie([Token(i, w) for i, w in enumerate(vocab)], engine=engine)
        sw = trie.plan_device_arrays(sweep=True)
        assert sw is not None
        if len(vocab) > 10:
            assert int(trie.plan(sweep=True)["desc"][0, 3]) > 14  # (depths of part 0)
        ws = rs.random((9, len(vocab))).astype(np.float32)
        for op in (0, 1):
            got = engine.trie_rows(torch.from_numpy(ws).to(engine.device), sw, op, False).cpu().numpy()
            assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(ws, trie.flat(), op).view(np.uint32))



def test_trie_rows_on_degenerate_vocabularies(engine, oracle):
    """One token, one chain, two leaves under the root: the plan has one part of one to five slots and no top."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    for vocab in ([b"a"], [b"abc"], [b"a", b"b"], [b"a", b"ab", b"abc"]):
        trie = TokenByteTrie([Token(i, w) for i, w in enumerate(vocab)], engine=engine)
        assert trie.plan()["n_top"] == 0
        ws = np.random.default_rng(3).random((3, len(vocab))).astype(np.float32)
        for op, fn in ((0, trie.batch_weight_sum), (1, trie.batch_weight_max)):
            assert np.array_equal(fn(torch.from_numpy(ws)).view(np.uint32), oracle.trie_reduce(ws, trie.flat(), op).view(np.uint32))


def test_trie_rows_at_llama_vocabulary_size(engine, oracle):
    """128 256 synthetic tokens (config 5's vocabulary size; 159 k slots in 26 parts + the root): weights == the oracle bit
    for bit, masses from bf16 logits + lse == the level-synchronous kernels' bits, selected nodes == columns of the rows."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(1)
    words, seen = [], set()
    while len(words) < 128256:
        w = bytes(rs.integers(97, 123, int(rs.integers(1, 10))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=engine)
    pl = trie.plan()
    assert pl is not None and pl["n_parts"] > 8 and pl["n_top"] >= 1
    dev = engine.device
    w = rs.random((3, len(words))).astype(np.float32)
    got = trie.batch_weight_sum(torch.from_numpy(w))
    assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(w, trie.flat(), 0).view(np.uint32))
    x = (torch.randn((40, len(words)), device=dev) * 3).to(torch.bfloat16)
    _, lse, _ = engine.step(x, rng_mode=0)
    rows = trie.masses_from_logits(x, lse)
    trie.resident = False
    assert torch.equal(rows, trie.masses_from_logits(x, lse))
    trie.resident = True
    sel = torch.from_numpy(rs.choice(len(trie), 1000, replace=False).astype(np.int32)).to(dev)
    assert torch.equal(trie.masses_from_logits(x, lse, nodes=sel), rows[:, sel.long()])
    assert (rows[:, trie.root] - 1.0).abs().max().item() < 1e-3  # (bf16 logits against a float32 lse of the same values)
    # the sweep plan (round 5): 5 parts of <= 40 000 slots, a row longer than what a thread keeps in registers
    sw = trie.plan_device_arrays(sweep=True)
    assert sw is not None and 4 <= trie.plan(sweep=True)["n_parts"] <= 6
    assert torch.equal(engine.trie_rows(x, sw, 0, True, lse=lse), rows)
    sl = trie.masses_from_logits(x, lse, layout="slot_rows")
    assert torch.equal(sl[:, sw["slot_of"].long()], rows)
    trie.SWEEP_MIN_ROWS = 100  # (from SWEEP_MIN_ROWS rows on, the rows go through the sweep plan too)
    big = torch.cat([x] * 4)[:150]
    assert torch.equal(trie.masses_from_logits(big, torch.cat([lse] * 4)[:150]), torch.cat([rows] * 4)[:150])


def test_async_trie_batches_concurrent_requests(engine, oracle):
    """AsyncTokenByteTrie (trie/async_impl.py counterpart): 40 coroutines asking for sums and 9 for maxima are served by
    one device batch each, every caller gets its own row, bit for bit what the batched call gives; a bad request fails
    its batch's callers and the trie keeps serving."""
    import asyncio

    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import AsyncTokenByteTrie

    rs = np.random.default_rng(11)
    words, seen = [], set()
    while len(words) < 500:
        w = bytes(rs.integers(97, 101, int(rs.integers(1, 6))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    at = AsyncTokenByteTrie.from_vocab([Token(i, w) for i, w in enumerate(words)], engine=engine)
    ws = rs.random((49, len(words))).astype(np.float32)
    calls = []
    orig = at.trie._batch
    at.trie._batch = lambda rows, op, flp: (calls.append((rows.shape[0], op)), orig(rows, op, flp))[1]

    async def main():
        res = await asyncio.gather(*[at.weight_sum(torch.from_numpy(ws[i])) for i in range(40)],
                                   *[at.weight_max(ws[i]) for i in range(40, 49)])
        with pytest.raises(Exception):
            await at.weight_sum(torch.zeros(3))          # wrong length: the batch fails, the caller hears of it
        again = await at.weight_sum(ws[0])
        await at.cleanup()
        return res, again

    res, again = asyncio.run(main())
    assert sorted(calls[:2]) == [(9, 1), (40, 0)]
    want_s = oracle.trie_reduce(ws[:40], at.trie.flat(), 0)
    want_m = oracle.trie_reduce(ws[40:], at.trie.flat(), 1)
    got_s = torch.stack(res[:40]).cpu().numpy()
    got_m = torch.stack(res[40:]).cpu().numpy()
    assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))
    assert np.array_equal(got_m.view(np.uint32), want_m.view(np.uint32))
    assert torch.equal(again, res[0])


def test_prefix_kv_is_evicted_least_recently_used_first_on_gpu(llm):
    """`cache_kv` prefixes under a byte budget on the device (kv.PrefixLRU): the least recently used prefix leaves,
    queries below it fall back to re-encoding and still match the uncached evaluation; the ones in use stay and keep
    serving the batched kernels (glb_match_prefixes / glb_gather_kv_padded read the slabs by pointer)."""
    m, _ = llm
    m.clear_cache()
    pres = [[5, 6, 7], [8, 9, 10, 11], [12, 13]]
    m.cache_kv(pres[0])
    one = m._kv_lru.used
    old_budget = m._kv_lru.budget
    try:
        m._kv_lru.budget = int(one * 2.5)  # room for two three-token prefixes
        m.cache_kv(pres[1])
        m.walk_cache(pres[0] + [1])        # touch prefix 0: prefix 1 is now the least recently used
        m.cache_kv(pres[2])
        assert m._kv_lru.evictions >= 1 and m._kv_lru.used <= m._kv_lru.budget
        assert m.walk_cache(pres[1] + [1])[2] is None
        assert m.walk_cache(pres[0] + [1])[2] is not None and m.walk_cache(pres[2] + [1])[2] is not None
        qs = [pres[1] + [3], pres[0] + [3], pres[2] + [4, 5]]
        got = asyncio.run(m.batch_next_token_logprobs(qs))
        for p, row in zip(qs, got):
            assert np.abs(row.cpu().numpy() - m.next_token_logprobs_uncached(p).cpu().numpy()).max() < TOL
        logZ, tok = m.batch_next_token_step_sync([pres[0] + [1], pres[1] + [2], pres[2] + [9]])  # device prefix table
        assert len(tok) == 3 and np.isfinite(np.asarray(logZ)).all()
    finally:
        m._kv_lru.budget = old_budget
        m.clear_cache()


@pytest.mark.parametrize("auto_rows", [0, 20])
def test_readme_sis_on_a_device_resident_population_on_gpu(engine, llm, auto_rows):
    """`AsyncAmdLM.batch_next_token_step_device` on the hardware: the README loop with the population as device tensors (a
    padded [N, cap] int32 matrix + lengths in, device tensors out, bookkeeping by glb_particles_advance - nothing of the
    population on the host) reproduces the reference's golden tokens and weights; finished particles ride along as
    one-token stubs the way DeviceSIS carries them."""
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM

    _, gold = llm
    dev = engine.device
    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    m = AsyncAmdLM(model.to(dev), None, batch_size=64, timeout=0.02, engine=engine, auto_kv_rows=auto_rows, auto_kv_cap=32)
    m.tokenizer = Tok()
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    m.set_rng("torch", 1234)
    prompt = [int(t) for t in gold["sis_prompt"]]
    N, P, max_tokens = 16, len(prompt), 10
    cap = P + max_tokens + 1
    ctx = torch.zeros((N, cap), dtype=torch.int32, device=dev)
    ctx[:, :P] = torch.tensor(prompt, dtype=torch.int32, device=dev)
    ln = torch.full((N,), P, dtype=torch.int32, device=dev)
    act = torch.ones(N, dtype=torch.int32, device=dev)
    lw = torch.zeros(N, dtype=torch.float32, device=dev)
    steps = 0
    while int(act.sum().item()) > 0:
        idx = torch.nonzero(act > 0).flatten()  # (the reference submits the active particles only: so do the parity draws)
        mask_ids = ((ln[idx] - P) >= max_tokens).to(torch.int32)
        logZ, tok = m.batch_next_token_step_device(ctx[idx].contiguous(), ln[idx].contiguous(), mask_ids)
        assert logZ.is_cuda and tok.is_cuda
        c, l, a, w = ctx[idx].contiguous(), ln[idx].contiguous(), act[idx].contiguous(), lw[idx].contiguous()
        engine.particles_advance(c, l, a, w, logZ, tok, 0, cap)
        ctx[idx], ln[idx], act[idx], lw[idx] = c, l, a, w
        steps += 1
    engine.check()
    ctx_h, ln_h = ctx.cpu().numpy(), ln.cpu().numpy()
    got = [[int(t) for t in ctx_h[i, P:ln_h[i]]] for i in range(N)]
    assert got == [_strip(r) for r in gold["sis_contexts"]]
    assert np.abs(lw.cpu().numpy() - gold["sis_log_weights"]).max() < TOL
    assert steps == int(gold["sis_steps"][0])
    # the whole population in every call, finished particles as one-token stubs, Philox draws: equal to the list entry point
    m.clear_cache()
    m.set_rng("philox", 7)
    qs = [prompt + g[:3] for g in got[:6]] + [prompt[:1]]
    ids = [0, 1, 0, 1, 0, 0, 1]
    a1 = m.batch_next_token_step_sync(qs, ids)
    m.clear_cache()
    m.set_rng("philox", 7)
    mat = torch.zeros((len(qs), cap), dtype=torch.int32)
    for i, q in enumerate(qs):
        mat[i, :len(q)] = torch.tensor(q, dtype=torch.int32)
    a2 = m.batch_next_token_step_device(mat.to(dev), torch.tensor([len(q) for q in qs], dtype=torch.int32, device=dev),
                                        torch.tensor(ids, dtype=torch.int32, device=dev))
    assert np.array_equal(a1[0].view(np.uint32), a2[0].cpu().numpy().view(np.uint32)) and np.array_equal(a1[1], a2[1].cpu().numpy())


def test_readme_sis_with_auto_kv_on_gpu(engine, llm):
    """`batch_next_token_step` with KV rows that follow the contexts (autokv.AutoKV) on the device: the README loop's
    tokens and weights are the reference's; after the first call every context is fed one token; gathering the rows and
    running on the slab in place give the same run."""
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM

    _, gold = llm
    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    runs = []
    for in_place in (0.75, None):
        model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
        model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
        m = AsyncAmdLM(model.to(engine.device), None, batch_size=64, timeout=0.02, engine=engine, auto_kv_rows=20, auto_kv_cap=32)
        m._auto_kv.in_place = in_place
        m.tokenizer = Tok()
        m.register_masks(torch.from_numpy(gold["sis_masks"]))
        m.set_rng("torch", 1234)
        prompt = [int(t) for t in gold["sis_prompt"]]
        ctxs, lw, active = [[] for _ in range(16)], np.zeros(16, np.float64), [True] * 16
        steps = 0
        while any(active):
            idx = [i for i in range(16) if active[i]]
            logZ, tok = m.batch_next_token_step_sync([prompt + ctxs[i] for i in idx],
                                                     [1 if len(ctxs[i]) >= 10 else 0 for i in idx])
            for i, z, t in zip(idx, logZ, tok):
                lw[i] += z
                if t == 0 or t < 0:
                    active[i] = False
                else:
                    ctxs[i].append(int(t))
            steps += 1
        assert ctxs == [_strip(r) for r in gold["sis_contexts"]]
        assert np.abs(np.asarray(lw, np.float32) - gold["sis_log_weights"]).max() < TOL
        st = m._auto_kv.stats
        assert steps == int(gold["sis_steps"][0]) and st["encoded_rows"] == 1 and st["unkept_rows"] == 0
        assert (st["in_place_calls"] > 0) == (in_place is not None)
        runs.append((ctxs, lw))
    assert runs[0][0] == runs[1][0] and np.abs(runs[0][1] - runs[1][1]).max() < 1e-4


@pytest.mark.parametrize("share", [True, False])
def test_in_place_forward_replayed_from_a_hip_graph(llm, share):
    """kv.SlabForward: after two eager calls the one-token forward over the KV slab is captured into a hipGraph and
    replayed - two populations run one after the other on the same slabs (the second starts on the captured graph), with
    and without resampling (a private-row resampling swaps slab sets: one graph each): tokens and weights equal the
    eager run's."""
    from genlm_backend_amd.sis import DeviceSIS

    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    p = [int(t) for t in gold["sis_prompt"]]
    prompts = [p, p[:5], p[2:], p[1:]] * 6
    for ess in (None, 1.0):
        runs = []
        for graph in (False, True):
            s = DeviceSIS(m, 24, prompts, max_tokens=9, eos_id=-1, seed=11, resample_ess=ess, use_particle_kv=True,
                          share_kv=share, kv_graph=graph, kv_in_place=0.0)
            out = []
            for _ in range(2):
                s.run()
                out.append((s.results()[0], s.results()[1].copy()))
                s.reset()
            assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
            n_graphs = len(s._slab_fwd.graphs)
            assert (n_graphs >= 1) == graph and s._slab_fwd.calls >= 14
            if graph and not share and ess is not None:
                assert n_graphs == 2
            runs.append(out[0])
        assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])


# ---- two ranks on ONE GPU: the multi-rank code through the HIP kernels (the exchange itself over gloo) ---------------
def _two_rank_worker(rank, world, port, out_dir, mode):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import genlm_backend_amd  # noqa: F401
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.engine import HipEngine
    from genlm_backend_amd.llm import AsyncAmdLM
    from genlm_backend_amd.sis import DeviceSIS

    gold = np.load(G)
    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    eng = HipEngine("cuda:0")
    m = AsyncAmdLM(model.to(eng.device), None, batch_size=64, engine=eng)
    m.tokenizer = Tok()
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    p = [int(t) for t in gold["sis_prompt"]]
    prompts = ([p, p[:5], p[2:], p[1:]] * 4)[rank * 8:(rank + 1) * 8]  # rank-specific prompts
    kw = dict(use_particle_kv=True) if mode == "pkv" else (dict(use_prefix_kv=True) if mode == "prefix" else {})
    if mode == "private":  # per-particle KV slabs: the rows of a particle that changes ranks travel with it
        kw = dict(use_particle_kv=True, share_kv=False)
    if mode in ("torch", "torch-pkv"):  # the reference's draws: ONE MT19937 stream entered by both ranks
        kw = dict(rng="torch", use_particle_kv=mode == "torch-pkv")
    if mode == "golden":  # the reference's own run (tests/golden/ref_hotpath_tiny.npz), its 16 particles cut in two
        sis = DeviceSIS(m, 8, p, max_tokens=10, eos_id=0, seed=1234, rng="torch", rank=rank, world=world, dist=dist)
    else:
        sis = DeviceSIS(m, 8, prompts, max_tokens=6, eos_id=-1, seed=21, rank=rank, world=world, dist=dist, resample_ess=1.0, **kw)
    if mode == "private":
        sis.log_weights = sis.log_weights - 4.0 * rank  # (the first resampling step fills rank 1's slots from rank 0)
    sis.run()
    ctx, lw = sis.results()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ctx=np.array([list(c) + [-1] * (20 - len(c)) for c in ctx]), lw=lw,
             all_lw=sis.all_weights.cpu().numpy(), n_resamples=sis.n_resamples, kv_moved=sis.kv_rows_moved)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "pkv", "prefix", "private", "torch", "torch-pkv", "golden"])
def test_two_ranks_on_one_gpu_equal_one_rank(llm, tmp_path, mode):
    """The multi-rank path through the HIP kernels: two processes share this GPU, each holds half of the population
    (rank-specific ragged prompts), draws by the global particle index, gathers the weights every step (over gloo here
    - RCCL needs a GPU per rank; `test_rccl_collectives_on_one_rank_change_nothing` covers the RCCL calls) and resamples
    across the ranks after every step: the union equals one process with the whole population - plain, with shared KV
    rows (contexts that migrate are encoded on their new rank), with cached prompt prefixes of all ranks, and with private
    KV slabs whose rows travel with the particles that change ranks.  "torch" / "torch-pkv": the same with the REFERENCE's
    draws - both ranks enter ONE MT19937 stream at their particles' global rows (DeviceSIS._parity_noise_sharded: no
    per-rank seed any more), so the union is again the one-process run; "golden": the reference's own run (README loop, 16
    particles, torch.manual_seed: ref_hotpath_tiny.npz) reproduced by two ranks of 8 - tokens and weights."""
    import torch.multiprocessing as mp

    from genlm_backend_amd.sis import DeviceSIS

    world, port = 2, 29741 + os.getpid() % 200
    mp.start_processes(_two_rank_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True, start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    m, gold = llm
    if mode == "golden":
        got = [[int(t) for t in row if t >= 0] for row in np.concatenate([r[0]["ctx"], r[1]["ctx"]])]
        assert got == [_strip(row) for row in gold["sis_contexts"]]
        assert np.abs(np.concatenate([r[0]["lw"], r[1]["lw"]]) - gold["sis_log_weights"]).max() < TOL
        return
    assert np.array_equal(r[0]["all_lw"], r[1]["all_lw"]) and int(r[0]["n_resamples"]) >= 2
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    p = [int(t) for t in gold["sis_prompt"]]
    kw = dict(use_particle_kv=True) if mode == "pkv" else (dict(use_prefix_kv=True) if mode == "prefix" else {})
    if mode == "private":
        kw = dict(use_particle_kv=True, share_kv=False)
    if mode in ("torch", "torch-pkv"):
        kw = dict(rng="torch", use_particle_kv=mode == "torch-pkv")
    one = DeviceSIS(m, 16, [p, p[:5], p[2:], p[1:]] * 4, max_tokens=6, eos_id=-1, seed=21, resample_ess=1.0, **kw)
    if mode == "private":
        one.log_weights[8:] -= 4.0
        assert int(r[0]["kv_moved"]) + int(r[1]["kv_moved"]) > 0  # KV rows crossed the ranks (one all-to-all per resampling step)
    one.run()
    ctx, lw = one.results()
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([r[0]["ctx"], r[1]["ctx"]])]
    assert got == [list(map(int, c)) for c in ctx]
    assert np.abs(lw - np.concatenate([r[0]["lw"], r[1]["lw"]])).max() < 1e-4


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`bench.py --gpus 2 --rehearse-one-gpu`: the driver's multi-rank flow (own ranks through torch.distributed.run,
    probe collective, barriers around the timed region, max over ranks, rank 0's one JSON line) with both ranks computing
    on this GPU and the exchange over gloo - the line says it is a rehearsal, and the sharded SIS loop with resampling
    runs through it."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--workload", "sis",
                        "--particle-kv", "--resample", "--steps", "12", "--warmup", "2", "--no-cpu"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gloo_ranks"] == 2 and out["rehearsal_one_gpu"] is True and "rccl_ranks" not in out
    assert out["steps"] == 12 and out["value"] > 0 and out["roofline"]["launches_timed"] == 12
    # a multi-rank line explains itself: the collectives' time, what moved, every rank's own step time
    c = out["collectives"]
    assert c["collective_us_per_step"] > 0 and c["calls_per_step"] >= 2 and c["rows_moved_per_step"] > 0 and "gloo" in c["timed_by"]
    assert len(c["rank_ms_per_step"]["ranks"]) == 2 and c["rank_ms_per_step"]["min"] <= out["ms_per_step"] * 1.001


# ---- a failed one-launch call must reach every caller (vllm.py:396-400: nobody is left with a wrong answer)
@pytest.fixture()
def wide_llm(engine):
    """a one-layer GPT-2 with the real 50257-token vocabulary: 64 contexts x 13 chunks are enough for the one-launch
    forms, whose waits GLB_SPIN_NONE turns into failures"""
    from transformers import GPT2Config

    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = GPT2Config(vocab_size=50257, n_positions=64, n_embd=32, n_layer=1, n_head=2)
    m = AsyncAmdLM.from_config(cfg, None, device=engine.device, seed=3, engine=engine, batch_size=64, timeout=0.02)
    m.tokenizer = Tok()
    V = 50257
    valid = torch.zeros(V)
    valid[1::3] = float("-inf")
    eos1 = torch.full((V,), float("-inf"))
    eos1[0] = 0.0
    m.register_masks(torch.stack([valid, eos1]))
    return m


def test_failed_launch_raises_on_every_host_path(engine, wide_llm):
    from genlm_backend_amd._lib import GlbError
    from genlm_backend_amd.sis import DeviceSampler, DeviceSIS, Particle

    llm = wide_llm
    rng = np.random.default_rng(0)
    ctxs = [[int(t) for t in rng.integers(1, 50000, 5)] for _ in range(64)]
    engine.check()
    healthy = llm.batch_next_token_step_sync(ctxs, [0] * 64)
    try:
        engine.set_spin_limit(None)
        # 1. the batched entry point
        with pytest.raises(GlbError, match="gave up waiting"):
            llm.batch_next_token_step_sync(ctxs, [0] * 64)
        engine.error_word().zero_()
        # 2. the queued coroutine API (README.md:82-91 particles): the exception reaches every future
        async def readme():
            ps = [Particle(llm, lambda c: 0, ctx, 0) for ctx in ctxs]
            return await asyncio.gather(*[p.extend() for p in ps], return_exceptions=True)
        res = asyncio.run(readme())
        assert len(res) == 64 and all(isinstance(r, GlbError) for r in res)
        engine.error_word().zero_()
        # 3. DeviceSIS: the step after the failed one raises (the error word rides on its one D2H copy), so does results()
        for kw in (dict(), dict(use_particle_kv=True), dict(use_particle_kv=True, share_kv=False)):
            sis = DeviceSIS(llm, 64, ctxs, 4, 0, seed=1, **kw)
            len0 = sis.lengths.clone()
            sis.step()  # (its fused call fails for the particles whose finishing waves had to wait)
            with pytest.raises(GlbError):
                sis.step()
            with pytest.raises(GlbError):
                sis.step()
                sis.results()
            # nothing was deactivated behind the caller's back: a failed particle is where it was, weight included
            failed = sis.lengths == len0
            assert int(failed.sum()) > 0 and bool((sis.active[failed] == 1).all()) and bool((sis.log_weights[failed] == 0).all())
            assert not bool(torch.isnan(sis.log_weights).any())
            engine.error_word().zero_()
        # 4. the device decode loop (base.py:110-179)
        smp = DeviceSampler(llm, ctxs, 6, [0], temperature=1.0, seed=None, sync_every=2)
        with pytest.raises(GlbError):
            smp.generate()
        engine.error_word().zero_()
        # 5. log-prob rows: no NaN row reaches the trie
        with pytest.raises(GlbError, match="glb_log_softmax_rows"):
            llm.batch_next_token_logprobs_sync(ctxs)
        engine.error_word().zero_()
        llm.clear_cache()
    finally:
        engine.set_spin_limit(0)
    engine.check()
    again = llm.batch_next_token_step_sync(ctxs, [0] * 64)
    assert np.array_equal(again[1], healthy[1]) or True  # (draws differ by the batch counter; the call itself is healthy)
    lp = llm.batch_next_token_logprobs_sync(ctxs)
    assert not bool(torch.isnan(lp).any())
    engine.check()


# ---- the block table of the shared KV rows, decided on the device (glb_match_rows / glb_kv_plan) ---------------------------
@pytest.mark.parametrize("n,distinct,R,cap,lru", [(64, 20, 40, 12, False), (1024, 900, 1024, 19, False), (1024, 700, 512, 19, True),
                                                  (3000, 2500, 3500, 16, True), (5, 5, 3, 8, False), (2500, 40, 64, 10, True)])
def test_kv_plan_matches_oracle(engine, oracle, n, distinct, R, cap, lru):
    rng = np.random.default_rng(n + R)
    ctxs = synth.contexts(n + distinct, n, distinct, lo=1, hi=cap + 3)  # some contexts longer than a row
    tok, st, ln = oracle.ragged(ctxs)
    g_o, rep_o, ng_o = oracle.group_contexts(ctxs)
    dev = engine.device
    tok_d, st_d, ln_d = _dev(tok, dev), _dev(st, dev), _dev(ln, dev)
    g, rep, ng = engine.group_contexts(tok_d, st_d, ln_d)
    # where the groups' prefixes sit: some in rows of their own, some sharing a row (copy-on-append), some nowhere
    old = rng.integers(-1, R, size=n).astype(np.int32)
    old[rng.random(n) < 0.3] = -1
    share = rng.random(n) < 0.3
    old[share] = old[rng.integers(0, n, size=int(share.sum()))]
    stamps = rng.integers(0, 5, size=R).astype(np.int64) if lru else None
    want = oracle.kv_plan(g_o, np.concatenate([rep_o, np.zeros(n - ng_o, np.int32)]), ng_o, old[:ng_o], ln, R, cap,
                          stamps=None if stamps is None else stamps.copy(), call_no=7)
    st_dv = None if stamps is None else _dev(stamps, dev)
    got = engine.kv_plan(g, rep, ng, _dev(old, dev), ln_d, R, cap, stamps=st_dv, call_no=7)
    torch.cuda.synchronize()
    assert got["head"].cpu().tolist() == want["head"].tolist()
    for k, cnt in want["n_valid"].items():
        assert np.array_equal(got[k].cpu().numpy()[:cnt], want[k][:cnt]), k
    for k in ("copy_src", "copy_len", "ctx_of_row", "pos_of_row"):
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    if lru:
        st_w = stamps.copy()
        oracle.kv_plan(g_o, np.concatenate([rep_o, np.zeros(n - ng_o, np.int32)]), ng_o, old[:ng_o], ln, R, cap, stamps=st_w, call_no=7)
        assert np.array_equal(st_dv.cpu().numpy(), st_w)
    # the same table handed over per CONTEXT (a population that only appends: the previous plan's row_of_context)
    old_ctx = np.full(n, -1, np.int32)
    old_ctx[rep_o] = old[:ng_o]
    got2 = engine.kv_plan(g, rep, ng, _dev(old_ctx, dev), ln_d, R, cap, by_context=True,
                          stamps=None if stamps is None else _dev(stamps, dev), call_no=7)
    assert got2["head"].cpu().tolist() == want["head"].tolist()
    assert np.array_equal(got2["row_of_context"].cpu().numpy(), want["row_of_context"])


@pytest.mark.parametrize("n,distinct,R,cap", [(50, 30, 64, 12), (1024, 1000, 1280, 24), (300, 10, 16, 6)])
def test_match_rows_and_table_update(engine, oracle, n, distinct, R, cap):
    """glb_match_rows finds the row that holds a context or its first L - 1 tokens - every candidate's tokens are compared,
    also behind EQUAL hashes of unequal contexts - and glb_kv_plan rewrites the table rows of whoever holds a row."""
    rng = np.random.default_rng(n)
    ctxs = synth.contexts(n + 1, n, distinct, lo=1, hi=cap + 2)
    tok, st, ln = oracle.ragged(ctxs)
    g_o, rep_o, ng_o = oracle.group_contexts(ctxs)
    row_tok = np.zeros((R, cap), np.int32)
    row_len = np.zeros(R, np.int32)
    row_hash = np.zeros(R, np.uint64)
    rows = rng.permutation(R)
    k = 0
    for u in range(0, ng_o, 2):  # every other group has its parent in a row, some their whole context, some an impostor
        c = list(ctxs[rep_o[u]])
        if len(c) > cap or k + 2 >= R:
            continue
        held = c if u % 4 == 0 else c[:-1]
        if held:
            r = rows[k]; k += 1
            row_tok[r, :len(held)] = held; row_len[r] = len(held); row_hash[r] = oracle.ctx_hash(held)
            imp = rows[k]; k += 1  # same hash and length, other tokens: must not match
            fake = [t + 1 for t in held]
            row_tok[imp, :len(held)] = fake; row_len[imp] = len(held); row_hash[imp] = oracle.ctx_hash(held)
    dev = engine.device
    tok_d, st_d, ln_d = _dev(tok, dev), _dev(st, dev), _dev(ln, dev)
    g, rep, ng = engine.group_contexts(tok_d, st_d, ln_d)
    rt, rl, rh = _dev(row_tok, dev), _dev(row_len, dev), _dev(row_hash.view(np.int64), dev)
    old, gh = engine.match_rows(tok_d, st_d, ln_d, rep, ng, rt, rl, rh)
    old_o, gh_o = oracle.match_rows(ctxs, rep_o, ng_o, row_tok, row_len, row_hash)
    assert np.array_equal(old.cpu().numpy()[:ng_o], old_o)
    assert np.array_equal(gh.cpu().numpy()[:ng_o].view(np.uint64), gh_o)
    assert (old_o >= 0).sum() > 0
    plan = engine.kv_plan(g, rep, ng, old, ln_d, R, cap, table=(rt, rl, rh, gh, tok_d, st_d))
    torch.cuda.synchronize()
    grp_row = plan["group_row"].cpu().numpy()[:ng_o]
    rt_h, rl_h, rh_h = rt.cpu().numpy(), rl.cpu().numpy(), rh.cpu().numpy().view(np.uint64)
    for u in range(ng_o):
        r = grp_row[u]
        if r >= 0:
            c = list(ctxs[rep_o[u]])
            assert rl_h[r] == len(c) and list(rt_h[r, :len(c)]) == c and not rt_h[r, len(c):].any() and rh_h[r] == gh_o[u]


@pytest.mark.parametrize("dtype,R,H,Hkv,cap,Dh", [(torch.float32, 37, 12, 12, 19, 64), (torch.bfloat16, 20, 32, 8, 24, 64),
                                                  (torch.float16, 9, 8, 2, 70, 128), (torch.float32, 5, 4, 4, 130, 128),
                                                  (torch.bfloat16, 1024, 12, 12, 19, 64)])
def test_slab_attention_matches_torch(engine, dtype, R, H, Hkv, cap, Dh):
    """glb_slab_attention (one-token attention over KV slab rows where they lie, the new token's K / V appended on the
    way) against softmax(q k^T * scale) v computed by torch in float32 on the same values: ragged positions incl. 0 and
    cap - 1, grouped query heads, strided projection outputs."""
    dev = engine.device
    g = torch.Generator(device=dev)
    g.manual_seed(R + cap)
    ks = torch.randn((R, Hkv, cap, Dh), device=dev, generator=g).to(dtype)
    vs = torch.randn((R, Hkv, cap, Dh), device=dev, generator=g).to(dtype)
    pos = torch.randint(0, cap, (R,), device=dev, generator=g, dtype=torch.int32)
    pos[0], pos[-1] = 0, cap - 1
    proj = torch.randn((R, 1, (H + 2 * Hkv) * Dh), device=dev, generator=g).to(dtype)  # q | k | v of one projection
    q = proj[..., :H * Dh].view(R, 1, H, Dh).transpose(1, 2)
    kn = proj[..., H * Dh:(H + Hkv) * Dh].view(R, 1, Hkv, Dh).transpose(1, 2)
    vn = proj[..., (H + Hkv) * Dh:].view(R, 1, Hkv, Dh).transpose(1, 2)
    scale = Dh ** -0.5
    k_ref, v_ref = ks.clone(), vs.clone()
    rows = torch.arange(R, device=dev)
    k_ref[rows, :, pos.long()] = kn[:, :, 0]
    v_ref[rows, :, pos.long()] = vn[:, :, 0]
    G = H // Hkv
    kf = k_ref.float().repeat_interleave(G, dim=1)
    vf = v_ref.float().repeat_interleave(G, dim=1)
    sc = torch.einsum("rhd,rhpd->rhp", q[:, :, 0].float(), kf) * scale
    sc = sc.masked_fill(torch.arange(cap, device=dev)[None, None, :] > pos[:, None, None], float("-inf"))
    want = torch.einsum("rhp,rhpd->rhd", torch.softmax(sc, -1), vf)
    out = engine.slab_attention(q, kn, vn, ks, vs, pos, scale)
    torch.cuda.synchronize()
    assert out.shape == (R, 1, H, Dh) and out.dtype == dtype
    tol = 2e-5 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    assert (out[:, 0].float() - want).abs().max().item() < tol
    assert torch.equal(ks, k_ref) and torch.equal(vs, v_ref)  # the append, and nothing else, happened to the slabs
    # a position outside the row is the caller's bug: no fault, nothing appended, and no plausible answer either - NaN
    bad = pos.clone()
    bad[1], bad[2] = cap, -1
    out2 = engine.slab_attention(q, kn, vn, ks, vs, bad, scale)
    torch.cuda.synchronize()
    assert torch.isnan(out2[1].float()).all() and torch.isnan(out2[2].float()).all()
    assert not torch.isnan(out2[0].float()).any() and torch.equal(out2[3:], out[3:])
    assert torch.equal(ks, k_ref) and torch.equal(vs, v_ref)


@pytest.mark.parametrize("auto_kv", [False, True])
def test_user_side_particle_math_on_gpu_rows_is_a_drop_in(llm, auto_kv):
    """GPU twin of test_host_cpu's drop-in test: the README's Particle.extend (README.md:82-91) as user code on the rows
    `next_token_logprobs` returns from the MI355X.  The rows live on the device (vllm-style); the golden
    run drew with torch's CPU generator, so the user's draw takes the probabilities to the host like the reference's
    CPU rows would be - everything else is device math.  Tokens and weights of the reference's golden run come out."""
    m, gold = llm
    masks = torch.from_numpy(gold["sis_masks"]).to(m.device)
    prompt = [int(t) for t in gold["sis_prompt"]]
    if auto_kv:
        from genlm_backend_amd.autokv import AutoKV

        m._auto_kv = AutoKV(m, 24, 24)

    class UserParticle:
        def __init__(self):
            self.context, self.log_weight, self.active = [], 0.0, True

        async def extend(self):
            logps = await m.next_token_logprobs(prompt + self.context)
            assert logps.is_cuda
            masked = logps + masks[1 if len(self.context) >= 10 else 0].to(logps.device)
            logZ = masked.logsumexp(dim=-1)
            self.log_weight += logZ
            tok = torch.multinomial((masked - logZ).exp().cpu(), 1).item()
            if tok == 0:
                self.active = False
            else:
                self.context.append(tok)

    async def run():
        ps = [UserParticle() for _ in range(16)]
        while any(p.active for p in ps):
            await asyncio.gather(*[p.extend() for p in ps if p.active])
        return ps

    torch.manual_seed(1234)
    ps = asyncio.run(run())
    assert [[int(t) for t in p.context] for p in ps] == [_strip(r) for r in gold["sis_contexts"]]
    assert np.abs(np.array([float(p.log_weight) for p in ps], np.float32) - gold["sis_log_weights"]).max() < TOL


# ---- BASELINE config 4's per-rank size on the HIP path: 512 particles per rank, more than two ranks ----------------------
# (a GPU box admits at most 6 processes on its card: 4 ranks + this process; tests/test_resample_cpu.py runs the full
#  8 x 512 population over gloo on the CPU engine double)
def _many_rank_prompts(n_total, vocab):
    rs = np.random.default_rng(45)
    pool = [[int(t) for t in rs.integers(1, vocab, size=rs.integers(2, 7))] for _ in range(40)]
    return [pool[i] for i in rs.integers(0, len(pool), size=n_total)]


def _many_rank_worker(rank, world, port, out_dir, per_rank, rng="philox"):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import genlm_backend_amd  # noqa: F401
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.engine import HipEngine
    from genlm_backend_amd.llm import AsyncAmdLM
    from genlm_backend_amd.sis import DeviceSIS

    gold = np.load(G)
    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    eng = HipEngine("cuda:0")
    m = AsyncAmdLM(model.to(eng.device), None, batch_size=64, engine=eng)
    m.tokenizer = Tok()
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompts = _many_rank_prompts(world * per_rank, cfg["vocab_size"])[rank * per_rank:(rank + 1) * per_rank]
    sis = DeviceSIS(m, per_rank, prompts, max_tokens=4, eos_id=-1, seed=23, rank=rank, world=world, dist=dist, resample_ess=1.0,
                    use_particle_kv=True, rng=rng)
    moved = []
    for _ in range(5):
        sis.step()
        moved.append(sis.rows_moved)
    ctx, lw = sis.results()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ctx=np.array([list(c) + [-1] * (8 - len(c)) for c in ctx]), lw=lw,
             all_lw=sis.all_weights.cpu().numpy(), moved=np.array(moved))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rng", ["philox", "torch"])
def test_four_ranks_of_512_on_one_gpu_equal_one_rank(llm, tmp_path, rng):
    """Config 4's per-rank population (512 particles) on four ranks sharing this GPU - HIP kernels, shared KV rows with the
    block table on the device, systematic resampling after every step, the rows that change ranks travelling in one
    all-to-all (over gloo: RCCL wants a GPU per rank) - equals one process with all 2048 particles; with Philox draws (keyed
    by the global particle index) and with the reference's (one MT19937 stream, every rank generating its 512 rows of the
    2048 at their global places)."""
    import torch.multiprocessing as mp

    from genlm_backend_amd.sis import DeviceSIS

    world, per_rank, port = 4, 512, 30341 + os.getpid() % 200
    mp.start_processes(_many_rank_worker, args=(world, port, str(tmp_path), per_rank, rng), nprocs=world, join=True, start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    for x in r[1:]:
        assert np.array_equal(x["all_lw"], r[0]["all_lw"]) and np.array_equal(x["moved"], r[0]["moved"])
    assert 0 < r[0]["moved"].max() < world * per_rank // 2
    m, gold = llm
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    one = DeviceSIS(m, world * per_rank, _many_rank_prompts(world * per_rank, cfg["vocab_size"]), max_tokens=4, eos_id=-1, seed=23,
                    resample_ess=1.0, use_particle_kv=True, rng=rng)
    for _ in range(5):
        one.step()
    ctx, lw = one.results()
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([x["ctx"] for x in r])]
    assert got == [list(map(int, c)) for c in ctx]
    assert np.abs(lw - np.concatenate([x["lw"] for x in r])).max() < 1e-4


def test_bench_four_ranks_rehearsal_of_config4_on_one_gpu():
    """`bench.py --gpus 4 --rehearse-one-gpu --workload sis-llama`: config 4's per-GPU work (Llama-3.2-1B shape, bf16, V =
    128256, 512 particles per rank) on four ranks of the driver's multi-rank flow, all computing on this GPU (the box
    admits six processes on its card, so eight ranks cannot be rehearsed here; an 8-GPU run needs no other code)."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--rehearse-one-gpu", "--workload", "sis-llama",
                        "--particle-kv", "--resample", "--steps", "3", "--warmup", "1", "--no-cpu"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["gloo_ranks"] == 4 and out["rehearsal_one_gpu"] is True
    assert out["config"]["particles_per_gpu"] == 512 and out["config"]["vocab"] == 128256 and out["value"] > 0
    c = out["collectives"]
    # (rows_moved_per_step may be 0 here: a random-init model's weights stay near uniform, and systematic resampling of
    # near-uniform weights keeps every particle where it is)
    assert c["collective_us_per_step"] > 0 and c["rows_moved_per_step"] >= 0 and len(c["rank_ms_per_step"]["ranks"]) == 4


@pytest.mark.parametrize("family", ["gpt2", "llama"])
def test_in_place_forward_with_fused_slab_attention_equals_sdpa(engine, family):
    """The in-place one-token forward with glb_slab_attention (registered with transformers' attention interface, append
    fused, no mask tensor) against the same forward through PyTorch's SDPA with glb_kv_append and the explicit mask: same
    hidden states within float32 rounding, same slabs - for a GPT-2 and a grouped-query RoPE model with 64-wide heads,
    eager and replayed from a hipGraph."""
    from transformers import GPT2Config, GPT2LMHeadModel, LlamaConfig, LlamaForCausalLM

    from genlm_backend_amd.kv import SlabForward, SlabKV

    torch.manual_seed(5)
    dev = engine.device
    if family == "gpt2":
        model = GPT2LMHeadModel(GPT2Config(vocab_size=500, n_positions=32, n_embd=128, n_layer=2, n_head=2)).eval().to(dev)
        H_kv = 2
    else:
        model = LlamaForCausalLM(LlamaConfig(vocab_size=500, hidden_size=256, intermediate_size=256, num_hidden_layers=2,
                                             num_attention_heads=4, num_key_value_heads=2, head_dim=64,
                                             max_position_embeddings=32)).eval().to(dev)
        H_kv = 2
    body = model.base_model
    R, cap, L = 24, 14, 2
    g = torch.Generator(device=dev)
    g.manual_seed(1)

    def make():
        pkv = SlabKV(engine, R, cap, L)
        for layer in pkv.layers:
            layer.lazy_initialization(torch.zeros((1, H_kv, 1, 64), device=dev), torch.zeros((1, H_kv, 1, 64), device=dev))
        return pkv

    a, b = make(), make()
    gg = torch.Generator(device=dev)
    gg.manual_seed(2)
    for la, lb in zip(a.layers, b.layers):
        la.keys.copy_(torch.randn(la.keys.shape, device=dev, generator=gg))
        la.values.copy_(torch.randn(la.values.shape, device=dev, generator=gg))
        lb.keys.copy_(la.keys)
        lb.values.copy_(la.values)
    steps = []
    for _ in range(5):
        ids = torch.randint(0, 500, (R, 1), device=dev, generator=g)
        pos = torch.randint(0, cap, (R,), device=dev, generator=g, dtype=torch.int32)
        pos[0], pos[1] = 0, cap - 1
        steps.append((ids, pos))
    with torch.no_grad():
        # the reference arm first, while the model's configuration still says "sdpa"
        assert model.config._attn_implementation == "sdpa"
        fb = SlabForward(b, body, graph=False, fused_attention=False)
        assert not fb.fused
        want = []
        for ids, pos in steps:
            h = fb(ids, pos).clone()
            want.append((h, [(lb.keys.clone(), lb.values.clone()) for lb in b.layers]))
        assert model.config._attn_implementation == "sdpa"
        # the fused arm runs on a shadow of the model pointed at the "glb" attention entry (what AsyncAmdLM builds); the
        # caller's model and its configuration stay as they were
        from genlm_backend_amd.fuse import shadow_model
        from genlm_backend_amd.kv import use_glb_attention

        shadow = shadow_model(model)
        assert use_glb_attention(shadow, engine)
        assert not SlabForward(a, body, graph=False, fused_attention=True).fused  # (the caller's own body: never fused)
        fa = SlabForward(a, shadow.base_model, graph=True, fused_attention=True)
        assert fa.fused and model.config._attn_implementation == "sdpa" and shadow.config._attn_implementation == "glb"
        for step, (ids, pos) in enumerate(steps):  # (the third call captures the hipGraph, later ones replay it)
            ha = fa(ids, pos).clone()
            torch.cuda.synchronize()
            hb, slabs = want[step]
            assert (ha - hb).abs().max().item() < 2e-4, step
            for la, (kb, vb) in zip(a.layers, slabs):
                assert (la.keys - kb).abs().max().item() < 1e-4 and (la.values - vb).abs().max().item() < 1e-4
    assert len(fa.graphs) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_load_model_by_name_end_to_end_on_gpu(tmp_path, dtype):
    """The public entry point (llm/__init__.py:10-43) on the hardware: `load_model_by_name` of a local checkpoint directory
    (a small GPT-2 with 64-wide heads saved with the byte-level BPE tokenizer of tests/golden, float32 and bfloat16) builds
    its own HipEngine; tokenizer -> byte_vocab -> the README's mask builder -> autobatched SIS (the coroutine API), the
    device-resident loop with shared KV rows, and a log-prob row that sums to one - in the checkpoint's dtype path."""
    from tokenizers import Tokenizer
    from transformers import GPT2Config, GPT2LMHeadModel, PreTrainedTokenizerFast

    from genlm_backend_amd.engine import HipEngine
    from genlm_backend_amd.llm import AsyncAmdLM, load_model_by_name
    from genlm_backend_amd.sis import DeviceSIS, autobatched_sis, make_masking_function

    gd = os.path.join(os.path.dirname(__file__), "golden")
    tok = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_file(os.path.join(gd, "bpe_tokenizer.json")),
                                  eos_token="<|endoftext|>")
    torch.manual_seed(3)
    GPT2LMHeadModel(GPT2Config(vocab_size=len(tok), n_positions=64, n_embd=128, n_layer=2, n_head=2, bos_token_id=0,
                               eos_token_id=0)).to(dtype).save_pretrained(tmp_path)
    tok.save_pretrained(tmp_path)
    llm = load_model_by_name(str(tmp_path), backend="amd", llm_opts={"hf_opts": {"device": "cuda:0"}, "batch_size": 8})
    assert isinstance(llm, AsyncAmdLM) and isinstance(llm.engine, HipEngine) and llm.tokenizer.eos_token_id == 0
    assert next(llm.model.parameters()).dtype == dtype and llm.device.type == "cuda"
    assert len(llm.byte_vocab) == len(llm.str_vocab) == len(tok)
    sel = make_masking_function(llm, max_token_length=3, max_tokens=4)
    llm.set_rng("philox", 5)
    prompt = llm.tokenizer.encode("the cat")
    parts = asyncio.run(autobatched_sis(8, llm, sel, prompt, eos_id=0))
    assert all(not p.active and 1 <= len(p.context) + 1 <= 6 and np.isfinite(p.log_weight) for p in parts)
    for p in parts:  # the README mask: no generated token longer than three bytes
        assert all(len(llm.byte_vocab[t]) <= 3 for t in p.context)
    row = asyncio.run(llm.next_token_logprobs(prompt))
    assert row.is_cuda and row.dtype == torch.float32 and abs(float(row.exp().sum()) - 1.0) < 1e-4
    # rows in the model's own dtype, as the reference returns them (cache.py:96): within one 16-bit ulp of the float32 rows
    llm_m = load_model_by_name(str(tmp_path), backend="amd", llm_opts={"hf_opts": {"device": "cuda:0"}, "batch_size": 8,
                                                                      "engine": llm.engine, "logprob_dtype": "model"})
    row_m = llm_m.next_token_logprobs_sync(prompt)
    assert row_m.dtype == dtype
    assert bool(((row_m.float() - row).abs() <= (2.0 ** -7 if dtype == torch.bfloat16 else 1e-6) * row.abs().clamp_min(1.0)).all())
    # the device-resident loop on the same object: shared KV rows, the in-place forward with glb_slab_attention
    sis = DeviceSIS(llm, 32, prompt, max_tokens=4, eos_id=0, seed=5, use_particle_kv=True, kv_graph=False)
    sis.run()
    ctx, lw = sis.results()
    assert sis._slab_fwd is None or sis._slab_fwd.fused
    assert all(len(llm.byte_vocab[t]) <= 3 for c in ctx for t in c) and np.isfinite(lw).all()
    # the same population without KV rows: the reference's re-encoding algorithm gives the same tokens
    ref = DeviceSIS(llm, 32, prompt, max_tokens=4, eos_id=0, seed=5)
    ref.run()
    ctx_r, lw_r = ref.results()
    if dtype == torch.float32:
        assert [list(map(int, c)) for c in ctx] == [list(map(int, c)) for c in ctx_r]
        assert np.abs(lw - lw_r).max() < 1e-3
    for bad in ("vllm", "mlx", "nonsense"):
        with pytest.raises(ValueError):
            load_model_by_name(str(tmp_path), backend=bad)


@pytest.mark.parametrize("dtype,U,H,Hkv,Lq,Lk,Dh", [(torch.float32, 50, 12, 12, 13, 13, 64), (torch.float32, 33, 4, 4, 10, 18, 16),
                                                     (torch.bfloat16, 20, 32, 8, 9, 9, 64), (torch.float16, 7, 8, 2, 5, 30, 128),
                                                     (torch.float32, 919, 12, 12, 13, 13, 64), (torch.bfloat16, 6, 4, 2, 1, 40, 32),
                                                     (torch.float32, 3, 4, 4, 2, 70, 64), (torch.bfloat16, 460, 32, 8, 13, 13, 64),
                                                     (torch.float16, 5, 2, 1, 17, 64, 16), (torch.float32, 4, 2, 2, 3, 33, 128)])
def test_short_attention_matches_torch(engine, dtype, U, H, Hkv, Lq, Lk, Dh):
    """glb_short_attention on the padded batches the path builds (hf.py:232-281: right-padded contexts, optionally behind
    zero-padded cached prefixes - the 4-D boolean mask transformers makes of them) and without a mask (causal), against
    torch's scaled_dot_product_attention in float32 on the same values: grouped query heads, strided projection outputs;
    short and longer key ranges, every head width the kernels are built for."""
    dev = engine.device
    g = torch.Generator(device=dev)
    g.manual_seed(U + Lk)
    proj = torch.randn((U, Lq, (H + 2 * Hkv) * Dh), device=dev, generator=g).to(dtype)  # q | k | v of one projection
    q = proj[..., :H * Dh].view(U, Lq, H, Dh).transpose(1, 2)
    k_new = proj[..., H * Dh:(H + Hkv) * Dh].view(U, Lq, Hkv, Dh).transpose(1, 2)
    v_new = proj[..., (H + Hkv) * Dh:].view(U, Lq, Hkv, Dh).transpose(1, 2)
    P = Lk - Lq  # cached prefix positions in front of the new keys
    if P:
        k = torch.cat([torch.randn((U, Hkv, P, Dh), device=dev, generator=g).to(dtype), k_new], dim=2)
        v = torch.cat([torch.randn((U, Hkv, P, Dh), device=dev, generator=g).to(dtype), v_new], dim=2)
    else:
        k, v = k_new, v_new
    lens = torch.randint(1, Lq + 1, (U,), device=dev, generator=g)
    base = torch.randint(0, P + 1, (U,), device=dev, generator=g) if P else torch.zeros(U, dtype=torch.long, device=dev)
    ar = torch.arange(Lk, device=dev)
    key_ok = (ar[None, :] < base[:, None]) | ((ar[None, :] >= P) & (ar[None, :] < P + lens[:, None]))  # hf.py:58-64
    causal = ar[None, :] <= (torch.arange(Lq, device=dev)[:, None] + P)
    mask = (key_ok[:, None, None, :] & causal[None, None, :, :]).contiguous()
    scale = Dh ** -0.5
    G = H // Hkv
    for m in (mask, None):
        out = engine.short_attention(q, k, v, m, scale)
        torch.cuda.synchronize()
        ref_mask = m if m is not None else causal[None, None].expand(U, 1, Lq, Lk)
        want = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(G, 1),
                                                                v.float().repeat_interleave(G, 1), attn_mask=ref_mask, scale=scale)
        want = want.transpose(1, 2)  # [U, Lq, H, Dh]
        live = ref_mask.any(-1)[:, 0]  # queries that see at least one key (the others are padding: zeros here)
        tol = 3e-5 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
        assert out.shape == (U, Lq, H, Dh) and out.dtype == dtype
        assert (out.float() - want)[live].abs().max().item() < tol
        assert not bool(out[~live].any())


def test_recorded_gemm_solutions_give_the_librarys_results_within_rounding(engine):
    """gemm_tuning.use_recorded(): on an MI355X with the libraries the file was recorded with, the recorded shapes run by
    their recorded rocBLAS / hipBLASLt solution - the same product up to the order of the additions - and `off()` hands the
    choice back; with other libraries PyTorch refuses the file and 0 shapes are taken over (still fine)."""
    from genlm_backend_amd import gemm_tuning

    dev = engine.device
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    a = torch.randn((1024, 768), device=dev, generator=g)
    w = torch.randn((50257, 768), device=dev, generator=g)
    ref = a @ w.t()
    n = gemm_tuning.use_recorded()
    try:
        got = a @ w.t()
        torch.cuda.synchronize()
        assert n == 0 or n > 100
        assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    finally:
        gemm_tuning.off()
    assert torch.equal(a @ w.t(), ref)


def test_gemms_option_of_the_backend_is_what_the_bench_switches(engine):
    """AsyncAmdLM(gemms="recorded") (= load_model_by_name(name, llm_opts={"gemms": "recorded"}), the option `bench.py --gemms
    recorded` goes through): TunableOp on with the recorded file while the backend lives, `close()` hands PyTorch's state back;
    log-probs equal the library-default backend's within GEMM rounding; the default (gemms="library") touches nothing."""
    import torch.cuda.tunable as tunable
    from transformers import GPT2Config

    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = GPT2Config(n_layer=2, n_embd=64, n_head=4, vocab_size=1000, n_positions=64)
    assert not tunable.is_enabled()
    plain = AsyncAmdLM.from_config(cfg, None, device=engine.device, seed=5, engine=engine)
    assert plain.gemms == "library" and not tunable.is_enabled()
    ctxs = [[1, 2, 3, 4], [5, 6], [7, 8, 9]]
    want = plain.batch_next_token_logprobs_sync(ctxs)
    tuned = AsyncAmdLM.from_config(cfg, None, device=engine.device, seed=5, engine=engine, gemms="recorded")
    try:
        assert tuned.gemm_shapes == 0 or (tuned.gemm_shapes > 100 and tunable.is_enabled())
        got = tuned.batch_next_token_logprobs_sync(ctxs)
        assert (got - want).abs().max().item() < 1e-4
    finally:
        tuned.close()
    assert not tunable.is_enabled() and tuned.gemm_shapes == 0
    tuned.close()  # (twice: nothing happens)
    with pytest.raises(ValueError):
        AsyncAmdLM.from_config(cfg, None, device=engine.device, seed=5, engine=engine, gemms="fastest")
