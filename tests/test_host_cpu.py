"""Host logic of the backend (queue, device-side dedup order, trie, prefix KV, sampling, SIS) against
goldens produced by the REFERENCE itself (tests/golden/ref_hotpath_tiny.npz, made by
oracle/make_goldens.py).  Runs on CPU: the HIP engine is replaced by the oracle-backed test double."""
import ast
import asyncio
import os

import numpy as np
import pytest
import torch

from tests.cpu_engine import CpuOracleEngine

G = os.path.join(os.path.dirname(__file__), "golden", "ref_hotpath_tiny.npz")
TOL = 1e-4  # north_star: log-probs within 1e-4 of the reference's transformers-CPU path


class Tok:
    pad_token_id = None
    eos_token_id = 0


@pytest.fixture(scope="module")
def gold():
    return np.load(G)


@pytest.fixture()
def llm(gold):
    from transformers import GPT2Config, GPT2LMHeadModel

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")}
    model.load_state_dict(sd)
    m = AsyncAmdLM(model, None, batch_size=64, timeout=0.02, engine=CpuOracleEngine())
    m.tokenizer = Tok()
    return m


def _strip(row):
    return [int(t) for t in row if t >= 0]


def test_batched_logprobs_match_reference(llm, gold):
    prompts = [_strip(r) for r in gold["lp_prompts"]]
    got = asyncio.run(llm.batch_next_token_logprobs(prompts)).numpy()
    assert np.abs(got - gold["lp_values"]).max() < TOL
    assert np.abs(got - gold["lp_uncached"]).max() < TOL
    # duplicate prompts (rows 0 and 2) were evaluated once and are bit-identical (hf.py:214-220)
    assert np.array_equal(got[0], got[2])
    assert llm.stats["batches"] == 1 and llm.stats["queries"] == 5 and llm.stats["unique"] == 4
    # every position of every prompt was cached along the trie (cache.py:90-100, test_hf_llm.py:72-75)
    for p in prompts:
        node = llm.cache
        for t in p:
            assert node.has_token(t)
            node = node.get_token(t)
        assert node.logprobs is not None
    # sync / uncached variants agree (test_hf_llm.py:45-81)
    llm.clear_cache()
    for p, want in zip(prompts, gold["lp_values"]):
        assert np.abs(llm.next_token_logprobs_sync(p).numpy() - want).max() < TOL
        assert np.abs(llm.next_token_logprobs_uncached(p).numpy() - want).max() < TOL


def test_empty_input_raises(llm):
    with pytest.raises(ValueError):
        asyncio.run(llm.next_token_logprobs([]))
    with pytest.raises(ValueError):
        llm.next_token_logprobs_sync([])
    with pytest.raises(ValueError):
        llm.next_token_logprobs_uncached([])
    with pytest.raises(ValueError):
        asyncio.run(llm.next_token_step([]))


def test_cache_operations_and_queue(llm):
    p = [4, 5, 6]
    asyncio.run(llm.next_token_logprobs(p))
    node, n, past, base = llm.walk_cache(p)
    assert n == 3 and past is None and base == 0
    llm.clear_cache()
    node, n, past, base = llm.walk_cache(p)
    assert n == 0 and past is None and base == 0
    repr(llm.cache)

    async def pending():
        fut = asyncio.get_running_loop().create_future()
        llm.add_query(p, fut, None)
        assert len(llm.queries) == 1
        llm.reset_async_queries()
        assert len(llm.queries) == 0
        llm.timer.cancel()

    asyncio.run(pending())
    llm.queries = []
    llm.batch_evaluate_queries()  # empty batch is a no-op (test_hf_llm.py:248-252)


def test_timer_and_full_batch(llm):
    async def timer_fires():
        fut = asyncio.get_running_loop().create_future()
        llm.add_query([1, 2], fut, None)
        await asyncio.sleep(llm.timeout * 3)
        assert fut.done()

    asyncio.run(timer_fires())

    async def full_batch():
        llm.batch_size, llm.timeout = 2, 10
        await asyncio.wait_for(asyncio.gather(llm.next_token_logprobs([0]), llm.next_token_logprobs([1])), 5)

    asyncio.run(full_batch())


def test_exception_reaches_every_future(llm):
    async def run():
        def boom(*a, **k):
            raise RuntimeError("injected forward failure")

        llm._body = boom
        res = await asyncio.gather(llm.next_token_logprobs([1, 2]), llm.next_token_logprobs([3]),
                                   return_exceptions=True)
        assert all(isinstance(r, RuntimeError) for r in res)

    asyncio.run(run())


def test_prefix_kv_cache_matches_reference(llm, gold):
    pre = [int(t) for t in gold["kv_prefix"]]
    llm.cache_kv(pre)
    node, n, past, base = llm.walk_cache(pre)
    assert n == len(pre) and node.past_key_values is not None  # test_hf_llm.py:127-140
    node, n, past, base = llm.walk_cache(pre + [20])
    assert past is not None and base == len(pre) and n == len(pre)
    qs = [_strip(r) for r in gold["kv_queries"]]
    got = asyncio.run(llm.batch_next_token_logprobs(qs)).numpy()
    assert np.abs(got - gold["kv_values"]).max() < TOL
    assert np.abs(got - gold["kv_uncached"]).max() < TOL
    llm.clear_kv_cache()
    assert llm.walk_cache(pre + [20])[2] is None


def test_seeded_sample_matches_reference_ids(llm, gold):
    ids = asyncio.run(llm.sample([int(t) for t in gold["sample_prompt"]], max_tokens=12, eos_token_ids=[0],
                                 temperature=0.5, seed=80808))
    assert ids == [int(t) for t in gold["sample_ids"]]
    ids2 = asyncio.run(llm.batch_sample([[3, 1, 4, 1, 5], [9, 9]], max_tokens=6, eos_token_ids=[], temperature=1.0,
                                        seed=7))
    assert np.array_equal(np.array(ids2), gold["batch_sample_ids"])


G3 = os.path.join(os.path.dirname(__file__), "golden", "ref_round3.npz")


@pytest.mark.parametrize("sync_every", [1, 4])
def test_batch_sample_ragged_prompts_two_stop_tokens(llm, sync_every):
    """base.py:148-179 on ten ragged prompts, temperature 0.2, two stopping tokens, sequences ending after 4, 7 and 10
    tokens (oracle/make_goldens_r3.py ran the reference): every sequence races against the same Exp(1) row per step."""
    g3 = np.load(G3)
    prompts = [_strip(r) for r in g3["bs_prompts"]]
    ids = llm.batch_sample_sync(prompts, max_tokens=int(g3["bs_params"][0]), eos_token_ids=[int(t) for t in g3["bs_eos"]],
                                temperature=float(g3["bs_temperature"][0]), seed=int(g3["bs_params"][1]),
                                sync_every=sync_every)
    assert ids == [_strip(r) for r in g3["bs_ids"]]
    assert len({len(r) for r in ids}) >= 3


def _check_sis(contexts, log_weights, gold):
    want_ctx = [_strip(r) for r in gold["sis_contexts"]]
    assert [list(map(int, c)) for c in contexts] == want_ctx  # sampled ids bit-exact under the fixed seed
    assert np.abs(np.asarray(log_weights, np.float32) - gold["sis_log_weights"]).max() < TOL


def test_readme_sis_async_matches_reference(llm, gold):
    """BASELINE.json config 1: 16 particles, prompt len 8, <= 10 tokens, reference plumbing."""
    from genlm_backend_amd.sis import autobatched_sis

    llm.register_masks(torch.from_numpy(gold["sis_masks"]))
    llm.set_rng("torch", 1234)
    prompt = [int(t) for t in gold["sis_prompt"]]
    parts = asyncio.run(autobatched_sis(16, llm, lambda c: 1 if len(c) >= 10 else 0, prompt, eos_id=0))
    _check_sis([p.context for p in parts], [p.log_weight for p in parts], gold)
    # step 0 deduplicates 16 identical contexts to one forward row
    assert llm.stats["batches"] == int(gold["sis_steps"][0])
    assert llm.stats["queries"] > llm.stats["unique"]


def test_gather_without_tasks_gives_the_same_particles(llm, gold):
    """AsyncAmdLM.gather (the coroutines of a step advanced by hand instead of one asyncio Task each): the reference's golden
    tokens and weights, the same number of batches; coroutines that also await other things, return values in order,
    exceptions with and without return_exceptions."""
    from genlm_backend_amd.sis import autobatched_sis

    llm.register_masks(torch.from_numpy(gold["sis_masks"]))
    llm.set_rng("torch", 1234)
    prompt = [int(t) for t in gold["sis_prompt"]]
    parts = asyncio.run(autobatched_sis(16, llm, lambda c: 1 if len(c) >= 10 else 0, prompt, eos_id=0, gather=llm.gather))
    _check_sis([p.context for p in parts], [p.log_weight for p in parts], gold)
    assert llm.stats["batches"] == int(gold["sis_steps"][0])

    async def mixed():
        async def plain(i):
            return i

        async def sleeper(i):
            await asyncio.sleep(0)
            await asyncio.sleep(0.002)
            z, t = await llm.next_token_step(prompt + [i + 1], 0)
            return i, t

        async def stepper(i):
            z, t = await llm.next_token_step(prompt + [i + 1], 0)
            z2, t2 = await llm.next_token_step(prompt + [i + 1, 7], 0)
            return i, t

        async def failing():
            await llm.next_token_step(prompt, 0)
            raise KeyError("boom")

        before = llm.stats["batches"]
        res = await llm.gather(plain(0), sleeper(1), stepper(2), stepper(3), plain(4))
        assert res[0] == 0 and res[4] == 4 and [r[0] for r in res[1:4]] == [1, 2, 3]
        assert llm.stats["batches"] - before <= 6
        res = await llm.gather(plain(1), failing(), return_exceptions=True)
        assert res[0] == 1 and isinstance(res[1], KeyError)
        with pytest.raises(KeyError):
            await llm.gather(stepper(5), failing())
        return True

    llm.set_rng("torch", 99)
    assert asyncio.run(mixed())


@pytest.mark.parametrize("share", [False, True])
def test_device_sis_parity_draws_with_kv_rows_run_reset_run(llm, gold, share):
    """Parity draws (rng="torch") with per-particle KV rows, private and shared: the noise rows are dealt by this step's
    dedup grouping - also on step 0 and on a used instance after reset() (`_noise_groups` was unset / stale)."""
    from genlm_backend_amd.sis import DeviceSIS

    llm.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    sis = DeviceSIS(llm, 16, prompt, max_tokens=10, eos_id=0, seed=1234, rng="torch", use_particle_kv=True, share_kv=share)
    for _ in range(2):
        steps = sis.run()
        assert steps == int(gold["sis_steps"][0])
        _check_sis(*sis.results(), gold)
        sis.reset()  # (starts the seeded noise stream over as well)


def test_device_sis_with_one_mask_per_particle(llm, gold):
    """`particle_masks`: every particle brings its own bit mask (what a grammar gives), handed to the fused
    step raw on every call.  With every particle's mask equal to the README's `valid` mask the run IS the reference's
    golden run; with particles that may only emit one token each, that is what they emit."""
    from genlm_backend_amd.sis import DeviceSIS

    masks = torch.from_numpy(gold["sis_masks"])
    llm.register_masks(masks)
    prompt = [int(t) for t in gold["sis_prompt"]]
    bits, _ = llm.engine.mask_to_bits(masks)
    pm = torch.cat([bits[:1].expand(16, -1), bits[1:2]]).contiguous()
    sis = DeviceSIS(llm, 16, prompt, max_tokens=10, eos_id=0, seed=1234, rng="torch", particle_masks=pm)
    assert sis.run() == int(gold["sis_steps"][0])
    _check_sis(*sis.results(), gold)
    V = masks.shape[1]
    only = torch.full((17, V), float("-inf"))
    for i in range(16):
        only[i, 3 + i] = 0.0
    only[16, 0] = 0.0
    pm2, _ = llm.engine.mask_to_bits(only)
    sis = DeviceSIS(llm, 16, prompt, max_tokens=3, eos_id=0, seed=1, particle_masks=pm2)
    sis.run()
    ctx, _ = sis.results()
    assert [list(map(int, c)) for c in ctx] == [[3 + i] * 3 for i in range(16)]
    # masks that change between steps (a grammar moving some particles on): only those rows are prepared again
    sis = DeviceSIS(llm, 16, prompt, max_tokens=3, eos_id=0, seed=1, particle_masks=pm2.clone())
    sis.step()
    moved = torch.tensor([2, 5], dtype=torch.int32)
    other = torch.full((2, V), float("-inf"))
    other[0, 40], other[1, 41] = 0.0, 0.0
    sis.update_particle_masks(moved, llm.engine.mask_to_bits(other)[0])
    sis.run()
    ctx, _ = sis.results()
    want = [[3 + i] * 3 for i in range(16)]
    want[2], want[5] = [5, 40, 40], [8, 41, 41]
    assert [list(map(int, c)) for c in ctx] == want
    # ... and masks the caller edits IN PLACE, or replaces, without telling (the pattern that was supported before the
    # prepared form existed): the next step sees the tensor's version / identity change and prepares everything again
    sis = DeviceSIS(llm, 16, prompt, max_tokens=3, eos_id=0, seed=1, particle_masks=pm2.clone())
    sis.step()
    sis.particle_masks[7] = llm.engine.mask_to_bits(other)[0][0]  # particle 7 may only emit 40 from now on
    sis.step()
    fresh = pm2.clone()
    fresh[9] = llm.engine.mask_to_bits(other)[0][1]
    fresh[7] = llm.engine.mask_to_bits(other)[0][0]
    sis.particle_masks = fresh                                     # a new tensor: particle 9 may only emit 41
    sis.step()
    ctx, _ = sis.results()
    want = [[3 + i] * 3 for i in range(16)]
    want[7], want[9] = [10, 40, 40], [12, 12, 41]
    assert [list(map(int, c)) for c in ctx] == want
    sis.reset()  # a new run prepares its masks again
    assert sis._pm_prepared is None


def test_device_sis_hands_moving_masks_over_raw_and_standing_ones_prepared(llm, gold):
    """Round 6: the policy of `DeviceSIS._finish_step` for per-particle bit masks, on the CPU engine (host logic: the hand-over
    kind of every step is counted in `pm_raw_steps`) - a tensor state seen for the first time goes RAW, seen again it is
    prepared; few changed rows (`update_particle_masks`) are prepared again, many make the step raw while the prepared form
    lags, a step before which nothing moved brings it up to date; `pm_raw_above = 1.0` never goes raw; the parity draw (two
    launches) never does either.  Whatever the mix, the same run."""
    from genlm_backend_amd.sis import DeviceSIS

    masks = torch.from_numpy(gold["sis_masks"])
    llm.register_masks(masks)
    prompt = [int(t) for t in gold["sis_prompt"]]
    V, N = masks.shape[1], 40
    g = torch.Generator()
    g.manual_seed(11)

    def rows(n):
        f = torch.where(torch.rand((n, V), generator=g) < 0.5, float("-inf"), 0.0)
        f[:, 1:4] = 0.0
        return llm.engine.mask_to_bits(f)[0]

    pm0 = torch.cat([rows(N), llm.engine.mask_to_bits(masks[1:2])[0]]).contiguous()
    script = [None, None, 3, 30, 2, None, N, 1]  # rows moved before each step (float32 rows: raw above a quarter = 10)
    edits = [None if e is None else (torch.randperm(N, generator=g)[:e].to(torch.int32), rows(e)) for e in script]
    runs = []
    for above, rng in ((None, "philox"), (1.0, "philox"), (None, "torch")):
        sis = DeviceSIS(llm, N, prompt, max_tokens=len(script), eos_id=0, seed=5, rng=rng, particle_masks=pm0.clone())
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
    assert runs[0][2] == [True, False, False, True, True, False, True, True]
    assert runs[1][2] == [False] * len(script) and runs[2][2] == [False] * len(script)
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("use_kv", [False, True])
def test_device_sis_matches_reference(llm, gold, use_kv):
    from genlm_backend_amd.sis import DeviceSIS

    llm.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    sis = DeviceSIS(llm, 16, prompt, max_tokens=10, eos_id=0, seed=1234, rng="torch", use_prefix_kv=use_kv)
    steps = sis.run()
    assert steps == int(gold["sis_steps"][0])
    ctx, lw = sis.results()
    _check_sis(ctx, lw, gold)
    probs, stats = sis.normalized_weights()
    assert np.abs(probs.numpy() - gold["sis_probs"]).max() < 1e-5  # README.md:108-110


def test_readme_sis_batched_submit_matches_reference(llm, gold):
    """The README loop with each step's requests handed over as ONE `batch_next_token_step` call (no coroutine per
    particle): same tokens and weights as the reference's run, with and without the prompt's KV cached."""
    llm.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    for with_kv in (False, True):
        llm.clear_cache()
        llm.set_rng("torch", 1234)
        if with_kv:
            llm.cache_kv(prompt)
        ctxs, lw, active = [[] for _ in range(16)], np.zeros(16, np.float64), [True] * 16
        steps = 0
        while any(active):
            idx = [i for i in range(16) if active[i]]
            logZ, tok = llm.batch_next_token_step_sync([prompt + ctxs[i] for i in idx],
                                                       [1 if len(ctxs[i]) >= 10 else 0 for i in idx])
            for i, z, t in zip(idx, logZ, tok):
                lw[i] += z
                if t == 0 or t < 0:
                    active[i] = False
                else:
                    ctxs[i].append(int(t))
            steps += 1
        _check_sis(ctxs, lw, gold)
        assert steps == int(gold["sis_steps"][0])
    with pytest.raises(ValueError):
        llm.batch_next_token_step_sync([[1], []])


def test_prefix_kv_is_evicted_least_recently_used_first(llm):
    pres = [[5, 6, 7], [8, 9, 10, 11], [12, 13]]
    llm.cache_kv(pres[0])
    one = llm._kv_lru.used
    llm._kv_lru.budget = int(one * 2.5)  # room for two three-token prefixes
    llm.cache_kv(pres[1])
    llm.walk_cache(pres[0] + [1])        # touch prefix 0: prefix 1 is now the least recently used
    llm.cache_kv(pres[2])
    assert llm._kv_lru.evictions >= 1
    assert llm.walk_cache(pres[1] + [1])[2] is None          # evicted: falls back to re-encoding ...
    assert llm.walk_cache(pres[0] + [1])[2] is not None      # ... the recently used ones stay
    assert llm.walk_cache(pres[2] + [1])[2] is not None
    got = asyncio.run(llm.batch_next_token_logprobs([pres[1] + [3], pres[0] + [3]]))
    for p, row in zip((pres[1] + [3], pres[0] + [3]), got):
        assert np.abs(row.numpy() - llm.next_token_logprobs_uncached(p).numpy()).max() < TOL


def test_logprob_rows_stay_under_the_byte_budget(llm):
    """50 steps of a growing population of contexts: the trie's log-prob rows never exceed the budget by more than the
    newest slab, evicted rows are recomputed on demand and equal the first computation."""
    V = llm.model.config.vocab_size
    rng = np.random.default_rng(0)
    ctxs = [[int(t) for t in rng.integers(1, V, 3)] for _ in range(48)]
    first = asyncio.run(llm.batch_next_token_logprobs(ctxs)).clone()
    slab = llm._rows.used
    llm._rows.budget = int(slab * 2.5)
    peak = 0
    for step in range(50):
        ctxs2 = [c + [int(rng.integers(1, V))] * (1 + step % 3) for c in ctxs]
        asyncio.run(llm.batch_next_token_logprobs(ctxs2))
        peak = max(peak, llm._rows.used)
    assert llm._rows.evictions > 10
    assert peak <= llm._rows.budget + 4 * slab  # (one batch's slab holds every new position: up to 3 per context)
    again = asyncio.run(llm.batch_next_token_logprobs(ctxs))  # rows of step 0 were evicted long ago: recomputed
    assert np.abs(again.numpy() - first.numpy()).max() < 1e-5
    one = asyncio.run(llm.next_token_logprobs(ctxs[0]))
    assert np.abs(one.numpy() - first[0].numpy()).max() < 1e-5


def test_prefix_store_bookkeeping(llm):
    """Caching a prompt again, or one that re-creates its leading nodes, must not leave orphaned entries under the
    budget; evicted entries take their token tuple and the device prefix table with them; the batched path counts
    the prefixes it uses as recently used."""
    llm.cache_kv([5, 6, 7])
    one = llm._kv_lru.used
    llm.cache_kv([5, 6, 7])
    assert len(llm._kv_lru) == 1 and llm._kv_lru.used == one and len(llm._kv_tokens) == 1
    llm.cache_kv([5, 6, 7, 8])  # re-creates the nodes of [5, 6, 7]: the shorter prefix's node is unreachable now
    assert len(llm._kv_lru) == 1 and len(llm._kv_tokens) == 1
    assert llm.walk_cache([5, 6, 7, 8, 9])[3] == 4
    llm.clear_cache()
    pres = [[5, 6, 7], [8, 9, 10], [12, 13, 14]]
    llm.cache_kv(pres[0])
    llm._kv_lru.budget = int(llm._kv_lru.used * 2.5)
    llm.cache_kv(pres[1])
    logZ, tok = llm.batch_next_token_step_sync([pres[0] + [1], pres[0] + [2]])  # uses prefix 0 only
    assert llm._ptab is not None and llm._ptab["n"] == 2
    llm.cache_kv(pres[2])  # evicts the least recently used: prefix 1
    assert llm._kv_lru.evictions == 1 and len(llm._kv_tokens) == 2 and llm._ptab is None
    assert llm.walk_cache(pres[1] + [1])[2] is None and llm.walk_cache(pres[0] + [1])[2] is not None


@pytest.mark.parametrize("family", ["gemma2", "cohere", "granite"])
def test_logits_include_the_models_post_head_transform(family):
    """hf.py:275 reads `model(...).logits`: soft-capping (Gemma-2), logit_scale (Cohere) and logits_scaling (Granite)
    are applied after the output embedding and must be part of the log-probs."""
    import transformers as T

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.llm import AsyncAmdLM

    torch.manual_seed(0)
    common = dict(vocab_size=200, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                  num_key_value_heads=2, max_position_embeddings=64, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    if family == "gemma2":
        model = T.Gemma2ForCausalLM(T.Gemma2Config(head_dim=8, final_logit_softcapping=2.0, query_pre_attn_scalar=8, **common))
    elif family == "cohere":
        model = T.CohereForCausalLM(T.CohereConfig(logit_scale=0.25, **common))
    else:
        model = T.GraniteForCausalLM(T.GraniteConfig(logits_scaling=4.0, **common))
    model = model.eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(3.0)  # make the logits large enough for the transform to matter
    m = AsyncAmdLM(model, None, engine=CpuOracleEngine())
    ids = [5, 17, 42, 9]
    with torch.no_grad():
        want = torch.log_softmax(model(torch.tensor([ids])).logits[0, -1].float(), -1).numpy()
        plain = torch.log_softmax(model.get_output_embeddings()(model.base_model(torch.tensor([ids])).last_hidden_state)[0, -1], -1)
    assert np.abs(plain.numpy() - want).max() > 1e-3  # the transform is not a no-op on this model
    assert np.abs(m.next_token_logprobs_uncached(ids).numpy() - want).max() < TOL
    assert np.abs(asyncio.run(m.next_token_logprobs(ids)).numpy() - want).max() < TOL


def test_gather_lets_the_event_loop_run_and_cleans_up_after_a_failure(llm):
    """AsyncAmdLM.gather (no reference counterpart) next to foreign awaitables: a coroutine that polls with bare yields for
    something only the event loop can deliver, beside one blocked on a foreign future, must not starve the loop; and when a
    coroutine raises, what the closed ones had queued does not ride into the next batch."""
    m = llm
    ids = [3, 1, 4, 1, 5]

    async def run():
        loop = asyncio.get_running_loop()
        flag = {"set": False}
        loop.call_later(0.05, lambda: flag.__setitem__("set", True))
        fut = loop.create_future()
        loop.call_later(0.08, lambda: fut.set_result("late"))

        async def poller():
            n = 0
            while not flag["set"]:
                await asyncio.sleep(0)
                n += 1
            return n

        async def foreign():
            return await fut

        async def ours():
            return await m.next_token_step(ids, mask_id=0)

        res = await asyncio.wait_for(m.gather(poller(), foreign(), ours()), timeout=5.0)
        assert res[0] > 0 and res[1] == "late" and len(res[2]) == 2

        async def boom():
            await asyncio.sleep(0)
            raise RuntimeError("boom")

        with pytest.raises(RuntimeError):
            await m.gather(ours(), boom(), ours())
        assert m._sq is None and m.queries == [] and m.timer is None
        logZ, tok = await m.next_token_step(ids, mask_id=0)  # the backend still works, alone in its batch
        assert m.stats["queries"] >= 2
        return logZ, tok

    m.batch_size = 64
    asyncio.run(run())


def test_the_callers_model_is_never_modified(gold):
    """hf.py:114-140 leaves the model it is handed alone; so does this backend: fused activations / norms / rotary embedding
    and the attention entry live in a private shadow of the module tree that shares the weights (fuse.shadow_model).  Two
    backends over one model, and the model's own forward from a foreign thread while a backend is evaluating, see the model
    as it was built."""
    import threading

    from transformers import GPT2Config, GPT2LMHeadModel, LlamaConfig, LlamaForCausalLM

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    torch.manual_seed(3)
    llama = LlamaForCausalLM(LlamaConfig(vocab_size=cfg["vocab_size"], hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                         num_attention_heads=4, num_key_value_heads=2, head_dim=8, max_position_embeddings=64,
                                         bos_token_id=1, eos_token_id=2)).eval()
    for model in (GPT2LMHeadModel(GPT2Config(**cfg)).eval(), llama):
        kinds = lambda: [type(c).__name__ for c in model.modules()]
        before, impl = kinds(), model.config._attn_implementation
        fwd_attrs = [("forward" in c.__dict__) for c in model.modules()]
        ids = torch.tensor([[3, 1, 4, 1, 5, 9, 2, 6]])
        with torch.no_grad():
            want = model(ids).logits
        plain = AsyncAmdLM(model, None, engine=CpuOracleEngine(), fuse_activations=False, glb_attention=False)
        assert plain._net is model and plain.fused == []
        a = AsyncAmdLM(model, None, engine=CpuOracleEngine())
        b = AsyncAmdLM(model, None, engine=CpuOracleEngine())
        assert a._net is not model and a._net is not b._net and a.fused
        assert kinds() == before and model.config._attn_implementation == impl
        assert [("forward" in c.__dict__) for c in model.modules()] == fwd_attrs
        assert a._net.config is not model.config
        # the shadow shares the weights: same log-probs as the caller's model (different rounding only), from both backends
        ref = torch.log_softmax(want[0, -1], -1).numpy()
        for m in (plain, a, b):
            assert np.abs(m.next_token_logprobs_uncached(ids[0].tolist()).numpy() - ref).max() < TOL
        # a foreign thread runs the caller's model while the backends evaluate: bit-identical to before
        got, stop = [], threading.Event()

        def foreign():
            while not stop.is_set():
                with torch.no_grad():
                    got.append(model(ids).logits)

        th = threading.Thread(target=foreign)
        th.start()
        try:
            for _ in range(5):
                for m in (a, b):
                    m.clear_cache()
                    assert np.abs(m.next_token_logprobs_sync(ids[0].tolist()).numpy() - ref).max() < TOL
        finally:
            stop.set()
            th.join()
        assert got and all(torch.equal(g, want) for g in got)
        # weights replaced or changed in place by the caller reach the shadow
        with torch.no_grad():
            next(model.parameters()).mul_(1.25)
            want2 = model(ids).logits
        ref2 = torch.log_softmax(want2[0, -1], -1).numpy()
        a.clear_cache()
        assert np.abs(a.next_token_logprobs_sync(ids[0].tolist()).numpy() - ref2).max() < TOL


def test_lazy_trie_paths_match_eager_ones():
    """TokenTrie.extend_cache_lazy makes one node and leaves the rest of the path as a tail that grows when a walk gets
    there: every prefix must then find the same row an eagerly built trie holds, whatever the order of extensions and
    walks, and rows a budget took away are gone in both."""
    import random

    from genlm_backend_amd.cache import RowLRU, TokenTrie

    rnd = random.Random(7)
    V = 8
    eager, lazy = TokenTrie(), TokenTrie()
    st_e, st_l = RowLRU(1 << 40), RowLRU(1 << 40)

    def walk(root, toks):
        node, k = root, 0
        while k < len(toks) and node.has_token(toks[k]):
            node, k = node.get_token(toks[k]), k + 1
        return node, k

    seqs = []
    for step in range(300):
        if seqs and rnd.random() < 0.5:  # extend an earlier context or branch off inside it
            base = list(rnd.choice(seqs))
            cut = rnd.randint(0, len(base))
            toks = base[:cut] + [rnd.randrange(5) for _ in range(rnd.randint(1, 6))]
        else:
            toks = [rnd.randrange(5) for _ in range(rnd.randint(1, 9))]
        seqs.append(toks)
        ne, ke = walk(eager, toks)
        nl, kl = walk(lazy, toks)
        assert ke == kl
        if ke == len(toks):
            continue
        rows = torch.full((len(toks) - ke, V), float(step))  # row of position j: rows[j - ke]
        last = ne.extend_cache_rows(ke, toks, rows, ke, store=st_e)
        t, i = nl.extend_cache_lazy(kl, list(toks), rows, kl, store=st_l)
        assert torch.equal(last.logprobs, t[i])
        if rnd.random() < 0.3:  # look at every prefix of a random earlier context
            probe = rnd.choice(seqs)
            for k in range(1, len(probe) + 1):
                a, ka = walk(eager, probe[:k])
                b, kb = walk(lazy, probe[:k])
                assert ka == kb == k and torch.equal(a.logprobs, b.logprobs)
    for toks in seqs:
        for k in range(1, len(toks) + 1):
            a, _ = walk(eager, toks[:k])
            b, _ = walk(lazy, toks[:k])
            assert torch.equal(a.logprobs, b.logprobs)
    # a budget of one slab: older rows are gone from both, the newest stay
    small_e, small_l = RowLRU(1), RowLRU(1)
    e2, l2 = TokenTrie(), TokenTrie()
    for step, toks in enumerate(([1, 2, 3], [1, 2, 4, 4], [3, 3])):
        rows = torch.full((len(toks), V), float(step))
        ne, ke = walk(e2, toks)
        nl, kl = walk(l2, toks)
        ne.extend_cache_rows(ke, toks, rows[ke:], ke, store=small_e)
        nl.extend_cache_lazy(kl, toks, rows, 0, store=small_l)
    for toks, alive in (([1, 2, 3], False), ([1, 2, 4, 4], False), ([3, 3], True), ([3], True)):
        a, ka = walk(e2, toks)
        b, kb = walk(l2, toks)
        assert ka == kb == len(toks) and a.has_row() == b.has_row() == alive


@pytest.mark.parametrize("auto_rows", [0, 64])
def test_readme_sis_on_a_device_resident_population_matches_reference(gold, auto_rows):
    """`AsyncAmdLM.batch_next_token_step_device`: the README loop (README.md:72-98) with the population as tensors - a padded
    [N, cap] int32 token matrix + lengths in, (logZ, token) tensors out, the bookkeeping one `particles_advance` call - gives
    the reference's tokens and weights, with and without KV rows that follow the contexts."""
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    eng = CpuOracleEngine()
    llm = AsyncAmdLM(model, None, batch_size=64, timeout=0.02, engine=eng, auto_kv_rows=auto_rows, auto_kv_cap=32)
    llm.tokenizer = Tok()
    llm.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    N, P, max_tokens = 16, len(prompt), 10
    cap = P + max_tokens + 1
    llm.set_rng("torch", 1234)
    ctx = torch.zeros((N, cap), dtype=torch.int32)
    ctx[:, :P] = torch.tensor(prompt, dtype=torch.int32)
    ln = torch.full((N,), P, dtype=torch.int32)
    act = torch.ones(N, dtype=torch.int32)
    lw = torch.zeros(N, dtype=torch.float32)
    steps = 0
    while int(act.sum()) > 0:
        # (the reference's loop submits only the active particles; on tensors a finished particle rides along as a one-token
        # stub whose result nobody reads - the draws of the others must not move, so it is left out of the call here)
        idx = torch.nonzero(act > 0).flatten()
        mask_ids = ((ln[idx] - P) >= max_tokens).to(torch.int32)
        logZ, tok = llm.batch_next_token_step_device(ctx[idx].contiguous(), ln[idx].contiguous(), mask_ids)
        assert isinstance(logZ, torch.Tensor) and tok.dtype == torch.int32
        c, l, a, w = ctx[idx].contiguous(), ln[idx].contiguous(), act[idx].contiguous(), lw[idx].contiguous()
        eng.particles_advance(c, l, a, w, logZ, tok, 0, cap)
        ctx[idx], ln[idx], act[idx], lw[idx] = c, l, a, w
        steps += 1
    got = [[int(t) for t in ctx[i, P:ln[i]]] for i in range(N)]
    _check_sis(got, lw.numpy().astype(np.float64), gold)
    assert steps == int(gold["sis_steps"][0])
    # the list entry point and the tensor entry point are the same evaluation
    llm.clear_cache()
    llm.set_rng("philox", 7)
    qs = [prompt + g[:3] for g in got[:6]] + [prompt[:4]]
    a1 = llm.batch_next_token_step_sync(qs, [0, 1, 0, 1, 0, 0, 1])
    llm.clear_cache()
    llm.set_rng("philox", 7)
    width = max(len(q) for q in qs)
    mat = torch.zeros((len(qs), width), dtype=torch.int32)
    for i, q in enumerate(qs):
        mat[i, :len(q)] = torch.tensor(q, dtype=torch.int32)
    a2 = llm.batch_next_token_step_device(mat, torch.tensor([len(q) for q in qs], dtype=torch.int32),
                                          torch.tensor([0, 1, 0, 1, 0, 0, 1], dtype=torch.int32))
    assert np.array_equal(a1[0].view(np.uint32), a2[0].numpy().view(np.uint32)) and np.array_equal(a1[1], a2[1].numpy())
    with pytest.raises(ValueError):
        llm.batch_next_token_step_device(mat.long(), torch.ones(len(qs), dtype=torch.int32))


@pytest.mark.parametrize("rows,cap", [(64, 32), (6, 32), (64, 12)])
def test_readme_sis_with_auto_kv_matches_reference(gold, rows, cap):
    """`batch_next_token_step` with KV rows that follow the contexts (autokv.AutoKV): after the first call every context
    finds the row of its first L - 1 tokens and feeds one token; tokens and weights of the README loop stay the
    reference's - with rows to spare, with six rows for sixteen particles (contexts without a row are encoded) and
    with rows too short for the later contexts (those are encoded and not kept)."""
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    llm = AsyncAmdLM(model, None, batch_size=64, timeout=0.02, engine=CpuOracleEngine(), auto_kv_rows=rows, auto_kv_cap=cap)
    llm.tokenizer = Tok()
    llm.register_masks(torch.from_numpy(gold["sis_masks"]))
    prompt = [int(t) for t in gold["sis_prompt"]]
    for rep in range(2):  # the second run starts from a cleared cache: same results
        llm.clear_cache()
        llm.set_rng("torch", 1234)
        ctxs, lw, active = [[] for _ in range(16)], np.zeros(16, np.float64), [True] * 16
        steps = 0
        while any(active):
            idx = [i for i in range(16) if active[i]]
            logZ, tok = llm.batch_next_token_step_sync([prompt + ctxs[i] for i in idx],
                                                       [1 if len(ctxs[i]) >= 10 else 0 for i in idx])
            for i, z, t in zip(idx, logZ, tok):
                lw[i] += z
                if t == 0 or t < 0:
                    active[i] = False
                else:
                    ctxs[i].append(int(t))
            steps += 1
        _check_sis(ctxs, lw, gold)
        assert steps == int(gold["sis_steps"][0])
        st = llm._auto_kv.stats
        assert st["calls"] == steps and st["one_token_rows"] > 0
        if rows == 64 and cap == 32:
            assert st["encoded_rows"] == 1 and st["unkept_rows"] == 0 and st["copied_rows"] > 0  # only the prompt is encoded
        if rows == 6:
            assert st["unkept_rows"] > 0
        if cap == 12:
            assert st["unkept_rows"] > 0  # contexts of 13 tokens and more do not fit a row
    # the same contexts again: every one finds the row that holds exactly it
    llm.set_rng("torch", 99)
    qs = [prompt + c[:3] for c in ctxs[:5]]
    a = llm.batch_next_token_step_sync(qs, [0] * 5)
    before = dict(llm._auto_kv.stats)
    llm.set_rng("torch", 99)
    b = llm.batch_next_token_step_sync(qs, [0] * 5)
    assert np.array_equal(a[1], b[1]) and np.abs(a[0] - b[0]).max() < 1e-5
    if rows == 64 and cap == 32:
        assert llm._auto_kv.stats["encoded_rows"] == before["encoded_rows"]


@pytest.mark.parametrize("collide", [False, True])
def test_auto_kv_on_a_changing_set_of_contexts(gold, collide):
    """Contexts that grow, shrink, repeat, fork and appear from nowhere between calls, eight KV rows for up to twelve
    of them: every call's logZ and tokens equal those of a model without the rows.  `collide`: every context hashes to
    the same value - the lookup then finds the wrong candidate most of the time, the token comparison rejects it, and
    the results still hold (a hash never decides alone)."""
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    V = cfg["vocab_size"]

    def make(**kw):
        model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
        model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
        m = AsyncAmdLM(model, None, batch_size=64, timeout=0.02, engine=CpuOracleEngine(), **kw)
        m.tokenizer = Tok()
        m.register_masks(torch.from_numpy(gold["sis_masks"]))
        return m

    plain, auto = make(), make(auto_kv_rows=8, auto_kv_cap=10)
    if collide:
        eng = auto.engine
        real = eng.hash_contexts
        eng.hash_contexts = lambda tok, st, ln: torch.full_like(real(tok, st, ln), 12345)
    rnd = np.random.default_rng(5)
    ctxs = [[int(t) for t in rnd.integers(1, V, int(rnd.integers(1, 5)))] for _ in range(6)]
    for call in range(14):
        nxt = []
        for c in ctxs:
            r = rnd.random()
            if r < 0.55:
                nxt.append(c + [int(rnd.integers(1, V))])             # grows by one token
            elif r < 0.65:
                nxt.append(list(c))                                    # asked again as it is
            elif r < 0.75 and len(c) > 1:
                nxt.append(c[:-1])                                     # shrinks
            elif r < 0.85:
                nxt.append(c + [int(t) for t in rnd.integers(1, V, 3)])  # jumps ahead: no row holds its first L - 1 tokens
            else:
                nxt.append([int(t) for t in rnd.integers(1, V, int(rnd.integers(1, 12)))])  # from nowhere (some too long for a row)
        if rnd.random() < 0.7:
            nxt.append(list(ctxs[int(rnd.integers(0, len(ctxs)))]) + [int(rnd.integers(1, V))])  # a fork: a second child of a row
        if len(nxt) > 12:
            nxt = nxt[:12]
        ctxs = nxt
        mids = [int(rnd.integers(0, 2)) for _ in ctxs]
        for m in (plain, auto):
            m.set_rng("philox", 77 + call)
        z0, t0 = plain.batch_next_token_step_sync(ctxs, mids)
        z1, t1 = auto.batch_next_token_step_sync(ctxs, mids)
        fin = np.isfinite(z0)
        assert np.array_equal(fin, np.isfinite(z1)) and np.abs(z0[fin] - z1[fin]).max() < 1e-4, call
        assert np.array_equal(t0, t1), call
    st = auto._auto_kv.stats
    assert st["encoded_rows"] > 0 and st["unkept_rows"] > 0
    if not collide:
        assert st["one_token_rows"] > 20 and st["copied_rows"] > 0


def test_logprob_requests_for_the_last_position_use_auto_kv(gold):
    """`next_token_logprobs` / `batch_next_token_logprobs` of contexts whose shorter prefixes are in the trie (a
    population that grew by one token) go through the KV rows too: one token per context is fed, the rows equal the
    uncached evaluation."""
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    llm = AsyncAmdLM(model, None, batch_size=64, timeout=0.02, engine=CpuOracleEngine(), auto_kv_rows=24, auto_kv_cap=16)
    llm.tokenizer = Tok()
    V = cfg["vocab_size"]
    rnd = np.random.default_rng(3)
    ctxs = [[int(t) for t in rnd.integers(1, V, 4)] for _ in range(10)]
    ctxs[7] = list(ctxs[2])  # duplicates share a row
    for step in range(6):
        got = asyncio.run(llm.batch_next_token_logprobs(ctxs))
        for c, row in zip(ctxs, got):
            assert np.abs(row.numpy() - llm.next_token_logprobs_uncached(c).numpy()).max() < TOL
        one = asyncio.run(llm.next_token_logprobs(ctxs[0]))  # served from the trie
        assert torch.equal(one, got[0])
        ctxs = [c + [int(rnd.integers(1, V))] for c in ctxs]
    st = llm._auto_kv.stats
    # step 0 is a full evaluation (every position's row is wanted); step 1 finds no rows yet and encodes its ten
    # contexts; steps 2.. feed one token each
    assert st["calls"] == 5 and st["encoded_rows"] == 10 and st["one_token_rows"] == 40 and st["unkept_rows"] == 0


@pytest.mark.parametrize("auto_kv", [False, True])
def test_user_side_particle_math_on_the_returned_rows_is_a_drop_in(llm, gold, auto_kv):
    """The reference's own usage pattern with nothing but `llm` swapped: every particle awaits `next_token_logprobs`,
    adds its mask, takes logsumexp and draws with torch.multinomial from torch's global generator (the README's
    Particle.extend, README.md:82-91, written out here as user code).  Rows within 1e-4 and the reference's resolution
    order of the futures (hf.py:285-288) make the tokens and weights of the golden run come out."""
    masks = torch.from_numpy(gold["sis_masks"])
    prompt = [int(t) for t in gold["sis_prompt"]]
    if auto_kv:  # the same user code on a backend whose requests find their KV rows (one token per context is fed)
        from genlm_backend_amd.autokv import AutoKV

        llm._auto_kv = AutoKV(llm, 24, 24)

    class UserParticle:
        def __init__(self):
            self.context, self.log_weight, self.active = [], 0.0, True

        async def extend(self):
            logps = await llm.next_token_logprobs(prompt + self.context)
            masked = logps + masks[1 if len(self.context) >= 10 else 0].to(logps.device)
            logZ = masked.logsumexp(dim=-1)
            self.log_weight += logZ
            tok = torch.multinomial((masked - logZ).exp(), 1).item()
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
    _check_sis([p.context for p in ps], [float(p.log_weight) for p in ps], gold)
    if auto_kv:
        assert llm._auto_kv.stats["one_token_rows"] > 50


def test_load_model_by_name_end_to_end_offline(tmp_path):
    """`load_model_by_name` (llm/__init__.py:10-43) from a local checkpoint directory - a tiny GPT-2 saved with the
    byte-level BPE tokenizer of tests/golden -: tokenizer, byte_vocab / str_vocab, the README's mask builder and a
    short autobatched SIS run through the returned object; the reference's other engines and unknown names are refused."""
    from tokenizers import Tokenizer
    from transformers import GPT2Config, GPT2LMHeadModel, PreTrainedTokenizerFast

    from genlm_backend_amd.llm import AsyncAmdLM, load_model_by_name
    from genlm_backend_amd.sis import autobatched_sis, make_masking_function

    gd = os.path.join(os.path.dirname(__file__), "golden")
    tok = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_file(os.path.join(gd, "bpe_tokenizer.json")),
                                  eos_token="<|endoftext|>")
    torch.manual_seed(3)
    GPT2LMHeadModel(GPT2Config(vocab_size=len(tok), n_positions=64, n_embd=32, n_layer=2, n_head=2, bos_token_id=0,
                               eos_token_id=0)).save_pretrained(tmp_path)
    tok.save_pretrained(tmp_path)
    llm = load_model_by_name(str(tmp_path), backend="amd",
                             llm_opts={"hf_opts": {"device": "cpu"}, "engine": CpuOracleEngine(), "batch_size": 8})
    assert isinstance(llm, AsyncAmdLM) and llm.tokenizer.eos_token_id == 0
    assert len(llm.byte_vocab) == len(llm.str_vocab) == len(tok)
    assert llm.byte_vocab[llm.tokenizer.encode("ab")[0]] in (b"ab", b"a")
    sel = make_masking_function(llm, max_token_length=3, max_tokens=4)
    llm.set_rng("philox", 5)
    prompt = llm.tokenizer.encode("the cat")
    parts = asyncio.run(autobatched_sis(8, llm, sel, prompt, eos_id=0))
    assert all(not p.active and 1 <= len(p.context) + 1 <= 6 and np.isfinite(p.log_weight) for p in parts)
    for p in parts:  # the README mask: no generated token longer than three bytes
        assert all(len(llm.byte_vocab[t]) <= 3 for t in p.context)
    row = asyncio.run(llm.next_token_logprobs(prompt))
    assert abs(float(row.exp().sum()) - 1.0) < 1e-4
    for bad in ("vllm", "mlx", "nonsense"):
        with pytest.raises(ValueError):
            load_model_by_name(str(tmp_path), backend=bad)


def test_recorded_gemm_solutions_file_is_well_formed_and_optional():
    """genlm_backend_amd.gemm_tuning: the shipped tuned/<arch>.csv is a TunableOp results file (validator lines, then one
    line per GEMM shape: operation, shape key, solution name, time); without a GPU nothing is switched on."""
    import os

    from genlm_backend_amd import gemm_tuning

    assert gemm_tuning.recorded_file() is None and gemm_tuning.use_recorded() == 0  # (no GPU here)
    path = os.path.join(os.path.dirname(gemm_tuning.__file__), "tuned", "gfx950.csv")
    lines = open(path).read().splitlines()
    vals = [ln for ln in lines if ln.startswith("Validator,")]
    assert {v.split(",")[1] for v in vals} >= {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"}
    ents = [ln.split(",") for ln in lines if ln and not ln.startswith("Validator,")]
    assert len(ents) > 100 and all(len(e) == 4 and e[0].startswith("Gemm") and float(e[3]) > 0 for e in ents)
    assert len({(e[0], e[1]) for e in ents}) == len(ents)  # (a shape once)
    assert any(e[1].startswith("tn_50257_1024_768") for e in ents)  # (BASELINE config 2's lm_head)
    # the product's handle on the process-wide switch (AsyncAmdLM(gemms="recorded") / close()): counted, the last holder
    # switches TunableOp off again
    calls = []
    real_use, real_off = gemm_tuning.use_recorded, gemm_tuning.off
    gemm_tuning.use_recorded = lambda path=None, device=None: calls.append("on") or 7
    gemm_tuning.off = lambda: calls.append("off")
    try:
        assert gemm_tuning.acquire() == 7 and gemm_tuning.acquire() == 7 and calls == ["on"]
        gemm_tuning.release()
        assert calls == ["on"]
        gemm_tuning.release()
        gemm_tuning.release()  # (one too many: nothing happens)
        assert calls == ["on", "off"]
    finally:
        gemm_tuning.use_recorded, gemm_tuning.off = real_use, real_off
