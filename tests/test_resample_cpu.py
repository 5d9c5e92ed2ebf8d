"""Resampling and device-resident particle state on CPU (host logic over the oracle-backed engine double):
the systematic comb against a float64 restatement, slab KV == re-encode == prefix KV on GPT-2- and Llama-shaped
models with ragged prompts, and 1-rank == 2-rank (gloo) through several resampling steps, including a shard that
finishes early."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_systematic_comb_matches_float64(oracle):
    rs = np.random.default_rng(5)
    for n in (1, 2, 7, 64, 1000, 4096):
        lw = (rs.standard_normal(n) * 3).astype(np.float32)
        lw[rs.random(n) < 0.1] = -np.inf
        if not np.isfinite(lw).any():
            lw[0] = 0.0
        anc, lse = oracle.resample_systematic(lw, seed=11, offset=n)
        assert (np.diff(anc) >= 0).all() and anc.min() >= 0 and anc.max() < n
        w = np.exp(lw.astype(np.float64) - np.logaddexp.reduce(lw.astype(np.float64)))
        counts = np.bincount(anc, minlength=n)
        assert np.abs(counts - n * w).max() < 1.0 + 1e-6  # systematic resampling: every count within one of n*w
        assert counts[~np.isfinite(lw)].sum() == 0
        assert abs(lse - np.logaddexp.reduce(lw.astype(np.float64))) < 1e-5
        # same weights, another draw: still a valid comb; same draw: same ancestors
        assert np.array_equal(anc, oracle.resample_systematic(lw, seed=11, offset=n)[0])
    anc, _ = oracle.resample_systematic(np.full(8, -np.inf, np.float32), 1, 1)
    assert np.array_equal(anc, np.arange(8))  # no mass anywhere: identity


class Tok:
    pad_token_id = None
    eos_token_id = 0


def _tiny(kind, seed=0):
    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.llm import AsyncAmdLM
    from tests.cpu_engine import CpuOracleEngine
    from transformers import GPT2Config, GPT2LMHeadModel, LlamaConfig, LlamaForCausalLM

    torch.manual_seed(seed)
    V = 300
    if kind == "gpt2":
        m = GPT2LMHeadModel(GPT2Config(vocab_size=V, n_positions=64, n_embd=32, n_layer=2, n_head=2, bos_token_id=0,
                                       eos_token_id=0)).eval()
    else:  # RoPE + grouped-query attention
        m = LlamaForCausalLM(LlamaConfig(vocab_size=V, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                         num_attention_heads=4, num_key_value_heads=2, head_dim=8,
                                         max_position_embeddings=64, bos_token_id=0, eos_token_id=0)).eval()
    llm = AsyncAmdLM(m, None, batch_size=64, engine=CpuOracleEngine())
    llm.tokenizer = Tok()
    masks = torch.zeros(2, V)
    masks[0, ::3] = float("-inf")
    masks[0, 7:40] = float("-inf")  # uneven allowed mass across contexts -> uneven weights -> real resampling
    masks[1, :] = float("-inf")
    masks[1, 0] = 0
    llm.register_masks(masks)
    return llm


PROMPTS = [[5, 6, 7, 8], [5, 6, 7, 8], [9, 10, 11], [12, 13, 14, 15, 16], [9, 10, 11], [20, 21], [5, 6, 7, 8], [33]]


@pytest.mark.parametrize("kind", ["gpt2", "llama"])
def test_slab_kv_equals_reencode_and_prefix_kv(kind):
    from genlm_backend_amd.sis import DeviceSIS

    llm = _tiny(kind)
    res = {}
    for mode in ("plain", "pkv", "prefix"):
        s = DeviceSIS(llm, len(PROMPTS), PROMPTS, max_tokens=6, eos_id=0, seed=3, use_particle_kv=mode == "pkv",
                      use_prefix_kv=mode == "prefix")
        s.run()
        res[mode] = s.results()
    assert res["plain"][0] == res["pkv"][0] == res["prefix"][0]
    assert np.abs(res["plain"][1] - res["pkv"][1]).max() < 1e-4
    assert np.abs(res["plain"][1] - res["prefix"][1]).max() < 1e-4
    # with resampling after every step the KV rows follow their ancestors
    a = DeviceSIS(llm, len(PROMPTS), PROMPTS, max_tokens=6, eos_id=0, seed=3, use_particle_kv=True, resample_ess=1.0)
    a.run()
    b = DeviceSIS(llm, len(PROMPTS), PROMPTS, max_tokens=6, eos_id=0, seed=3, resample_ess=1.0)
    b.run()
    assert a.n_resamples >= 2 and a.results()[0] == b.results()[0]
    assert np.abs(a.results()[1] - b.results()[1]).max() < 1e-4


@pytest.mark.parametrize("kind", ["gpt2", "llama"])
def test_shared_kv_rows(kind):
    """Particles with equal contexts share one KV row and one forward row: with resampling after every step the
    forward runs on the distinct contexts only (fewer rows than particles), rows are copied only when a shared row's
    particles diverge, and a row budget that is spent (kv_rows = 3 for 16 particles) changes nothing but the number of
    contexts encoded from their tokens."""
    from genlm_backend_amd.sis import DeviceSIS

    llm = _tiny(kind)
    prompts = PROMPTS * 2
    ref = DeviceSIS(llm, len(prompts), prompts, max_tokens=6, eos_id=0, seed=3, resample_ess=1.0)
    ref.run()
    runs = {}
    for rows in (None, 3):
        s = DeviceSIS(llm, len(prompts), prompts, max_tokens=6, eos_id=0, seed=3, use_particle_kv=True, resample_ess=1.0,
                      kv_rows=rows)
        s.run()
        runs[rows] = s
        assert s.results()[0] == ref.results()[0]
        assert np.abs(s.results()[1] - ref.results()[1]).max() < 1e-4
    full, tight = runs[None].kv_stats, runs[3].kv_stats
    n = len(prompts)
    assert full["forward_rows"] < n * full["steps"]          # dedup: duplicates of an ancestor are forwarded once
    assert full["encoded_rows"] == len({tuple(p) for p in prompts})   # only step 0 encodes (the distinct prompts)
    assert full["copied_rows"] > 0 and full["unkept_rows"] == 0
    assert tight["unkept_rows"] > 0 and tight["encoded_rows"] > full["encoded_rows"]
    assert runs[3].pkv.n == 3


def _worker(rank, world, port, out_dir, kind, pkv, prefix=False, share=True, migrate=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genlm_backend_amd.sis import DeviceSIS

    llm = _tiny(kind)
    n = len(PROMPTS) // world
    mine = PROMPTS[rank * n:(rank + 1) * n]
    sis = DeviceSIS(llm, n, mine, max_tokens=6, eos_id=0, seed=3, rank=rank, world=world, dist=dist,
                    use_particle_kv=pkv, use_prefix_kv=prefix, resample_ess=1.0, share_kv=share, migrate_kv=migrate)
    encoded = []
    if pkv and not share:
        # (the second rank's particles start 5 nats behind: the first resampling step fills its slots from rank 0)
        sis.log_weights = sis.log_weights - 5.0 * rank
        # count the rows whose KV is rebuilt from the context after a resampling step
        orig = sis._encode_into_slabs
        sis._encode_into_slabs = lambda idx: (encoded.append(-1 if idx is None else int(idx.numel())), orig(idx))[1]  # (None: step 0, every row)
    steps = sis.run()
    ctx, lw = sis.results()
    width = 8
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ctx=np.array([list(c) + [-1] * (width - len(c)) for c in ctx]),
             lw=lw, all_lw=sis.all_weights.numpy(), steps=steps, n_res=sis.n_resamples, kv_moved=sis.kv_rows_moved,
             rows_moved=sis.rows_moved, reencoded=sum(e for e in encoded if e > 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,pkv,prefix", [("gpt2", True, False), ("llama", False, False), ("gpt2", False, True)])
def test_two_ranks_equal_one_through_resampling(tmp_path, kind, pkv, prefix):
    """Replicated deterministic resampling: two gloo ranks (4 particles each; ancestors cross the shard boundary, so
    contexts travel and KV rows are rebuilt) end with the same tokens and weights as one process with all 8.  With
    cached prompt prefixes the two ranks hold DIFFERENT prompts (PROMPTS[4:] shares only one with PROMPTS[:4]): a
    migrated particle must still find its prompt's KV, so every rank caches the prompts of all ranks."""
    world, port = 2, 29741 + os.getpid() % 200
    mp.start_processes(_worker, args=(world, port, str(tmp_path), kind, pkv, prefix), nprocs=world, join=True,
                       start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert int(r[0]["steps"]) == int(r[1]["steps"]) and int(r[0]["n_res"]) >= 2
    assert np.array_equal(r[0]["all_lw"], r[1]["all_lw"])
    from genlm_backend_amd.sis import DeviceSIS

    one = DeviceSIS(_tiny(kind), len(PROMPTS), PROMPTS, max_tokens=6, eos_id=0, seed=3, use_particle_kv=pkv,
                    use_prefix_kv=prefix, resample_ess=1.0)
    one.run()
    ctx, lw = one.results()
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([r[0]["ctx"], r[1]["ctx"]])]
    assert got == [list(map(int, c)) for c in ctx]
    assert np.abs(lw - r[0]["all_lw"]).max() < 1e-4


@pytest.mark.timeout(900)
@pytest.mark.parametrize("migrate", [True, False])
def test_private_kv_rows_travel_with_their_particles(tmp_path, migrate):
    """Per-particle KV slabs (share_kv=False) on two gloo ranks: a particle resampled from the other rank brings its KV rows
    along in the resampling step's second all-to-all (migrate_kv, the default) instead of having them rebuilt from its
    context - same tokens and weights as one process either way; with migration nothing is re-encoded after step 0."""
    world, port = 2, 29941 + os.getpid() % 200
    mp.start_processes(_worker, args=(world, port, str(tmp_path), "gpt2", True, False, False, migrate), nprocs=world, join=True,
                       start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert int(r[0]["n_res"]) >= 2 and np.array_equal(r[0]["all_lw"], r[1]["all_lw"])
    from genlm_backend_amd.sis import DeviceSIS

    one = DeviceSIS(_tiny("gpt2"), len(PROMPTS), PROMPTS, max_tokens=6, eos_id=0, seed=3, use_particle_kv=True,
                    resample_ess=1.0, share_kv=False)
    one.log_weights[len(PROMPTS) // 2:] -= 5.0
    one.run()
    ctx, lw = one.results()
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([r[0]["ctx"], r[1]["ctx"]])]
    assert got == [list(map(int, c)) for c in ctx]
    assert np.abs(lw - r[0]["all_lw"]).max() < 1e-4
    moved = int(r[0]["kv_moved"]) + int(r[1]["kv_moved"])
    reenc = int(r[0]["reencoded"]) + int(r[1]["reencoded"])
    if migrate:
        assert moved > 0 and reenc == 0
    else:
        assert moved == 0 and reenc > 0


def _early_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genlm_backend_amd.sis import DeviceSIS

    llm = _tiny("gpt2")
    # rank 0's particles may only emit EOS (mask 1 from the first step: max_tokens = 0 generated tokens allowed)
    sis = DeviceSIS(llm, 4, PROMPTS[:4], max_tokens=0 if rank == 0 else 5, eos_id=0, seed=3, rank=rank, world=world,
                    dist=dist)
    sis.cap = max(sis.cap, 16)
    steps = sis.run(max_steps=7)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), steps=steps, all_lw=sis.all_weights.numpy(),
             active=int(sis.active.sum()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_shard_that_finishes_early_keeps_the_collectives_matched(tmp_path):
    """One rank's particles all hit EOS at step 1 while the other rank keeps generating: both ranks run the same
    number of steps (the loop's termination test is taken over all ranks), nothing hangs, weights agree."""
    world, port = 2, 29941 + os.getpid() % 200
    mp.start_processes(_early_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert int(r[0]["steps"]) == int(r[1]["steps"]) > 2
    assert int(r[0]["active"]) == 0
    assert np.array_equal(r[0]["all_lw"], r[1]["all_lw"])


# ---- BASELINE config 4's real sizes: 4096 particles over 8 ranks (512 each), resampling after every step --------------
def _config4_prompts(n_total):
    rs = np.random.default_rng(44)
    pool = [[int(t) for t in rs.integers(1, 290, size=rs.integers(2, 7))] for _ in range(48)]
    return [pool[i] for i in rs.integers(0, len(pool), size=n_total)]


def _config4_worker(rank, world, port, out_dir, per_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genlm_backend_amd.sis import DeviceSIS

    prompts = _config4_prompts(world * per_rank)[rank * per_rank:(rank + 1) * per_rank]
    sis = DeviceSIS(_tiny("llama"), per_rank, prompts, max_tokens=4, eos_id=0, seed=9, rank=rank, world=world, dist=dist,
                    use_particle_kv=True, resample_ess=1.0)
    moved = []
    steps = 0
    while steps < 5:
        _, n_global = sis.step()
        steps += 1
        moved.append(sis.rows_moved)
        if n_global == 0:
            break
    ctx, lw = sis.results()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ctx=np.array([list(c) + [-1] * (6 - len(c)) for c in ctx]), lw=lw,
             all_lw=sis.all_weights.numpy(), steps=steps, n_res=sis.n_resamples, moved=np.array(moved))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_eight_ranks_of_512_equal_one_process(tmp_path):
    """Config 4's shape of the population (4096 particles, 512 per rank, 8 ranks; gloo on the CPU engine double, a tiny
    Llama-shaped model): rank-specific ragged prompts, shared KV rows, systematic resampling after every step.  The
    union equals one process with all 4096; and a resampling step moves only the particles whose ancestor lives on another
    rank (one all-to-all), not the population."""
    world, per_rank, port = 8, 512, 30141 + os.getpid() % 200
    mp.start_processes(_config4_worker, args=(world, port, str(tmp_path), per_rank), nprocs=world, join=True,
                       start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert len({int(x["steps"]) for x in r}) == 1 and int(r[0]["n_res"]) >= 3
    for x in r[1:]:
        assert np.array_equal(x["all_lw"], r[0]["all_lw"]) and np.array_equal(x["moved"], r[0]["moved"])
    moved = r[0]["moved"]
    assert 0 < moved.max() < world * per_rank // 2  # something crossed ranks, and far from everything
    from genlm_backend_amd.sis import DeviceSIS

    one = DeviceSIS(_tiny("llama"), world * per_rank, _config4_prompts(world * per_rank), max_tokens=4, eos_id=0, seed=9,
                    use_particle_kv=True, resample_ess=1.0)
    for _ in range(int(r[0]["steps"])):
        one.step()
    ctx, lw = one.results()
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([x["ctx"] for x in r])]
    assert got == [list(map(int, c)) for c in ctx]
    assert np.abs(lw - r[0]["all_lw"]).max() < 1e-4
