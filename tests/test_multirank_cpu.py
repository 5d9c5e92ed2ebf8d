"""N > 1 path on CPU: two gloo ranks shard the particle population, all-gather the log-weights every
step and derive identical normalised weights; the union of the shards equals the one-process run."""
import ast
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "ref_hotpath_tiny.npz")


class Tok:
    pad_token_id = None
    eos_token_id = 0


def _llm():
    from transformers import GPT2Config, GPT2LMHeadModel

    sys.path.insert(0, ROOT)
    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.llm import AsyncAmdLM
    from tests.cpu_engine import CpuOracleEngine

    gold = np.load(G)
    cfg = ast.literal_eval(bytes(gold["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")})
    m = AsyncAmdLM(model, None, batch_size=64, engine=CpuOracleEngine())
    m.tokenizer = Tok()
    m.register_masks(torch.from_numpy(gold["sis_masks"]))
    return m, [int(t) for t in gold["sis_prompt"]]


def _worker(rank, world, port, out_dir, mode="philox"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genlm_backend_amd.sis import DeviceSIS

    m, prompt = _llm()
    if mode == "golden":  # the reference's own run (16 particles, torch.manual_seed(1234), README loop), cut over the ranks
        sis = DeviceSIS(m, 16 // world, prompt, max_tokens=10, eos_id=0, seed=1234, rng="torch", rank=rank, world=world, dist=dist)
    elif mode == "torch":  # the reference's draws with resampling after every step (no reference run to compare with: one process)
        sis = DeviceSIS(m, 8, prompt, max_tokens=5, eos_id=0, seed=99, rng="torch", rank=rank, world=world, dist=dist, resample_ess=1.0)
    else:
        sis = DeviceSIS(m, 8, prompt, max_tokens=5, eos_id=0, seed=99, rank=rank, world=world, dist=dist)
    sis.run()
    ctx, lw = sis.results()
    probs, stats = sis.normalized_weights()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ctx=np.array([list(c) + [-1] * (12 - len(c)) for c in ctx]), lw=lw,
             all_lw=sis.all_weights.numpy(), probs=probs.numpy(), stats=stats.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_sis(tmp_path):
    world, port = 2, 29541 + os.getpid() % 200
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    # every rank holds the same gathered weights and derives bit-identical normalised weights / ESS
    assert np.array_equal(r[0]["all_lw"], r[1]["all_lw"])
    assert np.array_equal(r[0]["probs"], r[1]["probs"]) and np.array_equal(r[0]["stats"], r[1]["stats"])
    assert np.array_equal(r[0]["all_lw"], np.concatenate([r[0]["lw"], r[1]["lw"]]))
    # one process with the whole population draws the same tokens (Philox keyed by the global particle index)
    from genlm_backend_amd.sis import DeviceSIS

    m, prompt = _llm()
    one = DeviceSIS(m, 16, prompt, max_tokens=5, eos_id=0, seed=99)
    one.run()
    ctx, lw = one.results()
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([r[0]["ctx"], r[1]["ctx"]])]
    assert got == [list(map(int, c)) for c in ctx]
    assert np.abs(lw - r[0]["all_lw"]).max() < 1e-5
    assert abs(float(r[0]["probs"].sum()) - 1.0) < 1e-5


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_parity_draws_are_the_references_own_run(tmp_path, world):
    """with several ranks every rank used to seed an MT19937 stream of its own, so the only run whose ids
    equal torch.multinomial's was the one-process run.  Now every rank enters ONE global stream at its particles' rows (the
    population's dedup groups numbered in first-appearance order over the global particle index on every rank, from the
    all-gathered context hashes): two and four gloo ranks reproduce the REFERENCE's golden run (README.md:72-110 under
    torch.manual_seed: tokens and log-weights of ref_hotpath_tiny.npz)."""
    port = 29941 + os.getpid() % 200 + world
    mp.start_processes(_worker, args=(world, port, str(tmp_path), "golden"), nprocs=world, join=True, start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    gold = np.load(G)
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([x["ctx"] for x in r])]
    assert got == [[int(t) for t in row if t >= 0] for row in gold["sis_contexts"]]
    assert np.abs(np.concatenate([x["lw"] for x in r]) - gold["sis_log_weights"]).max() < 1e-4


@pytest.mark.timeout(600)
def test_sharded_parity_draws_with_resampling_equal_one_process(tmp_path):
    """The same global stream when particles change ranks every step (systematic resampling, replicated): the union of two
    ranks equals one process with the whole population."""
    world, port = 2, 30141 + os.getpid() % 200
    mp.start_processes(_worker, args=(world, port, str(tmp_path), "torch"), nprocs=world, join=True, start_method="spawn")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    from genlm_backend_amd.sis import DeviceSIS

    m, prompt = _llm()
    one = DeviceSIS(m, 16, prompt, max_tokens=5, eos_id=0, seed=99, rng="torch", resample_ess=1.0)
    one.run()
    ctx, lw = one.results()
    got = [[int(t) for t in row if t >= 0] for row in np.concatenate([r[0]["ctx"], r[1]["ctx"]])]
    assert got == [list(map(int, c)) for c in ctx]
    assert np.abs(lw - np.concatenate([r[0]["lw"], r[1]["lw"]])).max() < 1e-5
