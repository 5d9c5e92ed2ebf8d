"""Throughput of the reference-shaped asyncio API (one coroutine per particle, README.md:82-115) next to DeviceSIS."""
import asyncio, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from transformers import GPT2Config
from genlm_backend_amd.llm import AsyncAmdLM
from genlm_backend_amd.sis import autobatched_sis, DeviceSIS
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = GPT2Config()
llm = AsyncAmdLM.from_config(cfg, None, device=dev, dtype=torch.float32, seed=1, batch_size=N, timeout=0.002)
V = cfg.vocab_size
g = torch.Generator(device=dev); g.manual_seed(1)
valid = torch.where(torch.rand(V, device=dev, generator=g) < 1 / 3, float("-inf"), 0.0); valid[cfg.eos_token_id] = 0.0
eos1 = torch.full((V,), float("-inf"), device=dev); eos1[cfg.eos_token_id] = 0.0
llm.register_masks(torch.stack([valid, eos1]))
llm.set_rng("philox", 3)
sel = lambda ctx: 1 if len(ctx) >= 10 else 0
prompt = list(range(100, 108))
for rep in range(2):
    llm.clear_cache() if hasattr(llm, "clear_cache") else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    parts = asyncio.run(autobatched_sis(N, llm, sel, prompt, eos_id=cfg.eos_token_id))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = max(len(p.context) for p in parts)
    print(f"asyncio API, {N} particles: {dt:.3f} s for {steps} steps -> {dt / steps * 1e3:.1f} ms/step, {N * steps / dt:.0f} particle-steps/s; stats {llm.stats}", flush=True)
    llm.stats = {"batches": 0, "queries": 0, "unique": 0, "rows": 0}
sis = DeviceSIS(llm, N, prompt, 10, cfg.eos_token_id, seed=3)
for rep in range(2):
    sis.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    sis.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"DeviceSIS, {N} particles: {dt:.3f} s for {sis.t} steps -> {dt / max(sis.t,1) * 1e3:.1f} ms/step", flush=True)
if len(sys.argv) > 2:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    parts = asyncio.run(autobatched_sis(N, llm, sel, prompt, eos_id=cfg.eos_token_id))
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
