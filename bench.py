#!/usr/bin/env python3
"""bench.py — headline benchmark of the genlm-backend hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One process per GPU.  `python bench.py --gpus N` with N > 1 starts the N ranks itself: the parent process (which
never touches the GPU) runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
... bench.py ...` as a child and relays rank 0's JSON line; launched through torch.distributed.run directly it
reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Ranks talk RCCL (torch.distributed backend "nccl").

A step is one pass of the hot path over one batch of synthetic input that is already resident in HBM:

  sis           (default; BASELINE.json config 2) one sequential-importance-sampling step of 1024 particles per GPU
                on a GPT-2-small-shaped random-init fp32 model (README.md:82-98 of the reference): context dedup ->
                ragged-to-padded gather -> PyTorch-ROCm forward -> lm_head on the last position -> fused
                log-softmax + mask + logsumexp + sample -> particle bookkeeping (-> RCCL all-gather of the
                log-weights when N > 1).  --prefix-kv / --particle-kv select the KV variants (config 3 / beyond).
  sis-llama     (config 4) the same step with a Llama-3.2-1B-shaped random-init bf16 model, V = 128256,
                512 particles per GPU (4096 over 8 GPUs).
  kernel        only the fused step on [1024, 50257] fp32 logits, rotating over 4 buffers so that the 256 MiB
                Infinity Cache cannot serve the rows.
  kernel-llama  (config 5) only the fused step on [512, 128256] bf16 logits, 4 rotating buffers.
  api           the README loop through the backend's API, the population submitted as one batched call per step
                (`AsyncAmdLM.batch_next_token_step`: contexts as Python lists in, logZ / tokens out).
  api-coro      the README loop with lines 82-87 as one fused call: 1024 coroutines awaiting `AsyncAmdLM.next_token_step`.
  trie          token->byte trie masses of 1024 rows of gpt2-sized logits (trie/base.py:147-213,346-393,
                trie/parallel.py:92-145) through glb_trie_rows: --trie-out rows (all nodes, row-major) | slots | selected.
  api-readme    the README loop VERBATIM, only `llm` swapped (README.md:72-98): 1024 coroutines await
                `next_token_logprobs`, then add their mask, take logsumexp and draw with torch.multinomial themselves
                (user-side torch ops on the returned device rows) - what a user who changes nothing else gets.
  api-logprobs  `batch_next_token_logprobs` of 1024 contexts per step, log-prob rows materialised ([1024, V] fp32).
  plumbing      CPU / gloo self-test of the multi-rank launch, barrier, all-gather and JSON relay (tests only; no
                kernel is run and the line says so).

Rank 0 prints ONE JSON line.  `value` = particles per second over all GPUs (weak scaling).  `roofline` prices the
fused step: algorithmic bytes per call over the mean duration of its launch (fused_step_kernel), measured
inside the timed region with HIP events the launch carries as its own start / stop stamps, and - reported beside it -
with events recorded around the call.  `cpu_baseline` times the
reference-semantics CPU port (oracle/, test infrastructure) of the same work on a bounded sample on this host.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V_GPT2, V_LLAMA = 50257, 128256
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s
METRIC = "particles/sec + logprob-kernel HBM GB/s (% of 8 TB/s), 1024 particles gpt2"
_LINE_OUT = sys.stdout  # where the ONE JSON line goes (main() keeps the real stdout for it and sends everything else to stderr)
WORKLOADS = ["sis", "sis-llama", "kernel", "kernel-llama", "api", "api-coro", "api-readme", "api-logprobs", "trie", "plumbing"]


def algorithmic_bytes(B, V, elem_size, n_masks, mask_words, n_particles=None):
    """unique logits rows once + distinct mask bytes + 8 B of outputs per particle."""
    return B * V * elem_size + n_masks * mask_words * 4 + (B if n_particles is None else n_particles) * 8


# ------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = test infrastructure; only this leg of bench.py touches it)
# ------------------------------------------------------------------------------------------------------------
def _cpu_particle_math(x, masks, seed, repeats):
    from oracle import oracle as O

    V = x.shape[1]
    t0 = time.perf_counter()
    st = None
    done = 0
    for _ in range(repeats):
        for r in range(x.shape[0]):
            lp = O.ref_log_softmax(x[r])
            E, st = O.mt_exponential(seed, V, st)
            O.ref_particle(lp, masks[r % len(masks)], E)
            done += 1
    return done, time.perf_counter() - t0


def cpu_baseline(workload, sample_rows, seed=1234):
    """The same work on this host's cores, in the reference's arithmetic (oracle layer A = port of cache.py:96
    log_softmax and README.md:84-87 mask + logsumexp + exp + multinomial incl. its serial MT19937 draws; for the
    SIS workloads preceded by the torch-CPU forward hf.py:275-281 runs - full-vocabulary logits for every position,
    as the reference computes them).  Bounded sample, scaled to particles/s; the forward is timed at k = all cores
    and at k = 1, the particle math is single-threaded as in the reference."""
    from oracle import oracle as O
    from tests import synth

    O.build()
    ncpu = os.cpu_count()
    if workload in ("sis", "api", "api-coro", "api-readme", "api-logprobs"):
        from transformers import GPT2Config, GPT2LMHeadModel

        masks = synth.binary_masks(seed, 2, V_GPT2)
        rows = min(sample_rows, 256)
        torch.manual_seed(seed)
        model = GPT2LMHeadModel(GPT2Config()).eval()
        ids = torch.randint(0, V_GPT2, (rows, 13))  # mid-loop context length: prompt 8 + 5 generated
        k_def = torch.get_num_threads()
        k_all = ncpu or k_def  # all cores of the host, and one (+ torch's own default thread count)
        fwd, nrows = {}, {}
        # (all cores beyond torch's own default oversubscribe the host: a quarter of the sample shows it)
        for k, nrow in ((k_def, rows), (k_all, max(rows // 4, 8)), (1, 8)):
            if k in fwd:
                continue
            nrows[k] = nrow
            torch.set_num_threads(k)
            t0 = time.perf_counter()
            with torch.no_grad():
                logits = model(ids[:nrow]).logits  # [rows, 13, V], all positions (hf.py:275-281)
            fwd[k] = (time.perf_counter() - t0) / nrow
            if nrow == rows:
                x = logits[:, -1].contiguous().numpy()
        torch.set_num_threads(k_def)
        done, t_part = _cpu_particle_math(x, masks, seed, 1)
        per_particle = t_part / done
        by_k = {k: 1.0 / (t + per_particle) for k, t in fwd.items()}
        k_best = max((k for k in by_k if k > 1), key=lambda k: by_k[k], default=1)
        return {
            "value": by_k[k_best], "unit": "particles/s", "cores": k_best, "kind": "port",
            "value_k1": by_k[1], "value_by_threads": {str(k): v for k, v in sorted(by_k.items())},
            "sample": "torch-CPU gpt2-small forward over 13-token contexts, all-position logits: "
                      + ", ".join(f"{nrows[k]} rows on {k} thread{'s' if k > 1 else ''} ({fwd[k] * 1e3:.1f} ms/particle)"
                                  for k in sorted(fwd, reverse=True))
                      + f"; + per-particle log_softmax + mask + logsumexp + MT19937 multinomial (V={V_GPT2}, fp32) on {done} "
                        f"rows, single-threaded as in the reference ({per_particle * 1e3:.2f} ms/particle); value = the best "
                        f"multi-threaded run (k={k_best}), value_k1 = k=1; host has {ncpu} cores",
        }
    if workload == "sis-llama":
        from transformers import LlamaForCausalLM

        masks = synth.binary_masks(seed, 2, V_LLAMA)
        rows = min(sample_rows, 32)
        torch.manual_seed(seed)
        model = LlamaForCausalLM(llama_1b_config()).to(torch.bfloat16).eval()
        ids = torch.randint(0, V_LLAMA, (rows, 13))
        k_all = ncpu or torch.get_num_threads()
        torch.set_num_threads(k_all)
        t0 = time.perf_counter()
        with torch.no_grad():
            logits = model(ids).logits
        t_fwd = (time.perf_counter() - t0) / rows
        x = logits[:, -1].float().contiguous().numpy()
        done, t_part = _cpu_particle_math(x, masks, seed, 1)
        return {
            "value": 1.0 / (t_fwd + t_part / done), "unit": "particles/s", "cores": k_all, "kind": "port",
            "sample": f"torch-CPU Llama-3.2-1B-shaped bf16 forward, {rows} rows x 13 tokens on {k_all} threads "
                      f"({t_fwd * 1e3:.1f} ms/particle) + per-particle math on fp32-upcast rows (V={V_LLAMA}) "
                      f"{t_part / done * 1e3:.2f} ms/particle single-threaded; host has {ncpu} cores",
        }
    if workload == "trie":
        # the reference's per-row loop (trie/base.py:346-393 - numba-compiled there; here the oracle's C restatement of the
        # same loop, same order of additions) over the unfolded trie, one thread
        from genlm_backend_amd.tokenization import Token
        from genlm_backend_amd.trie import TokenByteTrie

        rs = np.random.default_rng(0)
        words, seen = [], set()
        while len(words) < V_GPT2:
            w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
            if w not in seen:
                seen.add(w)
                words.append(w)
        flat = TokenByteTrie([Token(i, w) for i, w in enumerate(words)]).flat()
        rows = min(sample_rows, 2048)
        ws = rs.random((rows, V_GPT2)).astype(np.float32)
        O.trie_reduce(ws[:8], flat, 0)
        t0 = time.perf_counter()
        O.trie_reduce(ws, flat, 0)
        dt = time.perf_counter() - t0
        return {"value": rows / dt, "unit": "particles/s", "cores": 1, "kind": "port",
                "sample": f"weight_sum of {rows} weight rows over the {flat['n_nodes']}-node trie of {V_GPT2} tokens (the reference's "
                          f"per-row loop, trie/base.py:346-393, restated in C) in {dt:.1f} s, single thread; host has {ncpu} cores"}
    # kernel-only analogues: per-particle math on synthetic rows
    V = V_GPT2 if workload == "kernel" else V_LLAMA
    rows = min(sample_rows, 1024 if workload == "kernel" else 256)
    masks = synth.binary_masks(seed, 2, V)
    x = synth.logits(seed, rows, V)
    if workload == "kernel-llama":
        x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()  # bf16 values, upcast as the reference would see them
    done, t_part = _cpu_particle_math(x, masks, seed, 8 if workload == "kernel" else 4)
    return {
        "value": done / t_part, "unit": "particles/s", "cores": 1, "kind": "port",
        "sample": f"{done} particle steps of per-particle log_softmax + mask + logsumexp + MT19937 multinomial "
                  f"(V={V}, {'fp32' if workload == 'kernel' else 'bf16 values in fp32'}) in {t_part:.1f} s, single "
                  f"thread (the reference's draws are serial); host has {ncpu} cores",
    }


def llama_1b_config():
    from transformers import LlamaConfig

    # Llama-3.2-1B architecture (16 layers, d 2048, 32 heads / 8 KV heads of 64, MLP 8192, tied embeddings, V 128256)
    return LlamaConfig(vocab_size=V_LLAMA, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                       num_attention_heads=32, num_key_value_heads=8, head_dim=64, max_position_embeddings=4096,
                       rope_theta=500000.0, rms_norm_eps=1e-5, tie_word_embeddings=True, bos_token_id=128000,
                       eos_token_id=128001)


# ------------------------------------------------------------------------------------------------------------
# multi-rank launch: the parent never initialises HIP
# ------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    else:
        sys.stdout.write(p.stdout)
    sys.exit(p.returncode if p.returncode else (0 if line is not None else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="sis", choices=WORKLOADS)
    ap.add_argument("--cpu-sample", type=int, default=1024, help="rows of the CPU baseline sample (upper bound)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--step-times", action="store_true", help="diagnostic: print the longest host-side steps of the timed region to stderr")
    ap.add_argument("--particle-kv", action="store_true",
                    help="sis workloads with device-resident per-particle KV (beyond the reference: one token per particle per step)")
    ap.add_argument("--prefix-kv", action="store_true",
                    help="sis workloads with the prompt's KV cached (hf.py:155-164 cache_kv; BASELINE config 3)")
    ap.add_argument("--prompts", type=int, default=1, help="distinct shared prompts over the population (config 3: 1 / 8 / 64)")
    ap.add_argument("--resample", action="store_true", help="systematic resampling after every step (replicated, deterministic)")
    ap.add_argument("--trie-out", choices=["rows", "slots", "selected", "rowsel", "rowsel-root"], default="rows",
                    help="trie workload: all nodes row-major / the folded trie's slots row-major / 4096 selected nodes / a selection "
                         "per row: the children of every row's own depth-1 node (a particle inside a token) / of the root (a particle "
                         "at a token boundary: every part of the trie)")
    ap.add_argument("--llm-gather", action="store_true",
                    help="api-coro / api-readme: run a step's coroutines with AsyncAmdLM.gather (advanced by hand, no asyncio Task per "
                         "particle) instead of asyncio.gather - the README's user code with ONE name changed")
    ap.add_argument("--auto-kv", action="store_true",
                    help="api workload: KV rows that follow the contexts handed to batch_next_token_step (beyond the reference: "
                         "one token per context per step instead of a re-encoding)")
    ap.add_argument("--device-batch", action="store_true",
                    help="api workload: the population lives on the device and is submitted as a padded [N, cap] int32 matrix + lengths "
                         "(AsyncAmdLM.batch_next_token_step_device); results stay on the device, the user-side bookkeeping is one "
                         "glb_particles_advance launch")
    ap.add_argument("--kv-gather", action="store_true",
                    help="--particle-kv: always gather the live KV rows into batch order (never run the forward on the slab in place)")
    ap.add_argument("--per-row-masks", action="store_true",
                    help="kernel workloads: one bit mask PER PARTICLE (GLB_MASK_BITS, n_masks == n_particles: what a grammar gives), "
                         "handed over raw every call - the call brings them into the kernels' layout itself; sis workloads: "
                         "one mask per particle, every particle its own reduction unit")
    ap.add_argument("--llama", choices=["3.2-1b", "3-8b"], default="3.2-1b",
                    help="sis-llama: the Llama-3.2-1B shape (config 4's per-GPU share, default) or the Llama-3-8B shape (config 5's model: "
                         "16 GB of bf16 weights, 512 particles)")
    ap.add_argument("--rng", choices=["philox", "parity"], default="philox",
                    help="kernel / sis workloads: in-kernel Philox draws (default), or the reference's draws - torch.multinomial's CPU "
                         "MT19937 stream (README.md:87), generated on the device (glb_mt19937_exponential_rows) and raced against "
                         "(GLB_RNG_NOISE): token ids identical to the reference's under a fixed seed")
    ap.add_argument("--logits", choices=["gaussian", "peaked"], default="gaussian",
                    help="kernel workloads: N(0, 3^2) rows (default) or real-shaped rows - top-1 probability 0.9 over a Zipf tail, "
                         "the top token forbidden by the row's mask in a third of the rows (the low-mass re-reduction path)")
    ap.add_argument("--mask", choices=["random", "eos-only"], default="random",
                    help="kernel workloads: two shared masks forbidding a random third of the vocabulary (default), or the README's "
                         "EOS-only mask (README.md:63-66: one allowed token) on every row")
    ap.add_argument("--mask-churn", type=float, default=1.0,
                    help="kernel workloads with --per-row-masks: the fraction of the particles whose mask changes between calls. 1 "
                         "(default): every call brings all masks into the kernels' layout (handed over raw); below 1: the masks are "
                         "prepared once and only that fraction is prepared again per call (glb_mask_prepare_rows)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="--gpus N on a box with ONE GPU: every rank computes on cuda:0 and the exchange goes over gloo through "
                         "host memory (RCCL wants a GPU per rank) - a rehearsal of the multi-rank code, not a measurement")
    ap.add_argument("--no-kv-line", action="store_true", help="default sis workload: do not also time the shared-KV-rows variant (value_kv)")
    ap.add_argument("--no-rccl-single", action="store_true",
                    help="N = 1 sis workloads: do NOT route the per-step exchange through a one-rank RCCL group")
    ap.add_argument("--gemms", choices=["recorded", "library", "tune"], default="recorded",
                    help="which of the GEMM library's own solutions PyTorch runs (genlm_backend_amd.gemm_tuning, torch.cuda.tunable): "
                         "recorded (default) - the shapes in genlm-backend_amd/tuned/<arch>.csv by their recorded solution, everything "
                         "else (and everything, if the file was made by another build of the libraries) by the library's default; "
                         "library - the defaults; tune - time the solutions of every new shape during the warm-up (seconds a shape) "
                         "and write --gemms-file at exit")
    ap.add_argument("--contract", choices=["poly", "hw", "auto"], default="auto",
                    help="arithmetic of the fused step's terms (include/glb.h GLB_STEP_HW_EXP): auto (default, the product's) - "
                         "v_exp_f32 for 16-bit logits (checked against the oracle by tolerance), the polynomial for float32 rows; hw - "
                         "v_exp_f32 for every element type; poly - the polynomial, bit for bit the oracle's")
    ap.add_argument("--gemms-file", default=None, help="--gemms tune: the file to extend (default: $TMPDIR/glb_tunableop.csv)")
    args = ap.parse_args()
    args.gemm_shapes = 0

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to file descriptor 1 when its
    # first communicator is made - everything but the line goes to stderr from here on
    global _LINE_OUT
    sys.stdout.flush()
    _LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (it starts its own "
                         "ranks) or torch.distributed.run with --nproc-per-node equal to --gpus")
    cpu_only = args.workload == "plumbing"
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cpu_only:
            dist.init_process_group("gloo")
        elif args.rehearse_one_gpu:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if cpu_only:
        return plumbing(args, rank, world, dist)
    dev = torch.device("cuda", local_rank)
    if args.workload not in ("kernel", "kernel-llama", "trie") and args.gemms == "tune":  # (here, in the rank's own process: the parent that starts the ranks never touches the GPU)
        from genlm_backend_amd import gemm_tuning

        gemm_tuning.record(args.gemms_file or os.path.join(os.environ.get("TMPDIR", "/tmp"), "glb_tunableop.csv"))
    # --gemms recorded goes through the product's own option, AsyncAmdLM(gemms="recorded") (= load_model_by_name(name,
    # llm_opts={"gemms": "recorded"})): what the line measures is what a user of the backend gets with that option
    lm_gemms = "recorded" if args.gemms == "recorded" else "library"
    torch.cuda.set_device(dev)

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.engine import HipEngine

    eng = HipEngine(dev, contract=args.contract)
    workload = args.workload
    # N = 1: the SIS workloads still run the multi-rank code - a one-rank "nccl" group carries the per-step all-gather
    # of log-weights (and the resampling exchange) through RCCL, exactly the calls an 8-GPU run makes
    force_coll, rccl_note = False, None
    if world == 1 and workload.startswith("sis") and not args.no_rccl_single:
        try:
            import socket

            import torch.distributed as dist1

            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            dist1.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            dist, force_coll = dist1, True
        except Exception as e:  # no RCCL for a single rank on this box: run without, and say so
            dist, rccl_note = None, f"one-rank nccl group unavailable: {type(e).__name__}: {e}"[:200]
    if workload == "trie":
        runner = TrieWorkload(eng, dev, rank, out=args.trie_out)
    elif workload in ("kernel", "kernel-llama"):
        runner = KernelWorkload(eng, dev, rank, world, dist, llama=workload == "kernel-llama", per_row_masks=args.per_row_masks,
                                logits=args.logits, mask_mode=args.mask, rng=args.rng, mask_churn=args.mask_churn)
    elif workload in ("api", "api-coro", "api-readme", "api-logprobs"):
        runner = ApiWorkload(eng, dev, rank, world, dist, logprobs=workload == "api-logprobs", coro=workload == "api-coro",
                             auto_kv=args.auto_kv, readme=workload == "api-readme", llm_gather=args.llm_gather,
                             device_batch=args.device_batch, gemms=lm_gemms)
    else:
        from genlm_backend_amd.sis import SisBenchWorkload

        runner = SisBenchWorkload(eng, dev, rank, world, dist, prefix_kv=args.prefix_kv, particle_kv=args.particle_kv,
                                  model=("llama-3-8b" if args.llama == "3-8b" else "llama-3.2-1b") if workload == "sis-llama" else "gpt2",
                                  n_particles=512 if workload == "sis-llama" else 1024, n_prompts=args.prompts,
                                  resample=args.resample, force_collectives=force_coll,
                                  kv_in_place=None if args.kv_gather else 0.75, per_particle_masks=args.per_row_masks,
                                  rng="torch" if args.rng == "parity" else "philox", gemms=lm_gemms)
    args.gemm_shapes = getattr(getattr(runner, "llm", None), "gemm_shapes", 0)

    rccl_ranks = None
    if dist is not None:  # one collective before the clock starts: proves every rank is on the RCCL communicator
        from genlm_backend_amd.sis import _gather_all, _reduce_all

        probe = torch.full((1,), float(rank), device=dev)
        got = torch.empty(world, device=dev)
        _gather_all(dist, got, probe)
        assert got.cpu().tolist() == [float(r) for r in range(world)]
        rccl_ranks = dist.get_world_size()

    def timed_run(r):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; the MAX over ranks."""
        for i in range(args.warmup):
            r.step(i, timed=False)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        clock = getattr(getattr(r, "sis", None), "coll_clock", None)
        if clock is not None and dist is not None:  # the run's collectives, timed (sis.CollectiveClock) from here on
            clock.start()
            r.sis.rows_moved_total = 0
        t0 = time.perf_counter()
        marks = []
        for i in range(args.steps):
            r.step(args.warmup + i, timed=True)
            if args.step_times:
                marks.append(time.perf_counter())
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0
        if args.step_times and marks:  # (diagnostic: where the host spent the timed region - enqueue times, not GPU times)
            d = np.diff(np.array([t0] + marks)) * 1e3
            worst = np.argsort(-d)[:5]
            print("longest host steps (ms):", [(int(k), round(float(d[k]), 3)) for k in worst], "median", round(float(np.median(d)), 4),
                  "sync tail", round((time.perf_counter() - marks[-1]) * 1e3, 3), file=sys.stderr)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            # every rank's own time of the timed region (its steps end when its last kernel ends; the closing barrier then
            # waits for the slowest): the spread says whether one rank holds the others up
            own = torch.tensor([dt_own], dtype=torch.float64, device=dev)
            every = torch.empty(dist.get_world_size(), dtype=torch.float64, device=dev)
            _gather_all(dist, every, own)
            r.rank_ms = [float(x) / args.steps * 1e3 for x in every.cpu().tolist()]
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            _reduce_all(dist, t, dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    dt = timed_run(runner)
    # what a scaling curve needs to explain itself: the collectives' share of a step on rank 0 (RCCL: device
    # time between two stream events around every call, waiting for the slowest rank included; the gloo rehearsal: host time
    # of the synchronous calls), the particles that changed ranks, every rank's own step time
    coll_info = None
    clock = getattr(getattr(runner, "sis", None), "coll_clock", None)
    if clock is not None and clock.on:
        coll_info = {
            "collective_us_per_step": clock.total_us() / args.steps,
            "calls_per_step": clock.calls / args.steps,
            "bytes_per_step": clock.bytes / args.steps,
            "rows_moved_per_step": runner.sis.rows_moved_total / args.steps,
            "timed_by": "stream events around each call (RCCL)" if clock._device_timed else "host clock around each call (gloo rehearsal)",
            "what": "all-gather of log-weights + active count per step (+ context hashes in parity mode); with --resample one "
                    "all_to_all_single of the particle rows that change ranks (+ one of their KV rows with private slabs)"}
        if getattr(runner, "rank_ms", None):
            coll_info["rank_ms_per_step"] = {"min": min(runner.rank_ms), "max": max(runner.rank_ms),
                                             "mean": float(np.mean(runner.rank_ms)), "ranks": runner.rank_ms}
        clock.on = False
    # The default line also carries the same loop with KV rows shared by equal contexts (`--particle-kv`: one new token
    # per distinct context per step instead of the reference's re-encoding; tokens proven equal to the reference's by
    # tests/test_host_gpu.py) as value_kv / ms_per_step_kv.  `value` itself stays BASELINE config 2's algorithm.
    kv_extra = None
    if (workload == "sis" and not (args.particle_kv or args.prefix_kv or args.per_row_masks or args.resample or args.rng != "philox")
            and not args.no_kv_line):
        from genlm_backend_amd.sis import SisBenchWorkload

        runner_kv = SisBenchWorkload(eng, dev, rank, world, dist, particle_kv=True, model="gpt2", n_particles=1024,
                                     n_prompts=args.prompts, force_collectives=force_coll, gemms=lm_gemms)
        dt_kv = timed_run(runner_kv)
        kv_extra = {"value_kv": runner_kv.particles_per_step * world * args.steps / dt_kv,
                    "ms_per_step_kv": dt_kv / args.steps * 1e3,
                    "config_kv": "the same loop with device-resident KV rows shared by particles of equal contexts (block table "
                                 "decided on the device, one new token per distinct context per step, in-place forward replayed "
                                 "from a hipGraph with glb_slab_attention): beyond the reference's re-encode-every-step algorithm; "
                                 "same tokens",
                    "kv_rows": dict(runner_kv.sis.kv_stats)}
        runner_kv.llm.close()
        del runner_kv

    # ... and `value` once more with the GEMM library's own choice of kernels (what AsyncAmdLM runs unless asked for the
    # recorded solutions): the default line carries both, so that neither hides behind the other
    lib_extra = None
    if workload == "sis" and kv_extra is not None and args.gemm_shapes:
        kern_first = runner.kernel_times_us()
        runner.llm.close()  # TunableOp off: the library's defaults from here on
        dt_lib = timed_run(runner)
        lib_extra = {"value_library_gemms": runner.particles_per_step * world * args.steps / dt_lib,
                     "ms_per_step_library_gemms": dt_lib / args.steps * 1e3}
        runner.keep_kernel_times(len(kern_first))  # the roofline block stays the first run's launches

    kern_us = runner.kernel_times_us()
    if rank == 0:
        total_particles = runner.particles_per_step * world * args.steps
        out = {
            "metric": METRIC,
            "value": total_particles / dt,
            "unit": "particles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": runner.dtype_name,
            "data": "synthetic",
            "config": runner.config(),
        }
        if args.workload not in ("kernel", "kernel-llama", "trie", "plumbing"):
            out["config"]["gemms"] = {
                "recorded": f"AsyncAmdLM(gemms=\"recorded\") = load_model_by_name(name, llm_opts={{\"gemms\": \"recorded\"}}): PyTorch "
                            f"TunableOp, {args.gemm_shapes} shapes of genlm-backend_amd/tuned/<arch>.csv run by their recorded "
                            "rocBLAS / hipBLASLt solution, the rest by the library's default; value_library_gemms = the same loop with "
                            "the backend's default option (gemms=\"library\")" if args.gemm_shapes else
                            "the library's default solutions (no recorded file for this build of the libraries)",
                "library": "the library's default solutions",
                "tune": "PyTorch TunableOp, tuned during the warm-up"}[args.gemms]
        if rccl_ranks is not None:
            if args.rehearse_one_gpu and world > 1:  # every rank on cuda:0, exchange over gloo: NOT a measurement
                out["rehearsal_one_gpu"] = True
                out["gloo_ranks"] = rccl_ranks
            else:
                out["rccl_ranks"] = rccl_ranks
        if rccl_note is not None:
            out["rccl_note"] = rccl_note
        if coll_info is not None:
            out["collectives"] = coll_info
        if kv_extra is not None:
            out.update(kv_extra)
        if lib_extra is not None:
            out.update(lib_extra)
        if kern_us is not None and len(kern_us):
            ach = runner.kernel_bytes / (np.mean(kern_us) * 1e-6) / 1e9
            each_bytes = getattr(runner, "kernel_bytes_each", None)
            if each_bytes is None or len(each_bytes) != len(kern_us):
                each_bytes = np.full(len(kern_us), float(runner.kernel_bytes))
            outer = runner.outer_times_us()
            traffic, traffic_src = pmc_traffic(workload + ("-rowmasks" if args.per_row_masks and workload.startswith("kernel") else "")
                                               + (f"-{args.trie_out}" if workload == "trie" else ""))
            out["roofline"] = {
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                # the median over the timed launches of (that launch's bytes / that launch's time): a loop's launches differ in
                # size (SIS step 0 reads one shared row), so bytes and time are paired per launch, never a mean over a median
                "frac_median": float(np.median(each_bytes / (kern_us * 1e-6))) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic,
                # (counters cannot be read from inside the process: NOT measured in this run, but by an earlier rocprofv3
                # --pmc pass of this same command whose summary is committed under profiles/)
                "traffic_source": (f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this "
                                   "command, not this run") if traffic_src else None,
                "kernel": getattr(runner, "roofline_kernel", None) or (
                          "glb::fused_step_kernel (fused log-softmax + mask + logsumexp + sample in ONE launch: stats waves "
                          "stream the rows chunk by chunk, finishing waves at the end of the grid fold the tagged records "
                          "and draw); the few calls too small for it (one shared row: SIS step 0) run "
                          "glb::chunk_stats_small_kernel + glb::finish_kernel and are timed first start to last stop"
                          + ("; one raw bit mask per particle: the stats waves of that launch read the caller's bit rows themselves "
                             "(no glb::mask_prepare_kernel launch; the parity draw's two-launch form still prepares them, inside "
                             "the span)" if args.per_row_masks else "")),
                "timing": getattr(runner, "roofline_timing", None) or (
                          "every fused call of the timed region, none left out: HIP events carried by the launch itself as "
                          "its start / stop stamps (hipExtLaunchKernel through glb_logprob_mask_sample_timed) = the launch "
                          "duration rocprofv3 reports; *_outer_events = the same calls between two hipEventRecord markers "
                          "on the stream (adds the marker packets)"),
                "bytes_per_launch": runner.kernel_bytes,
                "us_per_launch_mean": float(np.mean(kern_us)),
                "us_per_launch_median": float(np.median(kern_us)),
                "us_per_launch_outer_events_mean": float(np.mean(outer)) if len(outer) else None,
                "frac_outer_events": (runner.kernel_bytes / (np.mean(outer) * 1e-6) / 1e9 / HBM_PEAK_GBS) if len(outer) else None,
                "launches_timed": len(kern_us),
            }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(workload, args.cpu_sample)
        print(json.dumps(out), file=_LINE_OUT, flush=True)
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic(workload):
    """HBM bytes per fused call from the rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE in
    separate runs, corrected as MI355X_MICROARCH.md prescribes; collected with tools/profile_round.sh - counters
    cannot be read from inside the process).  The newest summary under profiles/ for this workload, else null."""
    import glob

    try:
        def order(path):  # profiles/rNN/<workload>_pmc_traffic_vM.json: latest round, then latest version
            import re

            m = re.search(r"r(\d+)[/\\][^/\\]*?(?:_v(\d+))?\.json$", path)
            return (int(m.group(1)), int(m.group(2) or 0)) if m else (-1, -1)

        prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{workload}_pmc_traffic*.json")), key=order)
        if prof:
            t = json.load(open(prof[-1]))
            return t["hbm_read_bytes_per_launch_corrected"] + t["hbm_write_bytes_per_launch"], os.path.relpath(prof[-1], ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def plumbing(args, rank, world, dist):
    """Multi-rank skeleton on CPU / gloo: barrier, per-step all-gather of fake log-weights, max-over-ranks clock,
    one JSON line from rank 0.  No kernel of the product runs here; the line is labelled accordingly."""
    lw = torch.full((8,), float(rank))
    gathered = torch.empty(8 * world)

    def step():
        if dist is not None:
            dist.all_gather_into_tensor(gathered, lw)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert gathered.view(world, 8)[:, 0].tolist() == [float(r) for r in range(world)]
    if rank == 0:
        print(json.dumps({"metric": "plumbing self-test (no kernel run; NOT a benchmark result)", "value": 8 * world * args.steps / dt,
                          "unit": "fake particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "plumbing self-test on CPU/gloo"},
                          "rccl_ranks": world if dist is not None else None}), file=_LINE_OUT, flush=True)
    if dist is not None:
        dist.destroy_process_group()


class TrieWorkload:
    """Token -> byte trie masses of 1024 rows of [1024, 50257] fp32 logits + their lse, straight from the
    logits (no log-prob matrix), on a synthetic gpt2-sized vocabulary of byte strings (tools/tbench.py's): one
    `TokenByteTrie.masses_from_logits` call per step through glb_trie_rows (a row of a part of the folded trie resident in
    LDS).  Algorithmic bytes: the logits once + the result once."""

    dtype_name = "f32"

    def __init__(self, eng, dev, rank, out="rows", B=1024, V=V_GPT2, nbuf=3):
        from genlm_backend_amd.tokenization import Token
        from genlm_backend_amd.trie import TokenByteTrie

        rs = np.random.default_rng(0)
        words, seen = [], set()
        while len(words) < V:
            w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
            if w not in seen:
                seen.add(w)
                words.append(w)
        self.trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=eng)
        self.plan = self.trie.slot_plan() if out in ("rows", "slots", "rowsel-root") else self.trie.plan()  # (the plan the timed call runs on)
        g = torch.Generator(device=dev)
        g.manual_seed(77 + rank)
        self.bufs = [torch.randn((B, V), device=dev, generator=g) * 3.0 for _ in range(nbuf)]
        self.lse = [eng.row_lse(x) for x in self.bufs]
        self.out_kind, self.B, self.V, self.eng = out, B, V, eng
        self.particles_per_step = B
        self.sel = torch.from_numpy(rs.choice(len(self.trie), 4096, replace=False).astype(np.int32)).to(dev)
        if out == "selected":  # one selection asked for again and again: its sub-forest is planned once, up front (host work)
            self.trie.prepare_selection(self.sel)
            sp = self.trie.selection_plan(self.sel, sweep=True) if B >= self.trie.SWEEP_MIN_ROWS else None
            self.plan = sp or self.trie.selection_plan(self.sel) or self.plan  # (the plan the timed call runs on)
        # a selection per row: the children (<= 256) of the row's current node - one of the root's children, or the root itself
        d1 = sorted(self.trie.children[self.trie.root].values())
        cur = [self.trie.root] * B if out == "rowsel-root" else [d1[int(k)] for k in rs.integers(0, len(d1), B)]
        K = max(len(self.trie.jump[c]) for c in set(cur))
        rowsel = np.full((B, K), -1, np.int32)
        for r, c in enumerate(cur):
            rowsel[r, :len(self.trie.jump[c])] = self.trie.jump[c]
        self.rowsel = torch.from_numpy(rowsel).to(dev)
        width = {"rows": len(self.trie), "slots": self.plan["n_slots"], "selected": 4096, "rowsel": K, "rowsel-root": K}[out]
        self.kernel_bytes = B * V * 4 + B * width * 4 + B * 4
        self.outer = []
        if self.plan.get("sweep"):
            self.roofline_kernel = ("(anonymous)::trie_sweep_kernel + trie_rows_kernel<top> (glb_trie_rows: a persistent workgroup per part "
                                    "of the folded trie reads row after row front to back, its tokens' slots and the part's internal "
                                    "nodes in registers, the part's values in LDS - reduced depth by depth in the reference's order, "
                                    "the part's run of the output row written; then the few nodes above the parts)")
        else:
            self.roofline_kernel = ("(anonymous)::trie_rows_kernel x 2 launches (glb_trie_rows: a workgroup per (row, part) of the "
                                    "folded trie - leaves gathered from the row, reduced depth by depth in LDS in the reference's order, "
                                    "the part's run of the output row written; then the few nodes above the parts)")
        self.roofline_timing = ("every call of the timed region between two hipEventRecord markers on the stream (both launches "
                                "and, for selected nodes, the slot look-up launch inside the span)")

    def step(self, i, timed):
        x, lse = self.bufs[i % len(self.bufs)], self.lse[i % len(self.bufs)]
        kw = {"rows": dict(layout="rows"), "slots": dict(layout="slot_rows"), "selected": dict(nodes=self.sel),
              "rowsel": dict(nodes=self.rowsel), "rowsel-root": dict(nodes=self.rowsel, wide_selections=True)}[self.out_kind]
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.trie.masses_from_logits(x, lse, **kw)
            e1.record()
            self.outer.append((e0, e1))
        else:
            self.trie.masses_from_logits(x, lse, **kw)

    def kernel_times_us(self):
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self.outer])

    def outer_times_us(self):
        return self.kernel_times_us()

    def config(self):
        pl = self.plan
        return {"workload": f"token->byte trie masses of {self.B} rows of [{self.B}, {self.V}] fp32 logits + lse -> "
                            f"{ {'rows': 'all nodes, row-major', 'slots': 'the folded trie slots, row-major', 'selected': '4096 selected nodes (only the subtrees below them planned)', 'rowsel': 'a selection per row: the children of the row-s own depth-1 node', 'rowsel-root': 'a selection per row: the children of the root (every part; wide_selections: the sweep plan)'}[self.out_kind] }"
                            f" (glb_trie_rows; {len(self.trie)} nodes, {pl['n_slots']} slots in {pl['n_parts']} parts of <= {pl['max_local']}; "
                            "synthetic vocabulary of 1-8 letter byte strings)",
                "rows_per_gpu": self.B, "vocab": self.V}


class KernelWorkload:
    """Fused step only: [1024, 50257] fp32 (or [512, 128256] bf16) logits, two shared {0,-inf} masks prepared once
    (like the README's two masks), mask ids per row, in-kernel Philox."""

    def __init__(self, eng, dev, rank, world, dist, llama=False, nbuf=4, per_row_masks=False, logits="gaussian",
                 mask_mode="random", rng="philox", mask_churn=1.0):
        self.eng, self.dev, self.rank, self.world, self.dist = eng, dev, rank, world, dist
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        B, V, dt = (512, V_LLAMA, torch.bfloat16) if llama else (1024, V_GPT2, torch.float32)
        self.B, self.V, self.llama = B, V, llama
        self.dtype_name = "bf16" if llama else "f32"
        self.particles_per_step = B
        self.per_row_masks = per_row_masks
        self.logits_kind, self.mask_mode = logits, mask_mode
        n_masks = B if per_row_masks else 2
        if mask_mode == "eos-only":  # README.md:63-66 `eos_one_hot.log()`: one allowed token
            maskf = torch.full((n_masks, V), float("-inf"), device=dev)
            maskf[:, V - 1] = 0.0
        else:
            maskf = torch.where(torch.rand((n_masks, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
            maskf[:, 0] = 0.0
        if logits == "peaked":
            self.bufs = [self._peaked_rows(B, V, g, maskf).to(dt) for _ in range(nbuf)]
        else:
            self.bufs = [(torch.randn((B, V), device=dev, generator=g) * 3.0).to(dt) for _ in range(nbuf)]
        self.bits, _ = eng.mask_to_bits(maskf)
        del maskf
        self.churn = float(mask_churn) if per_row_masks else 1.0
        self.incremental = per_row_masks and self.churn < 1.0
        self.masks = eng.prepare_masks(self.bits, V, dt) if (self.incremental or not per_row_masks) else None
        self.n_changed = int(round(self.churn * B)) if self.incremental else 0
        self.changed = [torch.from_numpy(np.random.default_rng(5 + k).choice(B, self.n_changed, replace=False).astype(np.int32)).to(dev)
                        for k in range(nbuf)] if self.n_changed else None
        self.mask_id = (torch.arange(B, device=dev) % 2).to(torch.int32)
        self.out = (torch.empty(B, device=dev), torch.empty(B, device=dev),
                    torch.empty(B, dtype=torch.int32, device=dev))
        self.lw = torch.zeros(B, device=dev)
        self.gathered = torch.empty(B * world, device=dev) if world > 1 else None
        self.kernel_bytes = algorithmic_bytes(B, V, 2 if llama else 4, self.n_changed if self.incremental else n_masks, (V + 31) // 32)
        self.parity = rng == "parity"
        self.noise_src = self.noise_buf = None
        if self.parity:
            # torch's CPU generator on the device: one Exp(1) row per particle per step (+ B V 4 bytes read)
            self.noise_src = eng.noise_rng(1234 + rank, V)
            self.noise_buf = torch.empty((B, V), dtype=torch.float32, device=dev)
            self.kernel_bytes += B * V * 4
            self.roofline_kernel = ("(anonymous)::mt_jump_kernel x 2 + mt_rows_kernel (glb_mt19937_exponential_rows: torch's CPU MT19937 "
                                    "stream entered at every particle's row, [B, V] float32 noise written) + glb::chunk_stats_kernel / "
                                    "finish_kernel in GLB_RNG_NOISE mode (the race p / E over the masked row)")
            self.roofline_timing = ("every call of the timed region between two hipEventRecord markers on the stream: noise generation + "
                                    "the step; bytes = logits + noise read once + masks + outputs (the noise is also WRITTEN once, "
                                    "which the byte count leaves out)")
        self.events = []
        self.outer = []
        self._pool = []
        # the argument block of every buffer's call is filled once: a step's host work is then a few microseconds, so
        # the host stays ahead of the GPU and the event interval holds no wait for the next launch packet
        # set-up, not measurement: what building the synthetic rows left in the allocator's cache (GBs of argsort / rand
        # temporaries for the peaked rows) goes back to the driver before the clock starts
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()
        draw = dict(rng_mode=2, noise=self.noise_buf) if self.parity else dict(rng_mode=1, seed=1234, offset=0)
        if self.incremental:  # one mask per particle, prepared once: a call prepares again only the masks that changed
            self.own_id = torch.arange(B, dtype=torch.int32, device=dev)
            self.plans = [eng.step_plan(x, mask=self.masks, mask_id=self.own_id, particle_base=rank * B, out=self.out, **draw)
                          for x in self.bufs]
        elif per_row_masks:  # raw bit rows, one per particle (mask ids = identity): the fused launch reads them as they are
            self.plans = [eng.step_plan(x, mask_kind=1, mask=self.bits, particle_base=rank * B, out=self.out, **draw)
                          for x in self.bufs]
        else:
            self.plans = [eng.step_plan(x, mask=self.masks, row_mask_id=self.mask_id, particle_base=rank * B, out=self.out, **draw)
                          for x in self.bufs]

    def _peaked_rows(self, B, V, g, maskf, p_top=0.9, zipf=1.1):
        """Real-shaped next-token rows: one token holds `p_top` of the mass, the others fall off as rank^-zipf in a random
        order; in every third row the top token sits where the row's mask (row % n_masks) forbids it - a model that is
        sure of a token the constraint rules out, constrained decoding's everyday case -, elsewhere where it is allowed."""
        dev = self.dev
        rank = torch.argsort(torch.rand((B, V), device=dev, generator=g), dim=1) + 1  # a permutation of 1..V per row
        x = -zipf * torch.log(rank.to(torch.float32))
        tail = float((torch.arange(2, V + 1, dtype=torch.float64) ** -zipf).sum())
        top_logit = float(np.log(p_top / (1.0 - p_top) * tail))
        rows = torch.arange(B, device=dev)
        top = (rank == 1).to(torch.int32).argmax(dim=1)
        allowed = maskf[rows % maskf.shape[0]] == 0  # [B, V]
        want_forbidden = (rows % 3 == 0) & (~allowed).any(dim=1)
        score = torch.rand((B, V), device=dev, generator=g)
        target = torch.where(want_forbidden[:, None], ~allowed, allowed).to(torch.float32) * (1.0 + score)
        col = target.argmax(dim=1)  # a random column of the wanted kind
        x[rows, top] = x[rows, col]
        x[rows, col] = top_logit
        return x

    def step(self, i, timed):
        plan = self.plans[i % len(self.plans)]
        if timed:
            # two clocks (see roofline.timing): events the launch carries as its own start / stop stamps, and a pair
            # recorded around the call
            if len(self._pool) < 2:
                self._pool = self.eng.timing_events(128)
            inner, (e0, e1) = self._pool.pop(), self._pool.pop()
            e0.record()
            if self.parity:
                self.noise_src.rows(self.B, out=self.noise_buf)
            if self.n_changed:
                self.eng.update_prepared_masks(self.masks, self.bits, self.changed[i % len(self.changed)])
            plan.run_timed(inner, offset=i)
            e1.record()
            self.events.append(inner)
            self.outer.append((e0, e1))
        else:
            if self.parity:
                self.noise_src.rows(self.B, out=self.noise_buf)
            if self.n_changed:
                self.eng.update_prepared_masks(self.masks, self.bits, self.changed[i % len(self.changed)])
            plan.run(offset=i)
        self.lw += self.out[0]
        if self.world > 1:
            from genlm_backend_amd.sis import _gather_all

            _gather_all(self.dist, self.gathered, self.lw)
            self.eng.normalize_weights(self.gathered)

    def kernel_times_us(self):
        if self.parity or self.n_changed:  # the span that holds the noise generation / the masks' update too
            return self.outer_times_us()
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self.events])

    def race_times_us(self):
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self.events])

    def outer_times_us(self):
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self.outer])

    def config(self):
        shape = "512 particles x Llama vocab 128256, bf16 logits [512,128256]" if self.llama else \
            "1024 particles x gpt2 vocab 50257, fp32 logits [1024,50257]"
        masks = (f"{self.B} bit masks, one per particle (GLB_MASK_BITS, handed over raw: read by the fused launch's stats waves "
                 "as they are; the parity draw's two-launch form prepares them first, inside the timed span)") \
            if self.per_row_masks else "2 shared bit masks (prepared once)"
        if self.incremental:
            masks = (f"{self.B} bit masks, one per particle, prepared once; {self.n_changed} of them ({self.churn:.0%}) change and are "
                     "prepared again before every call (glb_mask_prepare_rows, inside the timed span)")
        if self.mask_mode == "eos-only":
            masks += ", EOS-only (one allowed token, README.md:63-66) on every row"
        rows = ("N(0, 3^2) logits" if self.logits_kind == "gaussian" else
                "peaked logits (top-1 p = 0.9 over a Zipf(1.1) tail in random order; the top token forbidden by the row's mask in "
                "every third row)")
        draw = ("the reference's draw: torch.multinomial's CPU MT19937 stream generated on the device, first argmax p / E "
                "(ids identical to torch's)" if self.parity else "Philox draw")
        extra = {"us_race_only_mean": float(np.mean(self.race_times_us()))} if self.parity and self.events else {}
        return {"workload": f"fused step only: {shape} ld=V, {rows}, {masks}, {draw}, 4 rotating logits buffers",
                "particles_per_gpu": self.B, "vocab": self.V, "rng": "parity (torch CPU generator on the device)" if self.parity else "philox",
                "logits": self.logits_kind, "mask": self.mask_mode,
                "contract": ("hardware exponential (v_exp_f32; GLB_STEP_HW_EXP)"
                             if self.eng.contract == "hw" or (self.llama and self.eng.contract == "auto")
                             else "polynomial exponential (bit for bit the oracle's)"), **extra}


class ApiWorkload:
    """The reference's own API surface at 1024 particles on the GPT-2-small-shaped model.

    api:          README.md:72-98 - one coroutine per particle awaiting `next_token_step(context, mask_id)`; a step is
                  one `asyncio.gather` over the population (autobatched into one evaluation, batch_size = 1024).
    api-logprobs: `batch_next_token_logprobs` (base.py:47-60) of 1024 distinct contexts: [1024, V] fp32 log-prob rows
                  materialised on the device each step (trie cache cleared between steps so nothing is served from it)."""

    dtype_name = "f32"

    def __init__(self, eng, dev, rank, world, dist, logprobs=False, coro=False, n_particles=1024, max_tokens=10,
                 auto_kv=False, readme=False, llm_gather=False, device_batch=False, gemms="library"):
        import asyncio

        from transformers import GPT2Config

        from genlm_backend_amd.llm import AsyncAmdLM

        self.asyncio = asyncio
        cfg = GPT2Config()
        self.llm = AsyncAmdLM.from_config(cfg, None, device=dev, dtype=torch.float32, seed=1234, engine=eng,
                                          batch_size=n_particles, timeout=0.02, gemms=gemms,
                                          auto_kv_rows=n_particles + n_particles // 4 if auto_kv else 0, auto_kv_cap=24)
        self.auto_kv = auto_kv
        self.llm_gather = llm_gather
        self.device_batch = device_batch and not (logprobs or coro or readme)
        self.eng = eng
        V = cfg.vocab_size
        g = torch.Generator(device=dev)
        g.manual_seed(4321)
        valid = torch.where(torch.rand(V, device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
        valid[cfg.eos_token_id] = 0.0
        eos1 = torch.full((V,), float("-inf"), device=dev)
        eos1[cfg.eos_token_id] = 0.0
        self.llm.register_masks(torch.stack([valid, eos1]))
        self.user_masks = torch.stack([valid, eos1])  # api-readme: the user's own mask tensors (README.md:57-70)
        self.readme = readme
        self.llm.set_rng("philox", seed=1234 + rank)
        self.N, self.V, self.max_tokens, self.eos = n_particles, V, max_tokens, cfg.eos_token_id
        self.particles_per_step = n_particles
        self.logprobs, self.coro = logprobs, coro
        self.prompt = list(range(100, 108))
        self.loop = asyncio.new_event_loop()
        self.kernel_bytes = None
        self.world, self.dist, self.dev = world, dist, dev
        self.gathered = torch.empty(n_particles * world, device=dev) if world > 1 else None
        self._reset()
        rs = np.random.default_rng(7 + rank)
        self.ctx_pool = [[int(t) for t in rs.integers(0, V, 13)] for _ in range(n_particles)]
        # set-up, not measurement: allocator growth, GEMM selection - and, with KV rows, the hipGraph captures of the in-place
        # forward (kv.SlabForward captures a slab set's launch sequence at its third call; the second set's capture used to land
        # in the middle of the timed region: one 41 ms host step in thirty, `--step-times`, 4.9 instead of 3.5 ms a step)
        for i in range(3 * (max_tokens + 1) if auto_kv else 2):
            self.step(i, False)
        self._reset()

    def _reset(self):
        from genlm_backend_amd.sis import Particle

        if self.device_batch:  # the user's population as device tensors: [N, cap] tokens, lengths, active flags, log-weights
            N, P = self.N, len(self.prompt)
            self.cap = P + self.max_tokens + 1
            ctx = np.zeros((N, self.cap), np.int32)
            ctx[:, :P] = self.prompt
            self.d_ctx = torch.from_numpy(ctx).to(self.dev)
            self.d_len = torch.full((N,), P, dtype=torch.int32, device=self.dev)
            self.d_act = torch.ones(N, dtype=torch.int32, device=self.dev)
            self.d_lw = torch.zeros(N, dtype=torch.float32, device=self.dev)
            self.d_one = torch.ones(N, dtype=torch.int32, device=self.dev)
            self.t = 0
            return
        sel = lambda context: 1 if len(context) >= self.max_tokens else 0
        if self.readme:
            self.llm.clear_cache()
            llm, masks, prompt, eos = self.llm, self.user_masks, self.prompt, self.eos

            class ReadmeParticle:  # README.md:72-91, verbatim user code
                def __init__(self):
                    self.context, self.log_weight, self.active = [], 0.0, True

                async def extend(self):
                    logps = await llm.next_token_logprobs(prompt + self.context)
                    masked = logps + masks[sel(self.context)].to(logps.device)
                    logZ = masked.logsumexp(dim=-1)
                    p = (masked - logZ).exp()
                    next_token_id = torch.multinomial(p, 1).item()
                    self.log_weight += logZ
                    if next_token_id == eos:
                        self.active = False
                    else:
                        self.context.append(next_token_id)

            self.particles = [ReadmeParticle() for _ in range(self.N)]
        else:
            self.particles = [Particle(self.llm, sel, self.prompt, self.eos) for _ in range(self.N)]
        self.t = 0

    def step(self, i, timed):
        aio = self.asyncio
        if self.logprobs:
            self.llm.clear_cache()
            rows = self.loop.run_until_complete(self.llm.batch_next_token_logprobs(self.ctx_pool))
            assert rows.shape == (self.N, self.V)
            return
        if self.t >= self.max_tokens:
            self._reset()

        if self.device_batch:  # README.md:82-91 on device tensors: one batched call, one bookkeeping launch, nothing on the host
            P = len(self.prompt)
            mask_ids = ((self.d_len - P) >= self.max_tokens).to(torch.int32)
            len_eff = torch.where(self.d_act > 0, self.d_len, self.d_one)  # (a finished particle: its one-token stub)
            logZ, tok = self.llm.batch_next_token_step_device(self.d_ctx, len_eff, mask_ids)
            self.eng.particles_advance(self.d_ctx, self.d_len, self.d_act, self.d_lw, logZ, tok, self.eos, self.cap)
        elif self.coro or self.readme:
            gather = self.llm.gather if self.llm_gather else aio.gather

            async def one_step():
                await gather(*[p.extend() for p in self.particles if p.active])

            self.loop.run_until_complete(one_step())
        else:  # README.md:82-91 for every active particle, lines 82-87 as one batched call
            act = [p for p in self.particles if p.active]
            logZ, tok = self.llm.batch_next_token_step_sync([p.prompt_ids + p.context for p in act],
                                                            [p.mask_selector(p.context) for p in act])
            for p, z, t in zip(act, logZ.tolist(), tok.tolist()):
                p.log_weight += z
                if t == p.eos_id or t < 0:
                    p.active = False
                else:
                    p.context.append(t)
        self.t += 1
        if self.world > 1:
            from genlm_backend_amd.sis import _gather_all

            lw = self.d_lw if self.device_batch else \
                torch.tensor([p.log_weight for p in self.particles], dtype=torch.float32, device=self.dev)
            _gather_all(self.dist, self.gathered, lw)

    def kernel_times_us(self):
        return None

    def config(self):
        what = ("batch_next_token_logprobs of 1024 distinct 13-token contexts, [1024, V] fp32 log-prob rows materialised "
                "on the device (base.py:47-60)") if self.logprobs else \
            (("README loop VERBATIM (README.md:72-98, only `llm` swapped): 1024 coroutines await next_token_logprobs, then add "
              "their mask, take logsumexp and draw with torch.multinomial on the returned device rows themselves" if self.readme else
              "README loop: 1024 coroutines awaiting AsyncAmdLM.next_token_step, autobatched (batch_size 1024)" if self.coro else
              "README loop over 1024 Python-side particles, each step's requests submitted as one "
              "AsyncAmdLM.batch_next_token_step call") + ", prompt len 8, <=10 new tokens, 2 shared bit masks, Philox draws "
             "(README.md:72-98)")
        extra = {}
        if self.llm_gather and (self.coro or self.readme):
            what += "; the coroutines of a step run by AsyncAmdLM.gather (no asyncio Task per particle) instead of asyncio.gather"
        if self.device_batch:
            what += ("; the population lives on the device: a padded [N, cap] int32 matrix + lengths in, device tensors out "
                     "(AsyncAmdLM.batch_next_token_step_device), bookkeeping by one glb_particles_advance launch")
        if self.auto_kv:
            what += "; KV rows follow the contexts (AsyncAmdLM(auto_kv_rows=...): beyond the reference)"
            extra["auto_kv"] = dict(self.llm._auto_kv.stats)
        return {"workload": what + "; gpt2-small shape (random init, fp32)", "particles_per_gpu": self.N, "vocab": self.V, **extra}


if __name__ == "__main__":
    main()
