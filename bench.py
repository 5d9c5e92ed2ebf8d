#!/usr/bin/env python3
"""bench.py — headline benchmark of the genlm-backend hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sis|kernel]

One process per GPU (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`,
RCCL through torch.distributed backend "nccl").  A step is one pass of the hot path over one batch
of synthetic input that is already resident in HBM:

  workload "sis"    (default) one sequential-importance-sampling step of 1024 particles per GPU on
                    a GPT-2-small-shaped random-init model (README.md:82-98 of the reference):
                    context dedup -> ragged-to-padded gather -> PyTorch-ROCm forward -> lm_head on
                    the last position -> fused log-softmax + mask + logsumexp + sample kernel ->
                    particle bookkeeping (-> RCCL all-gather of log-weights when N > 1).
  workload "kernel" only the fused kernel on [1024, 50257] fp32 logits, rotating over 4 buffers so the
                    256 MiB Infinity Cache cannot serve the rows.

Rank 0 prints ONE JSON line.  `value` = particles per second over all GPUs (weak scaling: 1024
particles per GPU).  `roofline` prices the fused kernel: algorithmic bytes per launch
over its mean launch duration, measured with HIP events on the launch stream inside the timed
region.  `cpu_baseline` times the reference-semantics CPU port (oracle/, test infrastructure) of
the same particle step on a bounded sample, single thread, on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V_GPT2 = 50257
N_PARTICLES = 1024
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def algorithmic_bytes(B, V, elem_size, n_masks, mask_words):
    """logits once + distinct mask bytes + 8 B of outputs per particle."""
    return B * V * elem_size + n_masks * mask_words * 4 + B * 8


def cpu_baseline(workload, sample_rows, seed=1234):
    """The same work on this host's cores, in the reference's arithmetic (oracle layer A = port of
    cache.py:96 log_softmax and README.md:84-87 mask + logsumexp + exp + multinomial incl. its serial
    MT19937 draws; for "sis" preceded by the torch-CPU forward hf.py:275-281 runs, full-vocabulary logits
    for every position as the reference computes them).  Bounded sample, scaled to particles/s."""
    from oracle import oracle as O
    from tests import synth

    O.build()
    masks = synth.binary_masks(seed, 2, V_GPT2)
    threads = 1
    t_fwd = 0.0
    if workload == "sis":
        from transformers import GPT2Config, GPT2LMHeadModel

        rows = min(sample_rows, 512)
        torch.manual_seed(seed)
        model = GPT2LMHeadModel(GPT2Config()).eval()
        ids = torch.randint(0, V_GPT2, (rows, 13))  # mid-loop context length: prompt 8 + 5 generated
        threads = torch.get_num_threads()
        t0 = time.perf_counter()
        with torch.no_grad():
            logits = model(ids).logits  # [rows, 13, V], all positions (hf.py:275-281)
        t_fwd = time.perf_counter() - t0
        x = logits[:, -1].contiguous().numpy()
        repeats = 1
    else:
        rows = sample_rows
        x = synth.logits(seed, rows, V_GPT2)
        repeats = 8
    t0 = time.perf_counter()
    st = None
    done = 0
    for _ in range(repeats):
        for r in range(rows):
            lp = O.ref_log_softmax(x[r])
            E, st = O.mt_exponential(seed, V_GPT2, st)
            O.ref_particle(lp, masks[r % 2], E)
            done += 1
    t_part = time.perf_counter() - t0
    dt = t_fwd * repeats + t_part
    what = (f"torch-CPU gpt2-small forward [{rows}x13 tokens, all-position logits] {t_fwd:.1f} s on {threads} threads + "
            if workload == "sis" else "")
    return {
        "value": done / dt,
        "unit": "particles/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} particle steps: {what}per-particle log_softmax + mask + logsumexp + MT19937 multinomial "
                  f"(V={V_GPT2}, fp32) {t_part:.1f} s single-threaded; host has {os.cpu_count()} cores",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=["sis", "kernel"])
    ap.add_argument("--cpu-sample", type=int, default=1024, help="rows of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--particle-kv", action="store_true",
                    help="sis workload with device-resident per-particle KV (beyond the reference: one token per particle per step)")
    ap.add_argument("--prefix-kv", action="store_true",
                    help="sis workload with the prompt's KV cached (hf.py:155-164 cache_kv; BASELINE config 3)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.engine import HipEngine

    eng = HipEngine(dev)
    workload = args.workload
    if workload is None:
        try:
            from genlm_backend_amd import sis  # noqa: F401
            workload = "sis"
        except ImportError:
            workload = "kernel"

    if workload == "kernel":
        runner = KernelWorkload(eng, dev, rank, world, dist)
    else:
        from genlm_backend_amd.sis import SisBenchWorkload

        runner = SisBenchWorkload(eng, dev, rank, world, dist, prefix_kv=args.prefix_kv, particle_kv=args.particle_kv)

    for i in range(args.warmup):
        runner.step(i, timed=False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        runner.step(args.warmup + i, timed=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kern_us = runner.kernel_times_us()
    # HBM bytes per launch of the fused kernel from the rocprofv3 PMC passes of this same command (collected
    # with tools/profile_round.sh - counters cannot be read from inside the process); kernel workload only
    traffic = None
    if workload == "kernel":
        try:
            prof = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*", "kernel_workload_pmc_traffic_v*.json")))
            if prof:
                t = json.load(open(prof[-1]))
                traffic = t["hbm_read_bytes_per_launch_corrected"] + t["hbm_write_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
    if rank == 0:
        total_particles = runner.particles_per_step * world * args.steps
        ach = runner.kernel_bytes / (np.mean(kern_us) * 1e-6) / 1e9
        out = {
            "metric": "particles/sec + logprob-kernel HBM GB/s (% of 8 TB/s), 1024 particles gpt2",
            "value": total_particles / dt,
            "unit": "particles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": runner.config(),
            "roofline": {
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "glb::chunk_stats_kernel + glb::finish_kernel (fused log-softmax + mask + logsumexp + sample: "
                          "chunked streaming reduction, then lse / logZ / draw per particle); duration = both launches",
                "bytes_per_launch": runner.kernel_bytes,
                "us_per_launch_mean": float(np.mean(kern_us)),
                "us_per_launch_median": float(np.median(kern_us)),
                "launches_timed": len(kern_us),
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(workload, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


class KernelWorkload:
    """Fused kernel only, [1024, 50257] fp32 logits, two shared {0,-inf} masks, in-kernel Philox."""

    particles_per_step = N_PARTICLES

    def __init__(self, eng, dev, rank, world, dist, nbuf=4):
        self.eng, self.dev, self.rank, self.world, self.dist = eng, dev, rank, world, dist
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        B, V = N_PARTICLES, V_GPT2
        self.bufs = [torch.randn((B, V), device=dev, generator=g) * 3.0 for _ in range(nbuf)]
        maskf = torch.where(torch.rand((2, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
        maskf[:, 0] = 0.0
        self.bits, _ = eng.mask_to_bits(maskf)
        self.masks = eng.prepare_masks(self.bits, V, torch.float32)  # built once, like the README's two masks
        self.mask_id = (torch.arange(B, device=dev) % 2).to(torch.int32)
        self.out = (torch.empty(B, device=dev), torch.empty(B, device=dev),
                    torch.empty(B, dtype=torch.int32, device=dev))
        self.lw = torch.zeros(B, device=dev)
        self.gathered = torch.empty(B * world, device=dev) if world > 1 else None
        self.kernel_bytes = algorithmic_bytes(B, V, 4, 2, (V + 31) // 32)
        self.events = []

    def step(self, i, timed):
        x = self.bufs[i % len(self.bufs)]
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.eng.step(x, mask=self.masks, row_mask_id=self.mask_id, rng_mode=1, seed=1234, offset=i,
                      particle_base=self.rank * N_PARTICLES, out=self.out)
        if timed:
            e1.record()
            self.events.append((e0, e1))
        self.lw += self.out[0]
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.gathered, self.lw)
            self.eng.normalize_weights(self.gathered)

    def kernel_times_us(self):
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self.events])

    def config(self):
        return {"workload": "fused kernel only: 1024 particles x gpt2 vocab 50257, fp32 logits [1024,50257] ld=V, "
                            "2 shared bit masks, Philox draw, 4 rotating logits buffers",
                "particles_per_gpu": N_PARTICLES, "vocab": V_GPT2, "rng": "philox"}


if __name__ == "__main__":
    main()
