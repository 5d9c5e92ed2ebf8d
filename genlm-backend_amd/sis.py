"""Sequential importance sampling over the MI355X hot path.

Two drivers of the same per-step pipeline (reference: README.md:46-115):

* `autobatched_sis` / `Particle` — the README example restated on `AsyncAmdLM.next_token_step`: one
  coroutine per particle, requests autobatched by the backend's queue.  Drop-in shape, Python-bound.
* `DeviceSIS` — the same algorithm with the particle population resident on the GPU (contexts,
  lengths, weights are device tensors) and O(1) Python work per step: context dedup
  (glb_group_contexts) -> optional cached-prefix match (glb_match_prefixes) -> ragged-to-padded gather
  (glb_gather_padded / glb_gather_kv_padded) -> PyTorch-ROCm transformer body -> lm_head on the last
  position only -> fused log-softmax + mask + logsumexp + draw (glb_logprob_mask_sample) ->
  bookkeeping (glb_particles_advance).  With `dist` set, particles are sharded across ranks and the
  log-weights are all-gathered over RCCL after every step (README.md:108-110 needs all of them).
"""
import asyncio

import os

import numpy as np
import torch

from ._lib import MASK_BITS
from .cache import KVPrefix

RNG_PHILOX, RNG_NOISE = 1, 2


# ------------------------------------------------------------------------------------------------
# mask builders (README.md:57-70)
# ------------------------------------------------------------------------------------------------
def make_masks(llm, max_token_length):
    """The two shared log-masks of the README example, registered with the backend as mask 0 and 1:
       mask 0 (`valid_ids.log()`): 0 for EOS and for tokens of at most `max_token_length` bytes, -inf otherwise;
       mask 1 (`eos_one_hot.log()`): 0 for EOS only.
    `mask_selector(context)` of `make_masking_function` picks 1 once `len(context) >= max_tokens`."""
    eos_id = llm.tokenizer.eos_token_id
    V = len(llm.byte_vocab)
    valid = torch.tensor([tid == eos_id or len(tok) <= max_token_length for tid, tok in enumerate(llm.byte_vocab)],
                         dtype=torch.float).log()
    eos_one_hot = torch.nn.functional.one_hot(torch.tensor(eos_id), V).log()
    masks = torch.stack([valid, eos_one_hot.to(torch.float)])
    llm.register_masks(masks)
    return masks


def make_masking_function(llm, max_token_length, max_tokens):
    """README.md:57-70 shaped for the fused API: returns the mask *index* for a context."""
    make_masks(llm, max_token_length)
    return lambda context: 1 if len(context) >= max_tokens else 0


# ------------------------------------------------------------------------------------------------
# README-shaped driver (asyncio, one coroutine per particle)
# ------------------------------------------------------------------------------------------------
class Particle:
    """README.md:72-91 with lines 82-87 replaced by one fused, autobatched call."""

    def __init__(self, llm, mask_selector, prompt_ids, eos_id):
        self.context = []
        self.prompt_ids = prompt_ids
        self.log_weight = 0.0
        self.active = True
        self.llm = llm
        self.mask_selector = mask_selector
        self.eos_id = eos_id

    async def extend(self):
        logZ, token = await self.llm.next_token_step(self.prompt_ids + self.context,
                                                     mask_id=self.mask_selector(self.context))
        self.log_weight += logZ
        if token == self.eos_id or token < 0:
            self.active = False
        else:
            self.context.append(token)


async def autobatched_sis(n_particles, llm, mask_selector, prompt_ids, eos_id, gather=None):
    """README.md:94-98.  gather: what runs a step's coroutines together - `asyncio.gather` as in the README (default), or
    `llm.gather` (AsyncAmdLM.gather: the same results without a Task per particle)."""
    gather = gather or asyncio.gather
    particles = [Particle(llm, mask_selector, prompt_ids, eos_id) for _ in range(n_particles)]
    while any(p.active for p in particles):
        await gather(*[p.extend() for p in particles if p.active])
    return particles


# ------------------------------------------------------------------------------------------------
# device-resident driver
# ------------------------------------------------------------------------------------------------
def _gather_all(dist, out, inp):
    """all_gather_into_tensor - through host memory when the group's backend cannot take device tensors (a gloo group
    whose ranks compute on a GPU: the one-GPU rehearsal of the multi-rank path; RCCL groups take the tensors as they are)."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp)


def _all_to_all(dist, out, inp, out_splits, in_splits):
    """all_to_all_single with uneven splits (rows of an int32 matrix) - through host memory for a gloo group whose ranks
    compute on a GPU, like _gather_all."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)


def _all_to_all_any(dist, out, inp, out_splits, in_splits):
    """_all_to_all for tensors of any shape and element type (KV rows): split along the first dimension."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)


def _reduce_all(dist, t, op):
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=op)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op)


class CollectiveClock:
    """What a run's collectives cost, for bench lines that have to explain a scaling curve: on an RCCL group the device
    time between two events recorded on the stream around the call (it includes waiting for the slowest rank to arrive -
    that IS the cost of a collective in a step), on a gloo group (the one-GPU rehearsal, CPU tests) the host time of the
    call, which is synchronous there.  Off unless `start()` was called: nothing is recorded, nothing is timed."""

    def __init__(self):
        self.on = False
        self.events, self.host_s, self.calls, self.bytes = [], 0.0, 0, 0

    def start(self):
        self.on = True
        self.events, self.host_s, self.calls, self.bytes = [], 0.0, 0, 0

    def run(self, fn, tensor, nbytes):
        if not self.on:
            return fn()
        self.calls += 1
        self.bytes += int(nbytes)
        if tensor.is_cuda and self._device_timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            self.events.append((e0, e1))
            return out
        import time

        t0 = time.perf_counter()
        out = fn()
        self.host_s += time.perf_counter() - t0
        return out

    _device_timed = True

    def total_us(self):
        """(the caller has synchronised the device)"""
        return self.host_s * 1e6 + sum(a.elapsed_time(b) for a, b in self.events) * 1e3


class DeviceSIS:
    """N particles over one prompt (or one prompt per particle, any lengths), masks[0] while fewer than `max_tokens`
    tokens were generated and masks[1] afterwards (README.md:57-70's masking function).

    use_prefix_kv    the distinct prompts' KV is computed once (hf.py:155-164 `cache_kv`) and every step feeds only the
                     generated tokens (the reference's algorithm with its prefix cache; BASELINE config 3).
    use_particle_kv  beyond the reference: every particle owns a row of preallocated KV slabs (kv.SlabKV); step 0
                     encodes the distinct prompts and fans their KV out, later steps feed ONE token per particle.
    resample_ess     None: never resample (the reference's README loop).  Otherwise, after a step whose effective
                     sample size is below `resample_ess * N_total`, the population is resampled systematically from
                     the all-gathered log-weights; every rank computes the same ancestors (glb_resample_systematic),
                     contexts and KV rows follow their ancestors, weights are reset to the population mean.
    share_kv         (with use_particle_kv) particles with the same context share ONE KV row and one forward row: a block
                     table `row_of`, copy-on-append when a shared row's particles draw different tokens, resampling
                     re-points rows instead of copying them, contexts without a row (step 0, ancestors from another
                     rank, a row budget that is spent) are encoded from their tokens (kv.SharedSlabKV).  False: every
                     particle owns row i (kv.SlabKV), no dedup - the decode loop of DeviceSampler, whose sequences
                     never coincide.
    kv_rows          row budget of the shared store (default: one per particle, which always suffices); contexts that
                     find no free row are served without their KV being kept.
    kv_in_place      with shared KV rows: the fraction of live slab rows from which a forward runs on the slab in place
                     (free rows ride along) instead of gathering the live rows' prefixes into batch order
    kv_graph         the in-place forward is replayed from a hipGraph after its second call (kv.SlabForward)
    particle_masks   int32 bit rows [n_particles + 1, ceil(V / 32)] on the device: particle i's OWN mask (row i; what a
                     grammar gives: README.md:57-70 generalised) while it generates, row n_particles once
                     `max_tokens` are out.  Prepared for the kernels once; `update_particle_masks(rows, bit_rows)` changes
                     some of them and only those are prepared again; every particle is its own reduction unit.
    force_collectives  run the collectives of the multi-rank path (all-gather of log-weights / token matrices, the
                     set-up reductions) through `dist` even when world == 1: a one-rank "nccl" group exercises the RCCL
                     code of an 8-GPU run on a single GPU.
    """

    def __init__(self, llm, n_particles, prompt_ids, max_tokens, eos_id, seed=0, rng="philox", rank=0, world=1,
                 dist=None, use_prefix_kv=False, use_particle_kv=False, resample_ess=None, force_collectives=False,
                 share_kv=True, kv_rows=None, kv_in_place=0.75, kv_graph=True, particle_masks=None, migrate_kv=True):
        self.llm, self.eng, self.dev = llm, llm.engine, llm.device
        self.N, self.max_tokens, self.eos_id = n_particles, max_tokens, eos_id
        self.rank, self.world, self.dist = rank, world, dist
        self.collective = world > 1 or (bool(force_collectives) and dist is not None)
        self.coll_clock = CollectiveClock()  # (bench.py starts it: collective_us_per_step of the multi-rank lines)
        if dist is not None:
            self.coll_clock._device_timed = dist.get_backend() != "gloo"
        self.rows_moved_total = 0
        self.migrate_kv = bool(migrate_kv)  # private KV slabs: a particle resampled from another rank brings its KV rows along
        self.kv_rows_moved = 0
        self.seed = seed
        self.rng_mode = RNG_PHILOX if rng == "philox" else RNG_NOISE
        self.noise_src = None  # parity draws: torch's CPU generator on the device (engine.DeviceRng), made at the first step
        # GLB_RNG_AHEAD=1: the next step's noise rows generated on a low-priority side stream under the next forward
        # (DeviceRng.prefetch).  Off by default: measured on one box, 1024 x gpt2-small 4.05 / 4.04 ms a step with it against
        # 4.12 / 3.98 without, 512 x Llama-3.2-1B 4.37 / 4.44 against 4.55 / 4.50 - the generation is issue-bound work (GF(2)
        # convolutions, the recurrence, double-precision logarithms), not something idle cycles of the forward absorb, and it
        # costs another [N, V] buffer.
        self.noise_ahead = os.environ.get("GLB_RNG_AHEAD", "0") == "1"
        prompts = prompt_ids if isinstance(prompt_ids[0], (list, tuple)) else [prompt_ids] * n_particles
        assert len(prompts) == n_particles
        self._prompt_len0 = torch.tensor([len(p) for p in prompts], dtype=torch.int32, device=self.dev)
        self.max_prompt = max(len(p) for p in prompts)
        if self.collective:  # one token-matrix width over all ranks: rows travel between ranks when resampling
            mp_ = torch.tensor([self.max_prompt], dtype=torch.int64, device=self.dev)
            _reduce_all(dist, mp_, dist.ReduceOp.MAX)
            self.max_prompt = int(mp_.item())
        # the README mask is a function of the number of generated tokens; with prompts of ONE length (over all
        # ranks, if particles can migrate) that makes it a function of the context, i.e. of the logits row
        lens = {len(p) for p in prompts}
        if self.collective and resample_ess is not None:
            mm = torch.tensor([min(lens), -max(lens)], dtype=torch.int64, device=self.dev)
            _reduce_all(dist, mm, dist.ReduceOp.MIN)
            lens = {int(mm[0]), int(-mm[1])}
        self._mask_by_row = len(lens) == 1
        self._rep = None
        self.cap = self.max_prompt + max_tokens + 1
        ctx = np.zeros((n_particles, self.cap), np.int32)
        for i, p in enumerate(prompts):
            ctx[i, :len(p)] = p
        self._ctx0 = torch.from_numpy(ctx).to(self.dev)
        self.starts = (torch.arange(n_particles, device=self.dev, dtype=torch.int64) * self.cap)
        # cached prompt prefixes (hf.py:155-164): one KV slab set per distinct prompt
        self.prefixes = None
        if use_prefix_kv:
            distinct = {tuple(p) for p in prompts}
            if self.collective and resample_ess is not None:
                # particles migrate between ranks when the population is resampled, and every context must find its
                # prompt in the local table (the forward is fed the generated tokens only): cache the prompts of ALL ranks
                width = self.max_prompt
                all_p = torch.empty((world * n_particles, width), dtype=torch.int32, device=self.dev)
                all_l = torch.empty(world * n_particles, dtype=torch.int32, device=self.dev)
                _gather_all(dist, all_p.view(-1), self._ctx0[:, :width].contiguous().view(-1))
                _gather_all(dist, all_l, self._prompt_len0)
                all_p, all_l = all_p.cpu().numpy(), all_l.cpu().numpy()
                distinct = {tuple(int(t) for t in all_p[i, :all_l[i]]) for i in range(len(all_l))}
            self._build_prefixes(sorted(distinct))
        self.particle_kv = bool(use_particle_kv)
        self.share_kv = bool(share_kv) and self.particle_kv
        self.kv_rows = int(kv_rows) if kv_rows is not None else n_particles
        self.kv_stats = dict(forward_rows=0, encoded_rows=0, copied_rows=0, unkept_rows=0, steps=0, in_place_steps=0)
        # a forward runs on the KV slab rows where they lie when at least this fraction of them is live (None: always
        # gather the live rows into batch order)
        self.kv_in_place = kv_in_place
        self.kv_graph = kv_graph
        self._slab_fwd = None
        if self.particle_kv:
            assert not use_prefix_kv
        self.particle_masks = particle_masks
        self._pm_prepared, self._pm_dirty, self._pm_seen, self._pm_moved = None, None, None, False
        self.pm_raw_above = None  # fraction of changed rows above which a step hands the bit rows over raw (None: by dtype)
        self.pm_raw_steps = 0
        self.rows_moved = 0  # particles that changed ranks in the last resampling step (over all ranks)
        self.resample_ess = resample_ess
        self.n_resamples = 0
        self.sync_every = 1  # per-particle-KV steps: how often the active counts are read back (one small D2H copy)
        self.pkv = None
        self.reset()

    @torch.no_grad()
    def _build_prefixes(self, distinct):
        llm, dev = self.llm, self.dev
        kvs = []
        for p in distinct:
            out = llm._body(input_ids=torch.tensor([list(p)], device=dev), use_cache=True)
            kvs.append(KVPrefix.from_hf_cache(out.past_key_values))
        lens = np.array([len(p) for p in distinct], np.int32)
        starts = np.zeros(len(distinct), np.int64)
        if len(distinct) > 1:
            starts[1:] = np.cumsum(lens[:-1])
        flat = np.concatenate([np.array(p, np.int32) for p in distinct])
        ptrs = [[torch.tensor([kv.layers[l][j].data_ptr() for kv in kvs], dtype=torch.int64, device=dev)
                 for j in range(2)] for l in range(len(kvs[0].layers))]
        self.prefixes = dict(kvs=kvs, tokens=torch.from_numpy(flat).to(dev), starts=torch.from_numpy(starts).to(dev),
                             lengths=torch.from_numpy(lens).to(dev), ptrs=ptrs, p_max=int(lens.max()))

    def reset(self):
        if self.noise_src is not None:  # a run starts its seeded noise stream over
            self.noise_src.reset()
        self._pm_prepared, self._pm_dirty, self._pm_seen, self._pm_moved = None, None, None, False  # (a new run may come with new masks)
        self.contexts = self._ctx0.clone()
        self.prompt_len = self._prompt_len0.clone()
        self.lengths = self.prompt_len.clone()
        self.active = torch.ones(self.N, dtype=torch.int32, device=self.dev)
        self.log_weights = torch.zeros(self.N, dtype=torch.float32, device=self.dev)
        self.t = 0
        self.max_len_now = self.max_prompt
        self.last_stats = None
        self.kernel_events = []
        self.outer_events = []
        self._event_pool = []
        # (the slabs stay - and the hipGraphs captured over them; shared rows start from an empty block table, private rows
        # are refilled by step 0's encoding)
        self._head_cache = None
        self._noise_buf = getattr(self, "_noise_buf", None)
        self._fwd_tokens = None  # tokens the transformer body was fed in the running step, when that is not n_unique x l_max
        self._noise_groups = None  # parity draws: the dedup grouping the noise rows are dealt by (set per step)
        self._kv_stale = None  # bool [N]: rows whose KV has to be rebuilt from the context (ancestor on another rank)
        self._row_of_d = torch.full((self.N,), -1, dtype=torch.int32, device=self.dev)  # shared KV: particle -> slab row (-1: none)
        # active particles over all ranks as of the last exchange (device scalar; read with the step's one D2H copy)
        self._global_active = torch.tensor(self.N * self.world, dtype=torch.int32, device=self.dev)
        self.all_weights = None
        self._all_stamp = None  # parity draws over several ranks: which (step, resampling) the gathered hashes belong to
        self._rehash()

    def _rehash(self):
        """Context hashes for the per-step dedup: a context's hash extends token by token (glb_particles_advance keeps
        it up to date), so the grouping never reads the contexts again except to confirm duplicates.  A finished
        particle dedups as its 1-token stub: that hash is kept beside it."""
        ctx_flat = self.contexts.view(-1)
        self.hashes = self.eng.hash_contexts(ctx_flat, self.starts, self.lengths)
        self._hash_stub = self.eng.hash_contexts(ctx_flat, self.starts, torch.ones_like(self.lengths))

    # -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _encode_into_slabs(self, rows):
        """(Re)build the KV rows `rows` (int64 device indices, or None = all) from their contexts: the distinct contexts
        are encoded once (dedup as in hf.py:214-220) and their KV fanned out with one gather launch."""
        eng, llm, dev, N = self.eng, self.llm, self.dev, self.N
        ctx_flat = self.contexts.view(-1)
        # a row's KV holds every token of its context but the newest one (that one is fed by the next forward); at
        # t == 0 the whole prompt is encoded and its last position also yields the step's logits
        kv_len = self.lengths if self.t == 0 else (self.lengths - 1).clamp_min(1)
        want = torch.ones(N, dtype=torch.bool, device=dev) if rows is None else \
            torch.zeros(N, dtype=torch.bool, device=dev).index_fill_(0, rows, True)
        lens_eff = torch.where(want, kv_len, torch.ones_like(kv_len))  # rows not rebuilt collapse to a 1-token stub
        group_of, rep, ng = eng.group_contexts(ctx_flat, self.starts, lens_eff)
        U = int(ng.item())
        l_max = int(lens_eff.max().item())
        ids, am, pos, _ = eng.gather_padded(ctx_flat, self.starts, lens_eff, rep, U, None, 0, 0, l_max)
        out = llm._body(input_ids=ids, attention_mask=am, position_ids=pos, use_cache=True)
        src = [(ly.keys.contiguous(), ly.values.contiguous()) for ly in out.past_key_values.layers]
        if self.pkv is None:
            from .kv import SlabKV

            self.pkv = SlabKV(eng, N, self.cap, len(src))
        src_row = torch.where(want, group_of, torch.full_like(group_of, -1))
        self.pkv.fill_rows(src, src_row, lens_eff)
        return out, group_of, rep, U

    @torch.no_grad()
    def _step_particle_kv(self, time_kernel):
        """Steps t >= 1 with per-particle KV: one new token per particle (logits row i = particle i), ragged lengths."""
        eng, llm, dev, N = self.eng, self.llm, self.dev, self.N
        if self.t % self.sync_every == 0 or self._head_cache is None:
            head = torch.stack([self.active.sum().to(torch.int32), self._global_active, eng.error_word()[0]]).cpu()  # the step's one D2H copy
            eng.raise_if_failed(int(head[2]))  # (the word of every call since the last copy rides along)
            self._head_cache = (int(head[0]), int(head[1]))
        n_active, n_global = self._head_cache
        if self._kv_stale is not None:
            self._encode_into_slabs(torch.nonzero(self._kv_stale).flatten())
            self._kv_stale = None
        rows = torch.arange(N, device=dev)
        pos = (self.lengths - 1).clamp_min(0)  # tokens already in the row's KV = index of the newest token
        newest = self.contexts[rows, pos.long()]
        ids = torch.where(self.active > 0, newest, torch.zeros_like(newest)).view(N, 1).long()
        if self._slab_fwd is None or self._slab_fwd.pkv is not self.pkv:
            from .kv import SlabForward

            self._slab_fwd = SlabForward(self.pkv, llm._body, graph=self.kv_graph, owner=llm)
        logits = llm._lm_head(self._slab_fwd(ids, pos))  # [N, V]
        self._noise_groups = None
        if self.rng_mode == RNG_NOISE:  # parity draws follow the reference's resolution order: by dedup group
            lengths_eff = torch.where(self.active > 0, self.lengths, torch.ones_like(self.lengths))
            self._noise_groups, _, _ = eng.group_contexts(self.contexts.view(-1), self.starts, lengths_eff)
        return self._finish_step(logits, None, N, n_active, n_global, time_kernel, l_max=1)

    @torch.no_grad()
    def _step_shared_kv(self, time_kernel):
        """One step with shared KV rows.  The distinct contexts (hf.py:214-220 dedup) are the forward's rows; each is
        (A) a context whose prefix sits in a slab row - one new token is fed, its K / V appended in place; when several
        new contexts grew out of one row, the first keeps it and the others get a copy of the prefix in a free row; or
        (B) a context without a row (step 0, an ancestor from another rank, a spent row budget) - encoded from its tokens
        like the reference does every step, its KV kept if a row is free.  The block table is decided on the device
        (glb_kv_plan, one launch); the host reads nine words - how many rows of which kind - and launches the forwards."""
        eng, llm, dev, N, R = self.eng, self.llm, self.dev, self.N, self.kv_rows
        ctx_flat = self.contexts.view(-1)
        lengths_eff = torch.where(self.active > 0, self.lengths, torch.ones_like(self.lengths))
        hashes_eff = torch.where(self.active > 0, self.hashes, self._hash_stub)
        group_of, rep, ng = eng.group_contexts(ctx_flat, self.starts, lengths_eff, hashes=hashes_eff)
        plan = eng.kv_plan(group_of, rep, ng, self._row_of_d, lengths_eff, R, self.cap, by_context=True)
        head = torch.cat([plan["head"][:6], torch.stack([self.active.sum().to(torch.int32), self._global_active,
                                                          eng.error_word()[0]])]).cpu().tolist()  # the step's one D2H copy
        U, nA, nB, n_copied, n_unkept, l_max_b, n_active, n_global = head[:8]
        eng.raise_if_failed(head[8])
        self._row_of_d = plan["row_of_context"]
        st = self.kv_stats
        st["forward_rows"] += U
        st["encoded_rows"] += nB
        st["copied_rows"] += n_copied
        st["unkept_rows"] += n_unkept
        st["steps"] += 1
        logits_parts = []
        fed_a = 0  # tokens the one-token forward is fed: its live rows, or every slab row when it runs in place
        if nA:
            if n_copied:
                self.pkv.copy_rows(plan["copy_src"], plan["copy_len"])
            if self.kv_in_place is not None and nA >= self.kv_in_place * R:
                # most rows are live: the forward runs on the slab rows where they lie (rows outside it ride along with a
                # dummy token at position 0) instead of gathering the live rows' prefixes into batch order
                pos_d = plan["pos_of_row"]
                ids = self.contexts[plan["ctx_of_row"].clamp_min(0).long(), pos_d.long()].view(-1, 1).long()
                if self._slab_fwd is None or self._slab_fwd.pkv is not self.pkv:
                    from .kv import SlabForward

                    self._slab_fwd = SlabForward(self.pkv, llm._body, graph=self.kv_graph, owner=llm)
                hidden = self._slab_fwd(ids, pos_d)
                logits_parts.append(llm._lm_head(hidden.index_select(0, plan["rows_a"][:nA].long())))
                st["in_place_steps"] += 1
                fed_a = R
            else:
                pos_a = plan["pos_a"][:nA].contiguous()
                ids = self.contexts[plan["ctx_a"][:nA].long(), pos_a.long()].view(-1, 1).long()
                self.pkv.set_forward(plan["rows_a"][:nA].contiguous(), pos_a)
                out = llm._body(input_ids=ids, position_ids=pos_a.view(-1, 1).long(),
                                attention_mask=self.pkv.attention_mask(pos_a), past_key_values=self.pkv, use_cache=True)
                logits_parts.append(llm._lm_head(out.last_hidden_state[:, 0]))
                fed_a = nA
        if nB:
            sel = plan["ctx_b"][:nB].contiguous()
            ids, am, pos, last = eng.gather_padded(ctx_flat, self.starts, lengths_eff, sel, nB, None, 0, 0, l_max_b)
            out = llm._body(input_ids=ids, attention_mask=am, position_ids=pos, use_cache=True)
            h_last = out.last_hidden_state[torch.arange(nB, device=dev), last.long()]
            logits_parts.append(llm._lm_head(h_last))
            if nB > n_unkept:  # rows that keep the KV of what was just encoded
                src = [(ly.keys.contiguous(), ly.values.contiguous()) for ly in out.past_key_values.layers]
                if self.pkv is None:
                    from .kv import SharedSlabKV

                    self.pkv = SharedSlabKV(eng, R, self.cap, len(src))
                rows_b = plan["rows_b"][:nB].long()
                slot = torch.where(rows_b >= 0, rows_b, torch.full_like(rows_b, R))  # (rows nobody keeps: a slot past the end)
                src_full = torch.full((R + 1,), -1, dtype=torch.int32, device=dev)
                len_full = torch.zeros(R + 1, dtype=torch.int32, device=dev)
                src_full[slot] = torch.arange(nB, dtype=torch.int32, device=dev)
                len_full[slot] = lengths_eff[sel.long()]
                src_full[R] = -1
                self.pkv.fill_rows(src, src_full[:R].contiguous(), len_full[:R].contiguous())
        self._fwd_tokens = fed_a + nB * l_max_b
        logits = logits_parts[0] if len(logits_parts) == 1 else torch.cat(logits_parts)
        self._rep = torch.cat([plan["ctx_a"][:nA], plan["ctx_b"][:nB]])  # the context behind every logits row
        self._noise_groups = group_of  # parity draws follow the reference's resolution order: by dedup group
        row_of = plan["logits_row"][group_of.long()]
        return self._finish_step(logits, row_of, U, n_active, n_global, time_kernel, l_max=1)

    def _finish_step(self, logits, group_of, U, n_active, n_global, time_kernel, l_max):
        eng, llm, N = self.eng, self.llm, self.N
        V = logits.shape[-1]
        mask_id = ((self.lengths - self.prompt_len) >= self.max_tokens).to(torch.int32)
        kw = llm.step_masks(logits.dtype) if self.particle_masks is None else {}
        if self.particle_masks is not None:  # one mask per particle: per-particle ids, no dedup of the math
            # Two ways to hand the bit rows over.  RAW: the fused launch's stats waves read the caller's rows themselves
            # (kMaskRaw: + 2 us a step on float32 rows, + 5 us on 16-bit ones, nothing to prepare).  PREPARED: the rows
            # transposed into the kernels' layout (glb_mask_prepare: 11 us for 1025 rows of 50257), afterwards only the rows
            # `update_particle_masks` named again (glb_mask_prepare_rows).  A grammar that moves every particle's mask every
            # token is served raw; masks that stand still, or of which a few move a step, prepared.  A step goes raw
            #  * when more than `pm_raw_above` of the rows changed since the prepared form was last brought up to date and
            #    some did since the last step (their names are kept: a step before which nothing moved brings it up to date);
            #  * when the tensor is in a state this object has not seen - rebound, or written in place by anyone but
            #    `update_particle_masks` (its version counter moves); seen twice in a row, that state is prepared.
            # The parity draw is two launches (no raw form): always prepared.
            own = torch.arange(N, dtype=torch.int32, device=self.dev)
            pm = self.particle_masks
            ids = torch.where(mask_id > 0, torch.full_like(own, N), own)
            state = (pm.data_ptr(), pm._version, logits.dtype)
            frac = self.pm_raw_above if self.pm_raw_above is not None else (0.25 if logits.dtype == torch.float32 else 0.5)
            raw = False
            if self._pm_prepared is None or self._pm_prepared[1] != state:
                if self.rng_mode != RNG_NOISE and frac < 1.0 and self._pm_seen != state:
                    raw, self._pm_prepared, self._pm_dirty = True, None, None
                else:
                    self._pm_prepared, self._pm_dirty = (eng.prepare_masks(pm, V, logits.dtype), state), None
            elif self._pm_dirty is not None:
                if self.rng_mode != RNG_NOISE and self._pm_moved and self._pm_dirty.numel() > frac * N:
                    raw = True  # (the prepared form stays behind by the rows named in _pm_dirty, until the masks stand still)
                else:
                    eng.update_prepared_masks(self._pm_prepared[0], pm, self._pm_dirty)
                    self._pm_dirty = None
            self._pm_seen, self._pm_moved = state, False
            self.pm_raw_steps += int(raw)
            kw = dict(mask_kind=MASK_BITS, mask=pm, mask_id=ids) if raw else dict(mask=self._pm_prepared[0], mask_id=ids)
        elif kw:
            # The mask depends on the number of generated tokens only; with prompts of one length that makes it a
            # function of the context, so identical contexts (one logits row) share it: ids go per ROW and a shared
            # row is reduced once (hf.py:214-220 dedup carried through the particle math).
            if self._mask_by_row and group_of is not None:
                kw["row_mask_id"] = mask_id[self._rep[:logits.shape[0]].long()].contiguous()
            elif self._mask_by_row:
                kw["row_mask_id"] = mask_id
            else:
                kw["mask_id"] = mask_id
        if self.rng_mode == RNG_NOISE:
            kw["noise"] = self._parity_noise(self._noise_groups if self._noise_groups is not None else group_of, V)
        if time_kernel:
            # two clocks on the fused call: HIP events the launch itself carries as its start / stop stamps (the launch
            # duration, as rocprofv3 reports it), and a pair recorded around the call on the stream (adds the two marker
            # packets and whatever the stream does between them)
            if len(self._event_pool) < 2:
                self._event_pool = eng.timing_events(64)
            inner, (e0, e1) = self._event_pool.pop(), self._event_pool.pop()
            e0.record()
            kw["timing_events"] = inner
        logZ, _, tok = eng.step(logits, vocab=V, row_of=group_of, rng_mode=self.rng_mode, seed=self.seed,
                                offset=self.t, particle_base=self.rank * N, want_lse=False, **kw)
        if time_kernel:
            e1.record()
            self.kernel_events.append(inner)
            self.outer_events.append((e0, e1))
        if self.rng_mode == RNG_NOISE:  # the next step's rows set out on a side stream, under the next forward
            self.noise_src.prefetch()
        eng.particles_advance(self.contexts, self.lengths, self.active, self.log_weights, logZ, tok, self.eos_id,
                              self.cap, hashes=self.hashes)
        self.t += 1
        self.max_len_now = min(self.max_len_now + 1, self.cap)
        ft = self._fwd_tokens if self._fwd_tokens is not None else U * l_max
        self._fwd_tokens = None
        self.last_stats = dict(n_unique=U, n_active=n_active, l_max=l_max, n_rows=U, fwd_tokens=ft, head_rows=logits.shape[0])
        self._exchange()
        if self.resample_ess is not None:
            self._maybe_resample()
        return U, n_global

    def update_particle_masks(self, rows, bit_rows):
        """Particles `rows` (int32 device tensor) get new masks `bit_rows` (int32 [len(rows), ceil(V / 32)]): only these are
        brought into the kernels' layout again before the next step."""
        pm = self.particle_masks
        known = self._pm_prepared is not None and self._pm_prepared[1][:2] == (pm.data_ptr(), pm._version)
        pm[rows.long()] = bit_rows
        if known:  # the prepared form follows this write row by row; any other write makes the next step prepare everything
            self._pm_prepared = (self._pm_prepared[0], (pm.data_ptr(), pm._version, self._pm_prepared[1][2]))
            self._pm_dirty = rows if self._pm_dirty is None else torch.unique(torch.cat([self._pm_dirty, rows]))
            self._pm_moved = True

    def _exchange(self):
        """All-gather of the per-shard log-weights and active counts (RCCL over xGMI when the backend is nccl): every
        rank then holds the population's weights (README.md:108-110 needs all of them) and knows whether anybody,
        anywhere, is still generating - the loop's termination test is collective."""
        count = self.active.sum().to(torch.float32).view(1)
        if self.collective:
            N = self.N
            parts = [self.log_weights, count]
            carry = self.rng_mode == RNG_NOISE and self.world > 1
            if carry:
                # parity draws of a sharded population: every rank deals the WHOLE population's rows of the one stream
                # (_parity_noise), which takes every particle's context hash (64 bits, carried as two float32 words: the
                # collective copies bytes) and whether it still draws - 12 bytes a particle beside the 4 of its weight
                parts += [self.hashes.view(torch.float32), self.active.to(torch.float32)]
            mine = torch.cat(parts)
            out = torch.empty((self.world, mine.numel()), dtype=torch.float32, device=self.dev)
            self.coll_clock.run(lambda: _gather_all(self.dist, out.view(-1), mine), mine, out.numel() * 4)
            self.all_weights = out[:, :N].reshape(-1)
            self._global_active = out[:, N].sum().to(torch.int32)
            if carry:
                self._all_hashes = out[:, N + 1:3 * N + 1].contiguous().view(torch.int64).reshape(-1)
                self._all_active = out[:, 3 * N + 1:].reshape(-1) > 0
                self._all_stamp = (self.t, self.n_resamples)
        else:
            self.all_weights = self.log_weights
            self._global_active = count[0].to(torch.int32)

    @torch.no_grad()
    def step(self, time_kernel=False):
        """One SIS step for every active particle.  Returns (n_unique, particles active over ALL ranks before it)."""
        eng, llm, dev, N = self.eng, self.llm, self.dev, self.N
        if self.share_kv:
            return self._step_shared_kv(time_kernel)
        if self.particle_kv and self.t > 0:
            return self._step_particle_kv(time_kernel)
        ctx_flat = self.contexts.view(-1)
        # finished particles still occupy a row: give them their 1-token stub so they dedup to one group
        lengths_eff = torch.where(self.active > 0, self.lengths, torch.ones_like(self.lengths))
        if self.particle_kv:  # step 0: encode the distinct prompts, keep their KV, fan it out to the particles
            head = torch.stack([self.active.sum().to(torch.int32), self._global_active, eng.error_word()[0]]).cpu()
            n_active, n_global = int(head[0]), int(head[1])
            eng.raise_if_failed(int(head[2]))
            out, group_of, rep, U = self._encode_into_slabs(None)
            self._rep = rep
            self._noise_groups = None
            last = (self.lengths[rep[:U].long()] - 1).long()
            h_last = out.last_hidden_state[torch.arange(U, device=dev), last]
            return self._finish_step(llm._lm_head(h_last), group_of, U, n_active, n_global, time_kernel, self.max_len_now)
        hashes_eff = torch.where(self.active > 0, self.hashes, self._hash_stub)
        group_of, rep, ng = eng.group_contexts(ctx_flat, self.starts, lengths_eff, hashes=hashes_eff)
        self._rep = rep
        head = torch.stack([ng[0], self.active.sum().to(torch.int32), self._global_active, eng.error_word()[0]]).cpu()  # the step's one D2H copy
        U, n_active, n_global = int(head[0]), int(head[1]), int(head[2])
        eng.raise_if_failed(int(head[3]))  # a fused call of an earlier step that did not complete
        base, p_max, cache = None, 0, None
        # at t == 0 every context *is* its prompt, so no cached prefix is a proper prefix yet (hf.py:334-342)
        use_kv = self.prefixes is not None and self.t > 0
        l_max = self.max_len_now
        if use_kv:
            P = self.prefixes
            pref, base = eng.match_prefixes(ctx_flat, self.starts, lengths_eff, P["tokens"], P["starts"], P["lengths"])
            p_max = P["p_max"]
            l_max = max(self.t, 1)  # every prompt is cached: only the generated tokens (<= t) are fed
        ids, am, pos, last = eng.gather_padded(ctx_flat, self.starts, lengths_eff, rep, U, base, 0, p_max, l_max)
        if use_kv:
            from transformers import DynamicCache

            P = self.prefixes
            kv0 = P["kvs"][0]
            pref_u = pref[rep[:U].long()].contiguous()
            data = [tuple(eng.gather_kv_padded(P["ptrs"][l][j], P["lengths"], pref_u, kv0.heads, kv0.head_dim, p_max,
                                               kv0.dtype) for j in range(2)) for l in range(len(kv0.layers))]
            cache = DynamicCache(ddp_cache_data=data)
        out = llm._body(input_ids=ids, attention_mask=am, position_ids=pos, past_key_values=cache,
                        use_cache=cache is not None)
        h_last = out.last_hidden_state[torch.arange(U, device=dev), last.long()]
        logits = llm._lm_head(h_last)  # [U, V]
        self._noise_groups = None
        return self._finish_step(logits, group_of, U, n_active, n_global, time_kernel, l_max)

    def _parity_noise(self, group_of, V):
        """Exp(1) rows in the order the reference's particles reach torch.multinomial: by dedup group (first appearance),
        duplicates contiguous, inactive particles draw nothing (hf.py:285-288, README.md:94-98).  The stream is torch's CPU
        generator, entered on the device at every particle's row at once (glb_mt19937_exponential_rows): the order is a
        stable sort of the group ids, the number of rows consumed a device scalar - nothing crosses the host."""
        N = self.N
        if self.noise_src is None:  # ONE stream for the whole population, whatever the number of ranks: the reference's
            self.noise_src = self.eng.noise_rng(self.seed, V, ahead=self.noise_ahead and self.world == 1)
        if self.collective and self.world > 1:
            return self._parity_noise_sharded(V)
        act = self.active > 0
        key = torch.where(act, group_of.to(torch.int64), torch.full((N,), 1 << 40, dtype=torch.int64, device=self.dev))
        order = torch.argsort(key, stable=True)
        rank = torch.empty(N, dtype=torch.int32, device=self.dev)
        rank[order] = torch.arange(N, dtype=torch.int32, device=self.dev)
        slot = torch.where(act, rank, torch.full_like(rank, -1))
        if self._noise_buf is None or self._noise_buf.shape != (N, V):
            self._noise_buf = torch.empty((N, V), dtype=torch.float32, device=self.dev)
        return self.noise_src.rows(N, row_slot=slot, n_draw=act.sum().to(torch.int32), max_draw=N, out=self._noise_buf)

    def _parity_noise_sharded(self, V):
        """The reference is ONE process: particle g of the whole population (rank * N + i) draws from the stream row it would
        draw from there - its place among the ACTIVE particles ordered by dedup group in first-appearance order over the
        global particle index, duplicates contiguous (hf.py:214-220,285-288).  Every rank computes that order for everybody
        (replicated, like resampling) from the all-gathered context hashes and active flags - equal hash = equal context,
        up to a 64-bit collision, which would only swap two rows of noise -, generates the rows of ITS particles at their
        global slots (the jump-ahead makes the offset free; windows nobody here reads are not made) and moves the stream on by
        the population's active count: the same tokens and weights as one process with all the particles."""
        N, dev = self.N, self.dev
        if getattr(self, "_all_stamp", None) != (self.t, self.n_resamples):  # (a run's first step: no exchange has carried them yet)
            mine = torch.cat([self.hashes.view(torch.float32), self.active.to(torch.float32)])
            out = torch.empty((self.world, 3 * N), dtype=torch.float32, device=dev)
            self.coll_clock.run(lambda: _gather_all(self.dist, out.view(-1), mine), mine, out.numel() * 4)
            self._all_hashes = out[:, :2 * N].contiguous().view(torch.int64).reshape(-1)
            self._all_active = out[:, 2 * N:].reshape(-1) > 0
            self._all_stamp = (self.t, self.n_resamples)
        h, act = self._all_hashes, self._all_active
        NT = h.numel()
        ar = torch.arange(NT, device=dev)
        sh, perm = torch.sort(h, stable=True)  # equal contexts side by side, each run in particle order
        start = torch.ones(NT, dtype=torch.bool, device=dev)
        start[1:] = sh[1:] != sh[:-1]
        first = perm[torch.cummax(torch.where(start, ar, torch.zeros_like(ar)), 0).values]  # a run's first = smallest particle index
        key = torch.empty(NT, dtype=torch.int64, device=dev)
        key[perm] = first
        key = torch.where(act, key, torch.full_like(key, 1 << 40))
        order = torch.argsort(key, stable=True)
        place = torch.empty(NT, dtype=torch.int32, device=dev)
        place[order] = torch.arange(NT, dtype=torch.int32, device=dev)
        slot = torch.where(act, place, torch.full_like(place, -1))[self.rank * N:(self.rank + 1) * N].contiguous()
        if self._noise_buf is None or self._noise_buf.shape != (N, V):
            self._noise_buf = torch.empty((N, V), dtype=torch.float32, device=dev)
        return self.noise_src.rows(N, row_slot=slot, n_draw=act.sum().to(torch.int32), max_draw=NT, out=self._noise_buf)

    # -------------------------------------------------------------------------------------------
    def gather_weights(self):
        """The population's log-weights as of the last step (all ranks hold the same vector)."""
        if self.all_weights is None:
            self._exchange()
        return self.all_weights

    def normalized_weights(self):
        return self.eng.normalize_weights(self.gather_weights())  # (probs, [logsumexp, ESS])  README.md:108-110

    @torch.no_grad()
    def _maybe_resample(self):
        n_total = self.N * self.world
        if self.resample_ess < 1.0:  # adaptive: needs the ESS on the host (one scalar; identical on every rank)
            _, stats = self.eng.normalize_weights(self.all_weights)
            if float(stats[1].item()) >= self.resample_ess * n_total:
                return
        self.resample()

    @torch.no_grad()
    def resample(self):
        """Systematic resampling of the whole population, replicated on every rank: identical gathered weights give
        identical ancestors (integer comb, one Philox draw keyed by (seed, step)); slot i of rank r takes ancestor
        anc[r*N + i].  A particle whose ancestor lives on this rank is one row gather; the rows that change ranks - and only
        those - travel in one all-to-all (every rank derives who sends what to whom from the replicated ancestors); KV rows
        follow a local ancestor and are rebuilt from the context when the ancestor lived elsewhere."""
        eng, N, dev = self.eng, self.N, self.dev
        n_total = N * self.world
        anc, lse = eng.resample_systematic(self.all_weights, self.seed ^ 0x5eed5a11, self.t)
        mine = anc[self.rank * N:(self.rank + 1) * N].contiguous()
        # a particle's state as one int32 row: its tokens, then (length, prompt length, active)
        state = torch.cat([self.contexts, torch.stack([self.lengths, self.prompt_len, self.active], dim=1)], dim=1).contiguous()
        local = mine - self.rank * N
        new_state = eng.gather_rows_i32(state, local.clamp(0, N - 1))  # (slots whose ancestor lives elsewhere: overwritten below)
        if self.collective and self.world > 1:
            # Only what moves travels: the ancestors are replicated, so every rank knows which of its rows the others take
            # and which rows it is sent - no request round, one all-to-all of exactly those rows (round 3 all-gathered the
            # whole N_total x cap token matrix on every resampling step).
            anc_h = anc.cpu().numpy()  # the resampling step's one D2H copy
            owner = anc_h // N
            r = self.rank
            send_idx = [anc_h[d * N:(d + 1) * N][owner[d * N:(d + 1) * N] == r] - r * N if d != r else anc_h[:0]
                        for d in range(self.world)]
            mine_owner = owner[r * N:(r + 1) * N]
            recv_cnt = [int((mine_owner == src).sum()) if src != r else 0 for src in range(self.world)]
            self.rows_moved = int((owner != np.repeat(np.arange(self.world), N)).sum())  # over all ranks (replicated value)
            self.rows_moved_total += self.rows_moved
            moved = None
            if self.rows_moved:
                take = np.concatenate(send_idx).astype(np.int64)
                take_d = torch.from_numpy(take).to(dev)
                send = state[take_d] if len(take) else state[:0]
                recv = torch.empty((sum(recv_cnt), state.shape[1]), dtype=torch.int32, device=dev)
                send_cnt = [len(x) for x in send_idx]
                send = send.contiguous()
                self.coll_clock.run(lambda: _all_to_all(self.dist, recv, send, recv_cnt, send_cnt), send,
                                    (send.numel() + recv.numel()) * 4)
                slots_d = None
                if sum(recv_cnt):
                    # rows arrive ordered by source rank, then by this rank's slot order - the order the senders used
                    slots = np.concatenate([np.nonzero(mine_owner == src)[0] for src in range(self.world) if src != r])
                    slots_d = torch.from_numpy(slots.astype(np.int64)).to(dev)
                    new_state[slots_d] = recv
                moved = (take_d, send_cnt, slots_d, recv_cnt)
        self.contexts = new_state[:, :self.cap].contiguous()
        self.lengths, self.prompt_len, self.active = (new_state[:, self.cap + k].contiguous() for k in range(3))
        # equal weights: log of the population's mean weight
        self.log_weights = (lse - float(np.log(n_total))).expand(N).contiguous()
        if self.share_kv:  # re-point: a particle takes its ancestor's row; ancestors of another rank leave it without one
            local = mine - self.rank * N
            ok = (local >= 0) & (local < N)
            self._row_of_d = torch.where(ok, self._row_of_d[local.clamp(0, N - 1).long()], torch.full_like(local, -1))
        elif self.particle_kv and self.pkv is not None:
            local = mine - self.rank * N
            is_local = (local >= 0) & (local < N)
            src = torch.where(is_local, local, torch.full_like(local, -1))
            kv_len = (self.lengths - 1).clamp_min(0)
            self.pkv.gather(src, kv_len)
            stale = ~is_local
            if self.migrate_kv and self.collective and self.world > 1:
                # the KV rows of the particles that changed ranks travel with them: ONE all-to-all of [rows, layers x {K, V},
                # heads, cap, head_dim] (the senders' rows as they were before the gather above: the second slab set)
                if self.rows_moved:
                    self._migrate_kv(*moved)
                self._kv_stale = None
            else:  # rebuilt from the context in the next step
                self._kv_stale = stale if bool(stale.any().item()) else None
        self._rehash()  # contexts moved between slots (and ranks): one launch, once per resampling step
        self.n_resamples += 1
        self._exchange()

    def _migrate_kv(self, take_d, send_cnt, slots_d, recv_cnt):
        """Private KV slabs after a resampling step: row take_d[k] of the OLD slabs (SlabKV.gather has just swapped them
        into its second set) goes to the rank that took it, the rows received land in slots_d of the new slabs.  One
        collective for all layers: fewer, larger messages are what xGMI's point-to-point links want."""
        pkv = self.pkv
        new = pkv._tensors(pkv.layers)
        old = pkv._alt
        n_send, n_recv = int(sum(send_cnt)), int(sum(recv_cnt))
        shape = tuple(new[0].shape[1:])
        for t in new:
            if tuple(t.shape[1:]) != shape or t.dtype != new[0].dtype:
                raise ValueError("KV layers of different shapes: rows cannot travel in one message")
        send = torch.stack([t[take_d] for t in old], dim=1).contiguous() if n_send else new[0].new_zeros((0, len(new)) + shape)
        recv = new[0].new_empty((n_recv, len(new)) + shape)
        self.coll_clock.run(lambda: _all_to_all_any(self.dist, recv, send, recv_cnt, send_cnt), send,
                            (send.numel() + recv.numel()) * send.element_size())
        if n_recv:
            for j, t in enumerate(new):
                t[slots_d] = recv[:, j]
        self.kv_rows_moved += n_recv

    @torch.no_grad()
    def run(self, max_steps=None):
        """README.md:94-98 `while any(p.active ...)`, with the test taken over ALL ranks: every rank runs the same number
        of steps, so the per-step collectives stay matched even when one shard finishes early."""
        steps = 0
        limit = max_steps if max_steps is not None else self.max_tokens + 1
        while steps < limit:
            _, n_global = self.step()
            steps += 1
            if n_global == 0:  # nobody was active before this step: it was a no-op everywhere
                break
        return steps

    def results(self):
        self.eng.check()  # the last steps' fused calls completed (glb_workspace_check; synchronises like the copies below)
        ctx = self.contexts.cpu().numpy()
        ln = self.lengths.cpu().numpy()
        pl = self.prompt_len.cpu().numpy()
        return [list(ctx[i, pl[i]:ln[i]]) for i in range(self.N)], self.log_weights.cpu().numpy()


# ------------------------------------------------------------------------------------------------
# multi-token sampling on the device (base.py:110-179)
# ------------------------------------------------------------------------------------------------
class DeviceSampler(DeviceSIS):
    """`AsyncLM.sample` / `batch_sample` as a device-resident loop: the sequences are the "particles" (per-sequence KV
    slabs, one new token per sequence per forward), the draw is the fused step with logit_scale = 1/temperature and no
    mask, stopping tokens are a set.  Seeded: every sequence draws from its own generator seeded alike (base.py:125-127),
    i.e. step t of every sequence races against the SAME Exp(1) row - one host-generated row per step, shared by pitch 0.
    Nothing is read back per token: the active count is fetched every `sync_every` steps."""

    def __init__(self, llm, prompts, max_tokens, eos_token_ids, temperature=1.0, seed=None, sync_every=4):
        super().__init__(llm, len(prompts), [list(p) for p in prompts], max_tokens, eos_id=-1,
                         seed=0 if seed is None else int(seed), rng="philox", use_particle_kv=True, share_kv=False,
                         kv_graph=max_tokens >= 64)  # (a capture costs a few dozen eager steps: long generations only)
        self.sync_every = max(1, int(sync_every))
        self.temperature = float(temperature)
        self.eos = torch.tensor(sorted(set(int(t) for t in eos_token_ids)), dtype=torch.int32, device=self.dev)
        self.seeded = seed is not None
        if not self.seeded:  # unseeded: in-kernel Philox keyed from torch's global generator
            self.seed = int(torch.randint(0, 2**62, (1,)).item())

    def _finish_step(self, logits, group_of, U, n_active, n_global, time_kernel, l_max):
        eng, N, dev = self.eng, self.N, self.dev
        V = logits.shape[-1]
        kw = {}
        mode = RNG_PHILOX
        if self.seeded:  # step t of every sequence races against the SAME row of torch's CPU stream (made on the device)
            mode = RNG_NOISE
            if self.noise_src is None:
                self.noise_src = eng.noise_rng(self.seed, V)
            kw["noise"] = self.noise_src.rows(1)
        _, _, tok = eng.step(logits, vocab=V, row_of=group_of, rng_mode=mode, seed=self.seed, offset=self.t,
                             logit_scale=1.0 / self.temperature, want_lse=False, **kw)
        act = (self.active > 0) & (tok != -2)  # token -2: a failed launch, never a result (raised at the next copy / results())
        stop = act & (torch.isin(tok, self.eos) | (tok < 0) | (self.lengths - self.prompt_len >= self.max_tokens))
        keep = act & ~stop
        rows = torch.arange(N, device=dev)
        at = self.lengths.long().clamp_max(self.cap - 1)
        self.contexts[rows, at] = torch.where(keep, tok, self.contexts[rows, at])
        self.lengths = self.lengths + keep.to(torch.int32)
        self.active = torch.where(tok == -2, self.active, (keep & (self.lengths - self.prompt_len < self.max_tokens)).to(torch.int32))
        self.t += 1
        self.max_len_now = min(self.max_len_now + 1, self.cap)
        self._exchange()
        return U, n_global

    def generate(self):
        steps = 0
        while steps < self.max_tokens:
            _, n_global = self.step()
            steps += 1
            if n_global == 0:
                break
        return self.results()[0]


# ------------------------------------------------------------------------------------------------
# bench.py workload
# ------------------------------------------------------------------------------------------------
class SisBenchWorkload:
    """N particles per GPU on a random-init model, prompt length 8, <= 10 new tokens, two shared {0,-inf} masks
    (README.md:57-70 shape), in-kernel Philox draws.  model "gpt2": GPT-2-small shape, fp32, 1024 particles
    (BASELINE.json config 2; with n_prompts / prefix_kv config 3).  model "llama-3.2-1b": Llama-3.2-1B shape, bf16,
    V = 128256, 512 particles (config 4: 4096 over 8 GPUs)."""

    def __init__(self, eng, dev, rank, world, dist, n_particles=1024, max_tokens=10, prefix_kv=False, particle_kv=False,
                 model="gpt2", n_prompts=1, resample=False, force_collectives=False, kv_in_place=0.75,
                 per_particle_masks=False, rng="philox", gemms="library"):
        from .llm import AsyncAmdLM

        if model == "gpt2":
            from transformers import GPT2Config

            cfg, dtype = GPT2Config(), torch.float32  # gpt2 small: 12 layers, d=768, 12 heads, vocab 50257
            self.model_name = "gpt2-small shape (random init, fp32)"
        elif model == "llama-3-8b":  # BASELINE config 5's model: 32 layers, d 4096, 32 heads / 8 KV heads of 128, MLP 14336, untied
            from transformers import LlamaConfig

            cfg = LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                              num_attention_heads=32, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096,
                              rope_theta=500000.0, rms_norm_eps=1e-5, tie_word_embeddings=False, bos_token_id=128000,
                              eos_token_id=128001)
            dtype = torch.bfloat16
            self.model_name = "Llama-3-8B shape (random init, bf16)"
        else:
            from transformers import LlamaConfig

            cfg = LlamaConfig(vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                              num_attention_heads=32, num_key_value_heads=8, head_dim=64, max_position_embeddings=4096,
                              rope_theta=500000.0, rms_norm_eps=1e-5, tie_word_embeddings=True, bos_token_id=128000,
                              eos_token_id=128001)
            dtype = torch.bfloat16
            self.model_name = "Llama-3.2-1B shape (random init, bf16)"
        self.dtype_name = "f32" if dtype == torch.float32 else "bf16"
        self.elem = 4 if dtype == torch.float32 else 2
        self.llm = AsyncAmdLM.from_config(cfg, None, device=dev, dtype=dtype, seed=1234, engine=eng,
                                          batch_size=n_particles, gemms=gemms)
        V = cfg.vocab_size
        g = torch.Generator(device=dev)
        g.manual_seed(4321)
        valid = torch.where(torch.rand(V, device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
        valid[cfg.eos_token_id] = 0.0
        eos1 = torch.full((V,), float("-inf"), device=dev)
        eos1[cfg.eos_token_id] = 0.0
        self.llm.register_masks(torch.stack([valid, eos1]))
        pm = None
        if per_particle_masks:  # every particle its own random third forbidden (+ the EOS-only row behind them)
            own = torch.where(torch.rand((n_particles, V), device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
            own[:, cfg.eos_token_id] = 0.0
            pm, _ = eng.mask_to_bits(torch.cat([own, eos1[None]]))
            del own
        self.per_particle_masks = per_particle_masks
        self.V, self.N, self.max_tokens = V, n_particles, max_tokens
        self.particles_per_step = n_particles
        rs = np.random.default_rng(99)
        base = [list(range(100, 108))] + [[int(t) for t in rs.integers(1000, 30000, 8)] for _ in range(n_prompts - 1)]
        prompts = [base[i % n_prompts] for i in range(n_particles)] if n_prompts > 1 else base[0]
        self.n_prompts = n_prompts
        self.rng = rng
        self.sis = DeviceSIS(self.llm, n_particles, prompts, max_tokens, cfg.eos_token_id,
                             seed=1234, rng=rng, rank=rank, world=world, dist=dist, use_prefix_kv=prefix_kv,
                             use_particle_kv=particle_kv, resample_ess=1.0 if resample else None,
                             force_collectives=force_collectives, kv_in_place=kv_in_place, particle_masks=pm)
        self.prefix_kv = prefix_kv
        self.particle_kv = particle_kv
        self.resample = resample
        self.kernel_bytes = None
        self._events = []
        self._outer = []
        self._bytes = []
        self.unique_hist = []
        self.fed_hist = []
        self.flops_hist = []
        # GEMM work of a step, analytically: 2 x (weights of the body's linear layers) per token fed + 2 x d x V per row
        # through the output embedding (tools/gemm_table.py divides by the Tensile kernels' time in a rocprofv3 trace)
        body = self.llm._body
        self._lin_w = sum(m.weight.numel() for m in body.modules() if type(m).__name__ in ("Linear", "Conv1D"))
        self._head_w = self.llm._head.weight.numel()
        # set-up, not measurement: one untimed pass over the loop's ten batch shapes (context lengths 8..17) so that
        # GEMM algorithm selection and allocator growth happen before bench.py's own warm-up / timed steps
        for _ in range(max_tokens):
            self.sis.step(time_kernel=False)
        self.sis.reset()
        torch.cuda.synchronize(dev)

    def step(self, i, timed):
        if self.sis.t >= self.max_tokens:  # population finished: start the next 10-step loop
            self._collect()
            self.sis.reset()
        U, _ = self.sis.step(time_kernel=timed)
        if timed:
            # algorithmic bytes of this call: the unique logits rows once + mask bit rows + outputs
            n_masks = self.N + 1 if self.per_particle_masks else 2
            self._bytes.append(U * self.V * self.elem + n_masks * ((self.V + 31) // 32) * 4 + self.N * 8
                               + (self.N * self.V * 4 if self.rng != "philox" else 0))  # parity draws: + one noise row per particle
            self.unique_hist.append(U)
            self.fed_hist.append(int(self.sis.last_stats["l_max"]))  # tokens per forward row of this step
            ls = self.sis.last_stats
            self.flops_hist.append(2.0 * self._lin_w * ls["fwd_tokens"] + 2.0 * self._head_w * ls["head_rows"])

    def _collect(self):
        self._events.extend(self.sis.kernel_events)
        self._outer.extend(self.sis.outer_events)
        self.sis.kernel_events = []
        self.sis.outer_events = []

    def kernel_times_us(self):
        self._collect()
        self.kernel_bytes = float(np.mean(self._bytes)) if self._bytes else 0.0
        self.kernel_bytes_each = np.array(self._bytes, dtype=np.float64)  # per timed launch, in the order of the times
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self._events])

    def keep_kernel_times(self, n):
        """Forget the timed launches after the first n (bench.py runs the loop a second time for another line of the same
        JSON object; the roofline block stays the first run's)."""
        self._collect()
        for name in ("_events", "_outer", "_bytes", "unique_hist", "fed_hist", "flops_hist"):
            setattr(self, name, getattr(self, name)[:n])

    def outer_times_us(self):
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self._outer])

    def config(self):
        return {"workload": f"SIS step: {self.N} particles/GPU, {self.model_name}, prompt len 8, <=10 new "
                            "tokens, " + (f"{self.N} per-particle bit masks (standing still: handed over raw at the first step, prepared from the second on)" if self.per_particle_masks
                                          else "2 shared bit masks") + ", device-resident population, "
                            + ("Philox draws" if self.rng == "philox" else "the reference's draws (torch.multinomial's CPU MT19937 stream, "
                               "generated on the device: ids identical to torch's under the seed)")
                            + (f", {self.n_prompts} distinct shared prompts" if self.n_prompts > 1 else "")
                            + (", prompt KV cached (cache_kv semantics, BASELINE config 3)" if self.prefix_kv else "")
                            + (", device-resident KV rows shared by particles with equal contexts (one new token per distinct "
                               "context per step; NOT the reference's re-encode-every-step algorithm)" if self.particle_kv else "")
                            + (", systematic resampling after every step" if self.resample else ""),
                "particles_per_gpu": self.N, "vocab": self.V,
                "rng": "philox" if self.rng == "philox" else "parity (torch CPU generator on the device)",
                "mean_unique_contexts_per_step": float(np.mean(self.unique_hist)) if self.unique_hist else None,
                # what the forward is fed per distinct context, and what the cached prompt KV saves
                "mean_tokens_fed_per_context": float(np.mean(self.fed_hist)) if self.fed_hist else None,
                "gemm_flops_per_step": float(np.mean(self.flops_hist)) if self.flops_hist else None,
                **({"prefixed_tokens_per_context": 8} if self.prefix_kv else {}),
                **({"kv_rows": {k: v for k, v in self.sis.kv_stats.items()}} if self.particle_kv else {})}
