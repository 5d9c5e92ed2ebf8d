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

import numpy as np
import torch

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


async def autobatched_sis(n_particles, llm, mask_selector, prompt_ids, eos_id):
    """README.md:94-98"""
    particles = [Particle(llm, mask_selector, prompt_ids, eos_id) for _ in range(n_particles)]
    while any(p.active for p in particles):
        await asyncio.gather(*[p.extend() for p in particles if p.active])
    return particles


# ------------------------------------------------------------------------------------------------
# device-resident driver
# ------------------------------------------------------------------------------------------------
class DeviceSIS:
    """N particles over one prompt (or one prompt per particle), masks[0] while fewer than `max_tokens`
    tokens were generated and masks[1] afterwards (README.md:57-70's masking function)."""

    def __init__(self, llm, n_particles, prompt_ids, max_tokens, eos_id, seed=0, rng="philox", rank=0, world=1,
                 dist=None, use_prefix_kv=False, use_particle_kv=False):
        self.llm, self.eng, self.dev = llm, llm.engine, llm.device
        self.N, self.max_tokens, self.eos_id = n_particles, max_tokens, eos_id
        self.rank, self.world, self.dist = rank, world, dist
        self.seed = seed
        self.rng_mode = RNG_PHILOX if rng == "philox" else RNG_NOISE
        self.host_rng = None
        if self.rng_mode == RNG_NOISE:
            from .engine import HostRng

            self.host_rng = HostRng(seed)
        prompts = prompt_ids if isinstance(prompt_ids[0], (list, tuple)) else [prompt_ids] * n_particles
        assert len(prompts) == n_particles
        self.prompt_len = torch.tensor([len(p) for p in prompts], dtype=torch.int32, device=self.dev)
        self.max_prompt = max(len(p) for p in prompts)
        self._mask_by_row = len({len(p) for p in prompts}) == 1
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
            distinct = sorted({tuple(p) for p in prompts})
            self._build_prefixes(distinct)
        # Device-resident per-particle KV (beyond the reference, which re-encodes every context every step,
        # hf.py:202-288): step 0 encodes the distinct prompts once and fans their KV out to the particles; every
        # later step feeds ONE token per particle against its own KV rows.  Needs equal prompt lengths.
        self.particle_kv = bool(use_particle_kv)
        if self.particle_kv:
            assert len({len(p) for p in prompts}) == 1, "per-particle KV needs prompts of one length"
            assert not use_prefix_kv
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
        self.contexts = self._ctx0.clone()
        self.lengths = self.prompt_len.clone()
        self.active = torch.ones(self.N, dtype=torch.int32, device=self.dev)
        self.log_weights = torch.zeros(self.N, dtype=torch.float32, device=self.dev)
        self.t = 0
        self.max_len_now = self.max_prompt
        self.last_stats = None
        self.kernel_events = []
        self.pkv = None

    # -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _step_particle_kv(self, time_kernel):
        """Steps t >= 1 with per-particle KV: one new token per particle, no dedup, logits row i = particle i."""
        from transformers import DynamicCache

        eng, llm, dev, N = self.eng, self.llm, self.dev, self.N
        n_active = int(self.active.sum().item())  # the step's only D2H sync
        cache_len = self.max_prompt + self.t - 1  # every particle's KV holds prompt + t - 1 tokens
        rows = torch.arange(N, device=dev)
        # active particles feed their newest token; finished ones a dummy (their rows are ignored afterwards)
        newest = self.contexts[rows, (self.lengths - 1).clamp_min(0).long()]
        ids = torch.where(self.active > 0, newest, torch.zeros_like(newest)).view(N, 1).long()
        pos = torch.full((N, 1), cache_len, dtype=torch.long, device=dev)
        out = llm._body(input_ids=ids, position_ids=pos, past_key_values=self.pkv, use_cache=True)
        self.pkv = out.past_key_values
        logits = llm._lm_head(out.last_hidden_state[:, 0])  # [N, V]
        return self._finish_step(logits, None, N, n_active, time_kernel, l_max=1)

    def _finish_step(self, logits, group_of, U, n_active, time_kernel, l_max):
        eng, llm, N = self.eng, self.llm, self.N
        V = logits.shape[-1]
        mask_id = ((self.lengths - self.prompt_len) >= self.max_tokens).to(torch.int32)
        kw = llm.step_masks(logits.dtype)
        if kw:
            # The mask depends on the number of generated tokens only; with prompts of one length that makes it a
            # function of the context, so identical contexts (one logits row) share it: ids go per ROW and a shared
            # row is reduced once (hf.py:214-220 dedup carried through the particle math).
            if self._mask_by_row and group_of is not None:
                kw["row_mask_id"] = mask_id[self._rep[:U].long()].contiguous()
            elif self._mask_by_row:
                kw["row_mask_id"] = mask_id
            else:
                kw["mask_id"] = mask_id
        if self.rng_mode == RNG_NOISE:
            kw["noise"] = self._parity_noise(group_of if group_of is not None else torch.arange(N, device=self.dev), V)
        if time_kernel:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        logZ, _, tok = eng.step(logits, vocab=V, row_of=group_of, rng_mode=self.rng_mode, seed=self.seed,
                                offset=self.t, particle_base=self.rank * N, want_lse=False, **kw)
        if time_kernel:
            e1.record()
            self.kernel_events.append((e0, e1))
        eng.particles_advance(self.contexts, self.lengths, self.active, self.log_weights, logZ, tok, self.eos_id,
                              self.cap)
        self.t += 1
        self.max_len_now = min(self.max_len_now + 1, self.cap)
        self.last_stats = dict(n_unique=U, n_active=n_active, l_max=l_max, n_rows=U)
        if self.world > 1:
            self.all_weights = self.gather_weights()
        return U, n_active

    @torch.no_grad()
    def step(self, time_kernel=False):
        """One SIS step for every active particle.  Returns (n_unique, n_active_before)."""
        eng, llm, dev, N = self.eng, self.llm, self.dev, self.N
        if self.particle_kv and self.t > 0:
            return self._step_particle_kv(time_kernel)
        ctx_flat = self.contexts.view(-1)
        # finished particles still occupy a row: give them their 1-token stub so they dedup to one group
        lengths_eff = torch.where(self.active > 0, self.lengths, torch.ones_like(self.lengths))
        group_of, rep, ng = eng.group_contexts(ctx_flat, self.starts, lengths_eff)
        self._rep = rep
        head = torch.stack([ng[0], self.active.sum().to(torch.int32)]).cpu()  # the step's only D2H sync
        U, n_active = int(head[0]), int(head[1])
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
        want_kv = self.particle_kv  # step 0: keep the prompts' KV and fan it out to the particles
        out = llm._body(input_ids=ids, attention_mask=None if want_kv else am, position_ids=pos, past_key_values=cache,
                        use_cache=(cache is not None) or want_kv)
        hidden = out.last_hidden_state
        if want_kv:
            from transformers import DynamicCache

            g = group_of.long()
            self.pkv = DynamicCache(ddp_cache_data=[(ly.keys.index_select(0, g), ly.values.index_select(0, g))
                                                    for ly in out.past_key_values.layers])
        h_last = hidden[torch.arange(U, device=dev), last.long()]
        logits = llm._lm_head(h_last)  # [U, V]
        return self._finish_step(logits, group_of, U, n_active, time_kernel, l_max)

    def _parity_noise(self, group_of, V):
        """Exp(1) rows in the order the reference's particles reach torch.multinomial: by dedup group
        (first appearance), duplicates contiguous, inactive particles draw nothing (hf.py:285-288,
        README.md:94-98)."""
        g = group_of.cpu().numpy()
        act = self.active.cpu().numpy() > 0
        idx = np.nonzero(act)[0]
        order = idx[np.argsort(g[idx], kind="stable")]
        noise = torch.ones((self.N, V), dtype=torch.float32)
        block = self.host_rng.exponential(len(order) * V).view(len(order), V)
        noise[torch.from_numpy(order)] = block
        return noise.to(self.dev)

    # -------------------------------------------------------------------------------------------
    def gather_weights(self):
        """All-gather of the per-shard log-weights (RCCL over xGMI when backend is nccl); every rank then
        holds the population's weights and derives identical normalised weights / ESS."""
        out = torch.empty(self.N * self.world, dtype=torch.float32, device=self.dev)
        self.dist.all_gather_into_tensor(out, self.log_weights)
        return out

    def normalized_weights(self):
        lw = self.gather_weights() if self.world > 1 else self.log_weights
        return self.eng.normalize_weights(lw)  # (probs, [logsumexp, ESS])  README.md:108-110

    @torch.no_grad()
    def run(self, max_steps=None):
        steps = 0
        limit = max_steps if max_steps is not None else self.max_tokens + 1
        while steps < limit:
            _, n_active = self.step()
            steps += 1
            if n_active == 0:
                break
        return steps

    def results(self):
        ctx = self.contexts.cpu().numpy()
        ln = self.lengths.cpu().numpy()
        pl = self.prompt_len.cpu().numpy()
        return [list(ctx[i, pl[i]:ln[i]]) for i in range(self.N)], self.log_weights.cpu().numpy()


# ------------------------------------------------------------------------------------------------
# bench.py workload
# ------------------------------------------------------------------------------------------------
class SisBenchWorkload:
    """1024 particles per GPU, GPT-2-small-shaped random-init fp32 model, prompt length 8, <= 10 new tokens,
    two shared {0,-inf} masks (README.md:57-70 shape), in-kernel Philox draws.  BASELINE.json config 2."""

    particles_per_step = 1024

    def __init__(self, eng, dev, rank, world, dist, n_particles=1024, max_tokens=10, prefix_kv=False, particle_kv=False):
        from transformers import GPT2Config

        from .llm import AsyncAmdLM

        cfg = GPT2Config()  # gpt2 small: 12 layers, d=768, 12 heads, vocab 50257
        self.llm = AsyncAmdLM.from_config(cfg, None, device=dev, dtype=torch.float32, seed=1234, engine=eng,
                                          batch_size=n_particles)
        V = cfg.vocab_size
        g = torch.Generator(device=dev)
        g.manual_seed(4321)
        valid = torch.where(torch.rand(V, device=dev, generator=g) < 1 / 3, float("-inf"), 0.0)
        valid[cfg.eos_token_id] = 0.0
        eos1 = torch.full((V,), float("-inf"), device=dev)
        eos1[cfg.eos_token_id] = 0.0
        self.llm.register_masks(torch.stack([valid, eos1]))
        self.V, self.N, self.max_tokens = V, n_particles, max_tokens
        self.particles_per_step = n_particles
        self.sis = DeviceSIS(self.llm, n_particles, list(range(100, 108)), max_tokens, cfg.eos_token_id,
                             seed=1234 + rank, rank=rank, world=world, dist=dist, use_prefix_kv=prefix_kv,
                             use_particle_kv=particle_kv)
        self.prefix_kv = prefix_kv
        self.particle_kv = particle_kv
        self.kernel_bytes = None
        self._events = []
        self._bytes = []
        self.unique_hist = []
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
        U, n_active = self.sis.step(time_kernel=timed)
        if timed:
            # algorithmic bytes of this launch: the unique logits rows once + mask bit rows + outputs
            self._bytes.append(U * self.V * 4 + 2 * ((self.V + 31) // 32) * 4 + self.N * 8)
            self.unique_hist.append(U)

    def _collect(self):
        self._events.extend(self.sis.kernel_events)
        self.sis.kernel_events = []

    def kernel_times_us(self):
        self._collect()
        self.kernel_bytes = float(np.mean(self._bytes)) if self._bytes else 0.0
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self._events])

    def config(self):
        return {"workload": "SIS step: 1024 particles/GPU, gpt2-small shape (random init, fp32), prompt len 8, <=10 new "
                            "tokens, 2 shared bit masks, device-resident population, Philox draws"
                            + (", prompt KV cached (cache_kv semantics, BASELINE config 3)" if self.prefix_kv else "")
                            + (", device-resident per-particle KV (one new token per particle per step; NOT the "
                               "reference's re-encode-every-step algorithm)" if self.particle_kv else ""),
                "particles_per_gpu": self.N, "vocab": self.V, "rng": "philox",
                "mean_unique_contexts_per_step": float(np.mean(self.unique_hist)) if self.unique_hist else None}
