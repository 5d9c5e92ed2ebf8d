"""AsyncLM interface and the MI355X backend behind it.

Mirrors, name for name, the surface of the reference that the hot path exposes:
  * `AsyncLM`                       genlm/backend/llm/base.py:9-179
  * `AsyncAmdLM` (= AsyncTransformer of genlm/backend/llm/hf.py:73-422 with the per-step work moved
    to the HIP library): `from_name`, `next_token_logprobs[_sync|_uncached]`,
    `batch_next_token_logprobs[_sync]`, `sample`, `batch_sample`, `cache_kv`, `clear_cache`,
    `clear_kv_cache`, `reset_async_queries`, `walk_cache`, `add_query`, `batch_evaluate_queries`,
    attributes `model tokenizer device cache queries batch_size timeout timer byte_vocab str_vocab`
  * `load_model_by_name`            genlm/backend/llm/__init__.py:10-43 (extra backend value "amd")
plus one addition the reference leaves to user code (README.md:82-91): `next_token_step`, the fused
log-softmax + mask + logsumexp + categorical draw, autobatched through the same queue.

All device arithmetic after the transformer's final hidden state goes through `HipEngine`
(C ABI, include/glb.h).  There is no CPU fallback: without the built library and a HIP device the
constructor raises.
"""
import asyncio
from abc import ABC, abstractmethod

import numpy as np
import torch

from .cache import KVPrefix, RowLRU, TokenTrie
from .tokenization import decode_vocab

MASK_NONE, MASK_BITS, MASK_F32 = 0, 1, 2
RNG_NONE, RNG_PHILOX, RNG_NOISE = 0, 1, 2


class AsyncLM(ABC):
    """base.py:9-179."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer
        self.byte_vocab, self.str_vocab = decode_vocab(self.tokenizer)

    @abstractmethod
    async def next_token_logprobs(self, token_ids):
        pass

    @abstractmethod
    def next_token_logprobs_sync(self, token_ids):
        pass

    async def batch_next_token_logprobs(self, token_ids_list):
        """base.py:47-60"""
        logprobs = await asyncio.gather(*[self.next_token_logprobs(t) for t in token_ids_list])
        return torch.stack(logprobs)

    def batch_next_token_logprobs_sync(self, token_ids_list):
        """base.py:62-73"""
        return torch.stack([self.next_token_logprobs_sync(t) for t in token_ids_list])

    def add_new_lora(self, lora_path, lora_name):
        raise NotImplementedError("add_new_lora must be implemented by subclasses")

    def set_lora(self, lora_path, lora_name):
        raise NotImplementedError("set_lora must be implemented by subclasses")

    def clear_lora(self):
        raise NotImplementedError("clear_lora must be implemented by subclasses")

    def clear_cache(self):
        pass

    def _draw(self, logprobs, temperature, generator_state):
        """One categorical draw from softmax(logprobs / temperature); subclasses override."""
        probs = torch.softmax(logprobs / temperature, dim=-1)
        return torch.multinomial(probs.cpu(), num_samples=1, generator=generator_state).item()

    def _make_generator(self, seed):
        if seed is None:
            return None
        g = torch.Generator()
        g.manual_seed(seed)
        return g

    async def sample(self, prompt_token_ids, max_tokens, eos_token_ids, temperature=1.0, seed=None):
        """base.py:110-146"""
        gen = self._make_generator(seed)
        generated = []
        for _ in range(max_tokens):
            logprobs = await self.next_token_logprobs(prompt_token_ids + generated)
            tok = self._draw(logprobs, temperature, gen)
            if tok in eos_token_ids:
                break
            generated.append(tok)
        return generated

    async def batch_sample(self, prompt_token_ids_list, max_tokens, eos_token_ids, temperature=1.0, seed=None):
        """base.py:148-179 (every sequence gets its own generator seeded with `seed`)"""
        return await asyncio.gather(*[
            self.sample(prompt_token_ids=p, max_tokens=max_tokens, eos_token_ids=eos_token_ids,
                        temperature=temperature, seed=seed)
            for p in prompt_token_ids_list
        ])


class Query:
    """A pending request (hf.py:14-70).  Padding, masks and position ids are no longer built per query in
    Python: the gather kernel produces them for the whole batch (glb_gather_padded)."""

    __slots__ = ("prompt", "future", "past", "past_len", "first_new", "kind", "mask_id")

    def __init__(self, prompt, future, past=None, first_new=0, kind="logprobs", mask_id=0):
        self.prompt = prompt
        self.future = future
        self.past = past
        self.past_len = len(past) if past is not None else 0
        self.first_new = first_new  # first prompt position whose next-token row the caller needs
        self.kind = kind            # "logprobs" | "step"
        self.mask_id = mask_id


class AsyncAmdLM(AsyncLM):
    """Autobatching wrapper around a HuggingFace causal LM on one MI355X (hf.py:73-422)."""

    @classmethod
    def from_name(cls, model_id, bitsandbytes_opts=None, hf_opts=None, **kwargs):
        """hf.py:80-112.  bitsandbytes is a CUDA-only dependency and is rejected."""
        from transformers import AutoModelForCausalLM, AutoTokenizer

        if bitsandbytes_opts:
            raise NotImplementedError("bitsandbytes quantisation is CUDA-only; not available on MI355X")
        _hf_opts = {"torch_dtype": "auto"}
        if hf_opts:
            _hf_opts.update(hf_opts)
        device = _hf_opts.pop("device", None) or _hf_opts.pop("device_map", None) or "cuda:0"
        if device == "auto":
            device = "cuda:0"
        tok = AutoTokenizer.from_pretrained(model_id)
        mod = AutoModelForCausalLM.from_pretrained(model_id, **_hf_opts).to(device)
        return cls(mod, tok, **kwargs)

    @classmethod
    def from_config(cls, config, tokenizer, device="cuda:0", dtype=torch.float32, seed=0, **kwargs):
        """Random-init model of a given architecture (no hub access needed; synthetic benchmarks/tests)."""
        from transformers import AutoModelForCausalLM

        torch.manual_seed(seed)
        n_params = getattr(config, "num_hidden_layers", 0) * getattr(config, "hidden_size", 0) ** 2 * 12
        if n_params > 2e9 and torch.device(device).type == "cuda":
            # a model of several billion parameters: made on the device in its own dtype (on the host it would first be
            # tens of GB of float32 and minutes of initialisation)
            old = torch.get_default_dtype()
            torch.set_default_dtype(dtype)
            try:
                with torch.device(device):
                    mod = AutoModelForCausalLM.from_config(config)
            finally:
                torch.set_default_dtype(old)
            mod = mod.to(dtype)
        else:
            mod = AutoModelForCausalLM.from_config(config).to(dtype).to(device)
        return cls(mod, tokenizer, **kwargs)

    @staticmethod
    def _post_head(config):
        """What the model's own forward applies to `lm_head(hidden)` (the reference reads `model(...).logits`, hf.py:275):
        Cohere `logit_scale`, Granite `logits_scaling`, Gemma-2 `final_logit_softcapping`.  Returned as
        (multiplier, softcap); other families leave the head's output alone."""
        mult, cap = 1.0, None
        mt = getattr(config, "model_type", "")
        if mt.startswith("cohere") and getattr(config, "logit_scale", None) is not None:
            mult *= float(config.logit_scale)
        if mt.startswith("granite") and getattr(config, "logits_scaling", None):
            mult /= float(config.logits_scaling)
        if getattr(config, "final_logit_softcapping", None):
            cap = float(config.final_logit_softcapping)
        return mult, cap

    def _log_softmax(self, logits):
        """log-prob rows of `logits` in the dtype this backend returns (glb_log_softmax_rows)"""
        return self.engine.log_softmax_rows(logits, out_dtype=logits.dtype if self._lp_model_dtype else torch.float32)

    def _lm_head(self, hidden):
        """Logits of the given hidden rows exactly as `model(...).logits` has them: output embedding, then the
        family's post-head scaling / soft-capping."""
        x = self._head(hidden)
        if self._head_mult != 1.0:
            x = x * self._head_mult
        if self._head_cap is not None:
            x = torch.tanh(x / self._head_cap) * self._head_cap
        return x

    @torch.no_grad()
    def __init__(self, hf_model, hf_tokenizer, batch_size=20, timeout=0.02, engine=None, fuse_activations=True,
                 kv_budget_bytes=8 << 30, logprob_budget_bytes=16 << 30, auto_kv_rows=0, auto_kv_cap=64,
                 logprob_dtype="float32", glb_attention=True, merge_mlp=True, contract=None, gemms="library"):
        """The caller's `hf_model` is never modified (hf.py:114-140 leaves it alone too): with `fuse_activations` or
        `glb_attention` the forwards of this backend run on a private SHADOW of its module tree that shares every weight
        (fuse.shadow_model); `self.model` stays the caller's object.
        fuse_activations: in the shadow, GPT-2's eight-op `gelu_new` becomes one GELU kernel, Llama-family RMSNorm one
        `rms_norm` call, the rotary embedding three elementwise ops per tensor instead of six (same functions, different
        rounding); `merge_mlp` (Llama family): for few-token forwards gate_proj and up_proj are one GEMM on a derived
        [gate; up] weight - a second copy of those two matrices in device memory (False: not made).  glb_attention: the
        shadow's attention goes through this library's kernels where they apply (padded
        batches of short contexts: glb_short_attention; the in-place one-token forward over KV rows: glb_slab_attention),
        everything else still runs the SDPA path.  Both False: the forwards run on `hf_model` itself.
        logprob_dtype: "float32" (default: every row `next_token_logprobs` returns is float32, whatever the checkpoint's
        dtype) or "model" - rows in the model's own dtype, what the reference returns (cache.py:96 keeps the dtype; for a
        bfloat16 checkpoint a third fewer bytes per materialised row: glb_log_softmax_rows' out_dtype).
        auto_kv_rows > 0: `batch_next_token_step` keeps the KV of the contexts it evaluates in that many slab rows of
        `auto_kv_cap` positions and feeds one token to every context whose first L - 1 tokens it finds there
        (autokv.AutoKV; off by default: the reference re-encodes, hf.py:202-288).
        contract: "poly" / "hw" / "auto" - the arithmetic of the fused step's terms (HipEngine; None: the engine's own, "auto"
        for an engine made here: the hardware exponential for 16-bit logits, the polynomial for float32).
        gemms: "library" (default: PyTorch picks the GEMM kernels as it always does) or "recorded" - the forward's GEMM
        shapes run by the rocBLAS / hipBLASLt solution recorded for them in tuned/<arch>.csv (gemm_tuning: PyTorch TunableOp,
        a PROCESS-WIDE switch that also reaches the caller's other GEMMs; 8-10 % of a step at 1024 particles; silently the
        library's defaults when the file was made by another build of the libraries - `self.gemm_shapes` says how many
        shapes were taken over).  `close()` gives the switch back."""
        self.model = hf_model
        self.tokenizer = hf_tokenizer
        self.device = hf_model.device
        if gemms not in ("library", "recorded"):
            raise ValueError(f"gemms must be 'library' or 'recorded', got {gemms!r}")
        self.gemms, self.gemm_shapes = gemms, 0
        if gemms == "recorded" and self.device.type == "cuda":
            from . import gemm_tuning

            self.gemm_shapes = gemm_tuning.acquire(self.device)
            self._gemms_held = True
        if engine is None:
            from .engine import HipEngine  # raises without the library / a HIP device

            engine = HipEngine(self.device, contract=contract or "auto")
        elif contract is not None:
            if contract not in engine.CONTRACTS:
                raise ValueError(f"contract must be one of {engine.CONTRACTS}, got {contract!r}")
            engine.contract = contract
        self.engine = engine
        if logprob_dtype not in ("float32", "model"):
            raise ValueError(f"logprob_dtype must be 'float32' or 'model', got {logprob_dtype!r}")
        self._lp_model_dtype = logprob_dtype == "model"
        self.cache = TokenTrie()
        self.queries = []
        self._sq = None  # pending `next_token_step` requests without a cached prefix: [contexts, mask ids, ONE future]
        self.batch_size = batch_size
        self.timeout = timeout
        self.timer = None
        self.model.eval()
        self.glb_attention = False
        self.fused = []
        self._net = self.model  # what this backend's forwards run on: the caller's model, or its shadow
        want_attention = bool(glb_attention) and self.device.type == "cuda"
        if fuse_activations or want_attention:
            from .fuse import fuse_shadow, shadow_model

            self._net = shadow_model(self.model)
            self.fused = fuse_shadow(self._net, activations=bool(fuse_activations), merge_mlp=bool(merge_mlp))
            if want_attention:
                from .kv import use_glb_attention

                self.glb_attention = use_glb_attention(self._net, self.engine)
        self._head = self._net.get_output_embeddings()
        if self._head is None:
            raise NotImplementedError(f"{type(hf_model).__name__} has no output embedding (get_output_embeddings() is None)")
        self._head_mult, self._head_cap = self._post_head(self.model.config)
        self._body = self._net.base_model
        from .kv import PrefixLRU

        self._kv_tokens = {}  # id(trie node) -> the prefix's token ids (for the device prefix table)
        self._ptab = None
        # prompt prefixes pinned by cache_kv, least recently used out first; an entry that leaves the store takes its
        # token tuple and the device prefix table (which holds pointers into its slabs) with it
        self._kv_lru = PrefixLRU(kv_budget_bytes, on_remove=self._forget_prefix)
        # log-prob rows on trie nodes: views into their batch's [R, V] slab, least recently used slab out first
        self._rows = RowLRU(logprob_budget_bytes)
        # KV rows that follow the contexts handed to batch_next_token_step (off unless auto_kv_rows > 0)
        self._auto_kv = None
        if auto_kv_rows:
            from .autokv import AutoKV

            self._auto_kv = AutoKV(self, auto_kv_rows, auto_kv_cap)
        # fused-step state
        self._mask_kind = MASK_NONE
        self._masks = None
        self._masks_prepared = {}
        self._rng_mode = RNG_PHILOX
        self._rng_seed = 0
        self._noise_src = None  # "torch" draws: the CPU generator's stream on the device, made at the first evaluation
        self._batch_counter = 0
        self.stats = {"batches": 0, "queries": 0, "unique": 0, "rows": 0}
        if hf_tokenizer is not None:
            super().__init__(tokenizer=self.tokenizer)
        else:  # model-only use (synthetic benchmarks): no vocabulary to decode
            self.byte_vocab, self.str_vocab = None, None

    def refresh_weights(self):
        """Call after changing the model's weights in a way PyTorch's version counters do not see.  The shadow keeps derived
        copies of some weights ([q; k; v] and [gate; up] as one matrix each: fuse.py) and captured hipGraphs of the
        one-token forward read the copies they were captured with; both notice `param.copy_()` / `param.mul_()` and replaced
        parameters by themselves (`Tensor._version`, identity, address), but NOT writes through `param.data` (EMA and
        weight-merging code does that: `p.data.copy_(...)` bumps no counter and moves no address).  This drops every
        derived copy and every captured graph - they are rebuilt from the current weights at the next forward - and the
        cached log-prob rows and KV (made with the old weights: clear_cache())."""
        for mod in self._net.modules():
            mod.__dict__.pop("_glb_qkv", None)
            mod.__dict__.pop("_glb_gate_up", None)
        self.weights_epoch = getattr(self, "weights_epoch", 0) + 1  # SlabForward drops its graphs when this moves
        self.clear_cache()

    def close(self):
        """Give back what this backend holds of the process's state: the recorded GEMM solutions (gemms="recorded").  The
        object stays usable (its GEMMs then run by the library's defaults); safe to call more than once."""
        if getattr(self, "_gemms_held", False):
            from . import gemm_tuning

            self._gemms_held = False
            self.gemm_shapes = 0
            gemm_tuning.release()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def _forget_prefix(self, node):
        self._kv_tokens.pop(id(node), None)
        self._ptab = None

    # ---- cache management (hf.py:142-164) ---------------------------------------------------------
    def clear_cache(self):
        self._kv_lru.clear()
        self._kv_tokens.clear()
        self._rows.clear()
        self.cache = TokenTrie()
        if self._auto_kv is not None:
            self._auto_kv.reset()

    def clear_kv_cache(self):
        self._kv_lru.clear()
        self.cache.clear_kv_cache()
        if self._auto_kv is not None:
            self._auto_kv.reset()

    def reset_async_queries(self):
        self.queries = []
        self._sq = None

    @torch.no_grad()
    def cache_kv(self, prompt_tokens):
        """hf.py:155-164: run the prompt once, cache every position's log-probs and keep the KV states
        on the prompt's last node so later queries only feed their new tokens."""
        key = tuple(int(t) for t in prompt_tokens)
        # extend_cache makes fresh nodes, so caching the same prompt again (or one that shares leading tokens) orphans
        # the node that held the KV so far: its entry leaves the store instead of lingering under the budget
        for old in [n for n in self._kv_lru.nodes() if self._kv_tokens.get(id(n)) == key]:
            self._kv_lru.drop(old)
        ids = torch.tensor([prompt_tokens], device=self.device)
        out = self._body(input_ids=ids, use_cache=True)
        logits = self._lm_head(out.last_hidden_state[0])
        node = self.cache.extend_cache(0, prompt_tokens, logits, 0, engine=self.engine, store=self._rows,
                                       out_dtype=logits.dtype if self._lp_model_dtype else None)
        # (re-created ancestors: an older, shorter prefix whose node was replaced on the way is unreachable from the
        # trie now and would only hold memory)
        reach = set()
        walk = self.cache
        for t in prompt_tokens:
            if not walk.has_token(t):
                break
            walk = walk.get_token(t)
            reach.add(id(walk))
        for old in self._kv_lru.nodes():
            toks = self._kv_tokens.get(id(old))
            if toks is not None and len(toks) <= len(key) and key[:len(toks)] == toks and id(old) not in reach:
                self._kv_lru.drop(old)
        # pinned under a byte budget: the least recently used prefix loses its KV (its log-prob rows stay), the policy
        # of cache.py:103-191 applied to whole prefix slabs
        self._kv_lru.put(node, KVPrefix.from_hf_cache(out.past_key_values))
        self._kv_tokens[id(node)] = key
        self._ptab = None

    # ---- LoRA hooks (hf.py:166-200): weight management is outside the hot path ---------------------
    def add_new_lora(self, lora_path, lora_name="lora_1"):
        raise NotImplementedError("LoRA adapters are out of scope for the MI355X hot-path backend")

    def set_lora(self, lora_path=None, lora_name="lora_1"):
        raise NotImplementedError("LoRA adapters are out of scope for the MI355X hot-path backend")

    def clear_lora(self):
        raise NotImplementedError("LoRA adapters are out of scope for the MI355X hot-path backend")

    # ---- fused-step configuration -------------------------------------------------------------------
    def register_masks(self, masks):
        """masks: float tensor [K, V] (or list of [V]) of additive log-masks (README.md:57-70).  {0,-inf}
        masks are packed to bit rows on the device; anything else is kept as float."""
        if isinstance(masks, (list, tuple)):
            masks = torch.stack([m.to(torch.float32) for m in masks])
        masks = masks.to(self.device, torch.float32).contiguous()
        bits, flag = self.engine.mask_to_bits(masks)
        self._masks_prepared = {}
        if int(flag.item()) == 0:
            self._mask_kind, self._masks = MASK_BITS, bits
            self._mask_vocab = masks.shape[1]
        else:
            self._mask_kind, self._masks = MASK_F32, masks
        return masks.shape[0]

    def step_masks(self, logits_dtype):
        """Mask arguments of `HipEngine.step` for logits of `logits_dtype`: bit masks are brought into the kernels'
        layout once per element width (glb_mask_prepare) and reused by every step."""
        if self._mask_kind == MASK_NONE:
            return {}
        if self._mask_kind == MASK_F32:
            return dict(mask_kind=MASK_F32, mask=self._masks)
        wide = logits_dtype == torch.float32
        if wide not in self._masks_prepared:
            self._masks_prepared[wide] = self.engine.prepare_masks(self._masks, self._mask_vocab, logits_dtype)
        return dict(mask=self._masks_prepared[wide])

    def set_rng(self, mode="philox", seed=0):
        """"philox": in-kernel counter RNG (fast).  "torch": parity with the reference's CPU
        torch.multinomial under torch.manual_seed(seed) (README.md:87): the exponentials come from the same MT19937
        stream, in the reference's particle order - generated on the device (engine.DeviceRng)."""
        if mode == "philox":
            self._rng_mode, self._rng_seed, self._noise_src = RNG_PHILOX, int(seed), None
        elif mode == "torch":
            self._rng_mode, self._rng_seed, self._noise_src = RNG_NOISE, int(seed), None
        else:
            raise ValueError(f"unknown rng mode {mode!r}")
        self._batch_counter = 0

    def _noise_rows(self, V):
        if self._noise_src is None:
            self._noise_src = self.engine.noise_rng(self._rng_seed, V)
        return self._noise_src

    # ---- the batched evaluation (hf.py:202-288) -------------------------------------------------------
    @torch.no_grad()
    def batch_evaluate_queries(self):
        self._fire_step_batch()
        queries, self.queries = self.queries, []
        if len(queries) == 0:
            return
        try:
            self._evaluate(queries)
        except Exception as e:  # vllm.py:396-400 behaviour: nobody is left waiting
            stranded = [q.future for q in queries if q.future is not None and not q.future.done()]
            for f in stranded:
                f.set_exception(e)
            if not stranded:
                raise

    def _evaluate(self, queries):
        eng, dev = self.engine, self.device
        n = len(queries)
        if all(q.kind == "step" and q.past is None for q in queries):
            # a population of README particles (README.md:82-91) and no cached prefix: the whole batch goes through the
            # vectorised pipeline of batch_next_token_step_sync - three host arrays in, two out, one loop to resolve
            logZ, tok = self.batch_next_token_step_sync([q.prompt for q in queries], [q.mask_id for q in queries])
            for q, z, t in zip(queries, logZ.tolist(), tok.tolist()):
                if q.future is not None and not q.future.done():
                    q.future.set_result((z, t))
            return
        if (self._auto_kv is not None and not self._kv_tokens
                and all(q.kind == "logprobs" and q.past is None and q.first_new == len(q.prompt) - 1 for q in queries)):
            # every request wants the row after its LAST token only (its shorter prefixes are in the trie - a population
            # that grew by one token): the contexts find their KV rows (autokv.AutoKV) and one token each is fed
            import itertools

            lens = np.fromiter((len(q.prompt) for q in queries), np.int32, n)
            flat = np.fromiter(itertools.chain.from_iterable(q.prompt for q in queries), np.int32, int(lens.sum()))
            starts = np.zeros(n, np.int64)
            starts[1:] = np.cumsum(lens[:-1])
            tok_d, st_d, ln_d = (torch.from_numpy(a).to(dev) for a in (flat, starts, lens))
            group_of, rep, ng = eng.group_contexts(tok_d, st_d, ln_d)
            logits, row_of_group, _, U, _ = self._auto_kv.logits(tok_d, st_d, ln_d, group_of, rep, ng)
            lp = self._log_softmax(logits)
            rows = torch.cat([row_of_group[group_of.long()], eng.error_word()]).cpu().tolist()
            eng.raise_if_failed(rows.pop(), what="glb_log_softmax_rows")  # no NaN row may reach the trie
            self._batch_counter += 1
            self.stats["batches"] += 1
            self.stats["queries"] += n
            self.stats["unique"] += U
            self.stats["rows"] += U
            for q, r in zip(queries, rows):
                if q.future is not None and not q.future.done():
                    q.future.set_result((lp, r, q.first_new))
            return
        # -- flatten: context i = [prefix slot, past_len, prompt...]; two header words make the dedup key
        #    (hf.py:216 keys on the prompt only; adding the prefix identity cannot merge unequal requests)
        prefixes, prefix_slot = [], {}
        lens = np.empty(n, np.int32)
        for i, q in enumerate(queries):
            lens[i] = len(q.prompt) + 2
        starts = np.zeros(n, np.int64)
        if n > 1:
            starts[1:] = np.cumsum(lens[:-1])
        flat = np.empty(int(lens.sum()), np.int32)
        for i, q in enumerate(queries):
            slot = 0
            if q.past is not None:
                key = id(q.past)
                if key not in prefix_slot:
                    prefix_slot[key] = len(prefixes) + 1
                    prefixes.append(q.past)
                slot = prefix_slot[key]
            s = starts[i]
            flat[s] = slot
            flat[s + 1] = q.past_len
            flat[s + 2:s + lens[i]] = q.prompt
        tok_d = torch.from_numpy(flat).to(dev)
        st_d = torch.from_numpy(starts).to(dev)
        ln_d = torch.from_numpy(lens).to(dev)

        # -- shared-prefix dedup on the device, first-appearance order (glb_group_contexts)
        group_of_d, rep_d, ng_d = eng.group_contexts(tok_d, st_d, ln_d)
        U = int(ng_d.item())
        group_of = group_of_d.cpu().numpy()
        rep = rep_d[:U].cpu().numpy()
        uniq = [queries[r] for r in rep]
        members = [[] for _ in range(U)]
        for i, g in enumerate(group_of):
            members[g].append(queries[i])

        p_max = max(q.past_len for q in uniq)
        l_max = max(len(q.prompt) for q in uniq)
        pad_id = getattr(self.tokenizer, "pad_token_id", None) if self.tokenizer is not None else None
        pad_id = 0 if pad_id is None else pad_id

        # -- ragged -> padded gather (glb_gather_padded).  starts/lengths are shifted so that
        #    tokens[start + base + t] is prompt token t, with base = past_len.
        past_len = np.array([q.past_len for q in queries], np.int32)
        base_d = torch.from_numpy(past_len).to(dev)
        st_adj = st_d + 2 - base_d.to(torch.int64)
        ln_adj = ln_d - 2 + base_d
        ids, am, pos, _last = eng.gather_padded(tok_d, st_adj, ln_adj, rep_d, U, base_d, pad_id, p_max, l_max)

        # -- batched prefix KV (glb_gather_kv_padded), one launch per layer and K/V
        cache = None
        if p_max > 0:
            from transformers import DynamicCache

            slot_of_u = np.array([prefix_slot[id(q.past)] - 1 if q.past is not None else -1 for q in uniq], np.int32)
            slot_d = torch.from_numpy(slot_of_u).to(dev)
            plen_d = torch.tensor([len(p) for p in prefixes], dtype=torch.int32, device=dev)
            p0 = prefixes[0]
            for p in prefixes[1:]:  # one gather launch per layer serves every prefix: they must come from one model
                if (p.heads, p.head_dim, p.dtype, len(p.layers)) != (p0.heads, p0.head_dim, p0.dtype, len(p0.layers)):
                    raise ValueError("cached prefixes of different KV shapes in one batch")
            data = []
            for layer in range(len(p0.layers)):
                kv = []
                for j in range(2):
                    ptrs = torch.tensor([p.layers[layer][j].data_ptr() for p in prefixes], dtype=torch.int64, device=dev)
                    kv.append(eng.gather_kv_padded(ptrs, plen_d, slot_d, p0.heads, p0.head_dim, p_max, p0.dtype))
                data.append(tuple(kv))
            cache = DynamicCache(ddp_cache_data=data)

        # -- transformer body (PyTorch-ROCm; the only MFMA work on the path)
        hidden = self._body(input_ids=ids, attention_mask=am, position_ids=pos, past_key_values=cache,
                            use_cache=cache is not None).last_hidden_state  # [U, l_max, d]

        # -- which (unique, position) rows are needed: log-prob queries want every new position,
        #    step queries only the last one
        row_u, row_t, lp_rows, step_row = [], [], {}, {}
        for u, q in enumerate(uniq):
            L = len(q.prompt)
            lp_first = min((m.first_new for m in members[u] if m.kind == "logprobs"), default=None)
            if lp_first is not None:
                lp_rows[u] = (len(row_u), lp_first)
                for t in range(lp_first, L):
                    row_u.append(u)
                    row_t.append(t)
                step_row[u] = len(row_u) - 1
            elif any(m.kind == "step" for m in members[u]):
                step_row[u] = len(row_u)
                row_u.append(u)
                row_t.append(L - 1)
        ru = torch.from_numpy(np.asarray(row_u, np.int64)).to(dev)
        rt = torch.from_numpy(np.asarray(row_t, np.int64)).to(dev)
        logits = self._lm_head(hidden[ru, rt])  # [R, V] plain library GEMM on the gathered rows
        V = logits.shape[-1]
        R = logits.shape[0]

        # -- log-prob rows in one launch (glb_log_softmax_rows; replaces cache.py:93-98)
        lp_slab, slab_row = None, None
        if lp_rows:
            need = sorted(r for u, (r0, f) in lp_rows.items() for r in range(r0, r0 + len(uniq[u].prompt) - f))
            if len(need) == R:
                lp_slab = self._log_softmax(logits)
            else:
                idx = torch.from_numpy(np.asarray(need, np.int64)).to(dev)
                lp_slab = self._log_softmax(logits[idx].contiguous())
                slab_row = {r: i for i, r in enumerate(need)}

        # -- fused step for every step query, in resolution order (group order, duplicates contiguous)
        step_q = [(u, m) for u in range(U) for m in members[u] if m.kind == "step"]
        step_out = None
        if step_q:
            row_of = torch.from_numpy(np.fromiter((step_row[u] for u, _ in step_q), np.int32, len(step_q))).to(dev)
            mask_id = torch.from_numpy(np.fromiter((m.mask_id for _, m in step_q), np.int32, len(step_q))).to(dev)
            kw = self.step_masks(logits.dtype)
            if kw:
                kw["mask_id"] = mask_id
            if self._rng_mode == RNG_NOISE:  # (step_q is in resolution order already: by group, duplicates contiguous)
                kw["noise"] = self._noise_rows(V).rows(len(step_q))
            logZ, _lse, tok = eng.step(logits, vocab=V, row_of=row_of, rng_mode=self._rng_mode, seed=self._rng_seed,
                                       offset=self._batch_counter, want_lse=False, **kw)
            step_out = (logZ.cpu().tolist(), tok.cpu().tolist())
            eng.raise_if_failed(tokens=step_out[1])
        if lp_slab is not None:  # no NaN row may reach the trie: the error word of the log-softmax launch (one small copy)
            eng.raise_if_failed(int(eng.error_word().item()), what="glb_log_softmax_rows")
        self._batch_counter += 1
        self.stats["batches"] += 1
        self.stats["queries"] += n
        self.stats["unique"] += U
        self.stats["rows"] += R

        # -- fan out (hf.py:285-288 order)
        si = 0
        for u in range(U):
            rows = None
            if u in lp_rows:
                r0, first = lp_rows[u]
                cnt = len(uniq[u].prompt) - first
                # (slab, index of the row of position `first`, first): no view per query
                rows = (lp_slab, r0 if slab_row is None else slab_row[r0], first)
            for m in members[u]:
                if m.kind == "step":
                    res = (step_out[0][si], step_out[1][si])
                    si += 1
                else:
                    res = rows
                if m.future is not None and not m.future.done():
                    m.future.set_result(res)

    # ---- queueing (hf.py:290-312) -----------------------------------------------------------------------
    def add_query(self, query, future, past, first_new=0, kind="logprobs", mask_id=0):
        """hf.py:290-312: queue the request; evaluate when the batch is full or `timeout` after the LAST request.
        (The reference cancels and re-arms its timer on every request; here one timer re-checks a moving deadline,
        which fires at the same moment without 2 x batch_size timer operations per batch.)"""
        self.queries.append(Query(query, future, past, first_new=first_new, kind=kind, mask_id=mask_id))
        if len(self.queries) >= self.batch_size:
            if self.timer:
                self.timer.cancel()
                self.timer = None
            self.batch_evaluate_queries()
        else:
            loop = asyncio.get_running_loop()
            self._deadline = loop.time() + self.timeout
            if self.timer is None:
                self.timer = loop.call_later(self.timeout, self._on_timer)

    def _on_timer(self):
        self.timer = None
        if not self.queries and self._sq is None:
            return
        loop = asyncio.get_running_loop()
        remaining = self._deadline - loop.time()
        if remaining > 1e-4:
            self.timer = loop.call_later(remaining, self._on_timer)
        else:
            self.batch_evaluate_queries()

    def walk_cache(self, token_ids):
        """hf.py:314-344: deepest matching node, tokens matched, deepest KV on the way and its depth."""
        node = self.cache
        prev = None
        next_token_index = 0
        past = None
        base = 0
        while next_token_index < len(token_ids):
            if node.past_key_values is not None:
                past = node.past_key_values
                base = next_token_index
                self._kv_lru.touch(node)
            if node.has_token(token_ids[next_token_index]):
                prev, node = node, node.get_token(token_ids[next_token_index])
                next_token_index += 1
            else:
                break
        if next_token_index == len(token_ids) and prev is not None:
            if not node.has_row():  # the row was evicted under the byte budget (RowLRU): a miss on the last position
                node, next_token_index = prev, next_token_index - 1
            else:
                self._rows.touch(node)
        return node, next_token_index, past, base

    async def next_token_logprobs(self, token_ids):
        """hf.py:346-373.  Returns a float32 [V] tensor on the model's device."""
        if not token_ids:
            raise ValueError("Token ids must not be empty")
        node, next_token_index, past, base = self.walk_cache(token_ids)
        if next_token_index == len(token_ids):
            return node.logprobs
        future = asyncio.get_running_loop().create_future()
        self.add_query(token_ids[base:], future, past, first_new=next_token_index - base)
        slab, row0, first = await future  # slab[row0 + t - first]: the row after prompt position t (= context position base + t)
        t, i = node.extend_cache_lazy(next_token_index, token_ids, slab, base + first - row0, store=self._rows)
        return t[i]

    def _evaluate_one(self, prompt, past, first_new):
        """Synchronous single-query evaluation through the same batched machinery."""

        class _Slot:
            def __init__(self):
                self.value = None
                self.exc = None

            def done(self):
                return self.value is not None or self.exc is not None

            def set_result(self, v):
                self.value = v

            def set_exception(self, e):
                self.exc = e

        slot = _Slot()
        self._evaluate([Query(prompt, slot, past, first_new=first_new)])
        return slot.value

    @torch.no_grad()
    def next_token_logprobs_sync(self, token_ids):
        """hf.py:375-402 (uses the walked KV prefix; the reference passes the matched node's, hf.py:396)."""
        if not token_ids:
            raise ValueError("Token ids must not be empty")
        node, next_token_index, past, base = self.walk_cache(token_ids)
        if next_token_index == len(token_ids):
            return node.logprobs
        slab, row0, first = self._evaluate_one(token_ids[base:], past, next_token_index - base)
        t, i = node.extend_cache_lazy(next_token_index, token_ids, slab, base + first - row0, store=self._rows)
        return t[i]

    @torch.no_grad()
    def next_token_logprobs_uncached(self, token_ids):
        """hf.py:404-422: no KV, no output cache, no batching."""
        if not token_ids:
            raise ValueError("Token ids must not be empty")
        ids = torch.tensor([token_ids], device=self.device)
        h = self._body(input_ids=ids, use_cache=False).last_hidden_state[0, -1:]
        return self._log_softmax(self._lm_head(h))[0]  # (a single row: the three-launch form, no waits inside)

    # ---- fused particle step (README.md:82-91 moved behind the queue) ------------------------------------
    async def next_token_step(self, token_ids, mask_id=0):
        """Returns (logZ, token): logZ = logsumexp(next_token_logprobs + mask[mask_id]) and a categorical
        draw from the masked, renormalised distribution; token is -1 if the mask forbids everything."""
        if not token_ids:
            raise ValueError("Token ids must not be empty")
        if len(self._kv_lru):  # a cached KV prefix may serve this request: the trie walk finds it, the request is queued on its own
            _node, _n, past, base = self.walk_cache(token_ids)
            future = asyncio.get_running_loop().create_future()
            self.add_query(token_ids[base:] if base else token_ids, future, past, kind="step", mask_id=mask_id)
            return await future
        # No prefix is cached, so there is nothing to walk and nothing per-request to keep: the requests of a batch share
        # ONE future (a population of README particles awaits the same evaluation, hf.py:290-312's queue without a future,
        # a Query object and a timer operation per particle) and pick their slot out of its result.
        sq = self._sq
        if sq is None:
            sq = self._sq = ([], [], asyncio.get_running_loop().create_future())
        slot = len(sq[0])
        sq[0].append(token_ids)
        sq[1].append(mask_id)
        if slot + 1 + len(self.queries) >= self.batch_size:
            if self.timer:
                self.timer.cancel()
                self.timer = None
            self.batch_evaluate_queries()
        else:
            loop = asyncio.get_running_loop()
            self._deadline = loop.time() + self.timeout
            if self.timer is None:
                self.timer = loop.call_later(self.timeout, self._on_timer)
        logZ, tok = await sq[2]
        return logZ[slot], tok[slot]

    async def gather(self, *coros, return_exceptions=False):
        """`asyncio.gather` for coroutines that spend their time awaiting THIS backend (a population of README particles:
        `await llm.gather(*[p.extend() for p in particles])`), without a Task per coroutine.  asyncio.gather wraps every
        coroutine in a Task: creating it, one trip through the event loop to start it, one to wake it when the shared
        future resolves and one done-callback cost CPython ~15 us per particle and step - at 1024 particles more than
        the step on the GPU takes.  Here the coroutines are advanced by hand (`send`): each runs to the point where it
        awaits this backend, the queued requests are evaluated as one batch, each is resumed with the result.  A
        coroutine that awaits something else (a foreign future, `asyncio.sleep`) is simply awaited from here, so any
        coroutine works - only slower.  Results in argument order; the first exception closes the other coroutines and
        propagates (with `return_exceptions` it takes the failed coroutine's place in the results, as in asyncio).
        No counterpart in the reference (its callers use asyncio.gather, README.md:96)."""
        n = len(coros)
        results = [None] * n
        live = list(range(n))   # coroutines that have not finished
        blocked = [None] * n    # what coroutine i is waiting for: a future, or None = ready to be resumed
        try:
            while live:
                nxt = []
                for i in live:
                    fut = blocked[i]
                    if fut is not None and not fut.done():
                        nxt.append(i)
                        continue
                    blocked[i] = None
                    try:
                        y = coros[i].send(None)
                    except StopIteration as e:
                        results[i] = e.value
                        continue
                    except Exception as e:
                        if not return_exceptions:
                            raise
                        results[i] = e
                        continue
                    if y is not None:  # `await future`: the future itself comes out (asyncio's protocol), flagged as awaited
                        if getattr(y, "_asyncio_future_blocking", None) is None:
                            raise RuntimeError(f"coroutine yielded {y!r}: not an asyncio awaitable")
                        y._asyncio_future_blocking = False
                        blocked[i] = y
                    nxt.append(i)
                live = nxt
                if not live:
                    break
                did_batch = bool(self.queries or self._sq is not None)
                if did_batch:
                    if self.timer:
                        self.timer.cancel()
                        self.timer = None
                    self.batch_evaluate_queries()  # everything queued so far is one batch: its futures resolve here
                waiting = [blocked[i] for i in live if blocked[i] is not None and not blocked[i].done()]
                if waiting and len(waiting) == len(live):
                    await asyncio.wait(waiting, return_when=asyncio.FIRST_COMPLETED)  # (foreign futures: the event loop's business)
                elif not did_batch:
                    # nothing of this pass was this backend's to resolve - coroutines polling with bare yields, possibly
                    # next to others blocked on foreign futures: the event loop (its timers, the futures' owners) gets its turn
                    await asyncio.sleep(0)
        except BaseException as exc:
            for i in live:
                coros[i].close()
            # what the closed coroutines had queued must not ride into somebody else's next batch: fail it (nobody is left
            # waiting on a future that would never resolve)
            self._fail_pending(exc if isinstance(exc, Exception) else asyncio.CancelledError())
            raise
        return results

    def _fail_pending(self, exc):
        """Drop everything queued and not yet evaluated; its futures get `exc`."""
        sq, self._sq = self._sq, None
        if sq is not None and not sq[2].done():
            sq[2].set_exception(exc)
            sq[2].exception()  # (marked retrieved: a gather that was abandoned has nobody left to read it)
        queries, self.queries = self.queries, []
        for q in queries:
            if q.future is not None and not q.future.done():
                q.future.set_exception(exc)
        if self.timer:
            self.timer.cancel()
            self.timer = None

    def _fire_step_batch(self):
        """Evaluate the pending `next_token_step` requests (one vectorised call) and resolve their shared future."""
        sq, self._sq = self._sq, None
        if sq is None:
            return
        contexts, mask_ids, future = sq
        try:
            logZ, tok = self.batch_next_token_step_sync(contexts, mask_ids)
        except Exception as e:  # vllm.py:396-400 behaviour: nobody is left waiting
            if not future.done():
                future.set_exception(e)
            return
        if not future.done():
            future.set_result((logZ.tolist(), tok.tolist()))

    # ---- batched submit: a whole population per call (no per-query futures, no per-query Python) ----------------------
    def _prefix_table(self):
        """Device table of the prompts whose KV `cache_kv` holds (tokens / starts / lengths for glb_match_prefixes and
        per-layer pointer tables for glb_gather_kv_padded); rebuilt when the set of cached prefixes changed."""
        entries = [(node, self._kv_tokens.get(id(node))) for node, _ in self._kv_lru._od.values()]
        entries = [(n, t) for n, t in entries if t is not None and n.past_key_values is not None]
        key = tuple(id(n) for n, _ in entries)
        if self._ptab is not None and self._ptab["key"] == key:
            return self._ptab
        if not entries:
            self._ptab = dict(key=key, n=0)
            return self._ptab
        dev = self.device
        kvs = [n.past_key_values for n, _ in entries]
        lens = np.array([len(t) for _, t in entries], np.int32)
        starts = np.zeros(len(entries), np.int64)
        starts[1:] = np.cumsum(lens[:-1])
        flat = np.concatenate([np.asarray(t, np.int32) for _, t in entries])
        ptrs = [[torch.tensor([kv.layers[l][j].data_ptr() for kv in kvs], dtype=torch.int64, device=dev)
                 for j in range(2)] for l in range(len(kvs[0].layers))]
        self._ptab = dict(key=key, n=len(entries), kvs=kvs, nodes=[n for n, _ in entries], tokens=torch.from_numpy(flat).to(dev),
                          starts=torch.from_numpy(starts).to(dev), lengths=torch.from_numpy(lens).to(dev), ptrs=ptrs,
                          p_max=int(lens.max()))
        return self._ptab

    @torch.no_grad()
    def batch_next_token_step_sync(self, contexts, mask_ids=None):
        """README.md:82-87 for a whole population in ONE call: for every context, logZ = logsumexp(next-token
        log-probs + mask[mask_id]) and a draw from the masked, renormalised distribution (-1 if nothing is allowed).
        Same pipeline as the queued `next_token_step` - dedup (hf.py:214-220), cached-prefix match (hf.py:334-342),
        ragged-to-padded gather, forward, lm_head on the last position, fused step - but the ragged batch is built
        once (three host arrays) and nothing is done per query in Python.  Returns (logZ float32 [n], token int32 [n])
        as NumPy arrays."""
        import itertools

        dev = self.device
        n = len(contexts)
        if n == 0:
            return np.zeros(0, np.float32), np.zeros(0, np.int32)
        lens = np.fromiter(map(len, contexts), np.int32, n)
        if int(lens.min()) == 0:
            raise ValueError("Token ids must not be empty")
        total = int(lens.sum())
        flat = np.fromiter(itertools.chain.from_iterable(contexts), np.int32, total)
        starts = np.zeros(n, np.int64)
        starts[1:] = np.cumsum(lens[:-1])
        mid_d = None
        if self._mask_kind != MASK_NONE:
            mid = np.zeros(n, np.int32) if mask_ids is None else np.ascontiguousarray(mask_ids, dtype=np.int32)
            mid_d = torch.from_numpy(mid).to(dev)
        logZ, tok = self._batch_step(torch.from_numpy(flat).to(dev), torch.from_numpy(starts).to(dev),
                                     torch.from_numpy(lens).to(dev), n, mid_d, l_max=int(lens.max()))
        out = torch.stack([logZ, tok.to(torch.float32)]).cpu().numpy()  # token ids < 2^24: exact in float32
        toks = out[1].astype(np.int32)
        self.engine.raise_if_failed(tokens=toks)  # a finishing wave of the one-launch step gave up waiting (include/glb.h)
        return out[0], toks

    @torch.no_grad()
    def batch_next_token_step_device(self, tokens, lengths, mask_ids=None):
        """`batch_next_token_step` for a population that LIVES ON THE DEVICE: `tokens` int32 [n, cap] (row i's first
        lengths[i] entries are context i - the padded particle matrix a device-resident loop keeps), `lengths` int32 [n],
        `mask_ids` int32 [n] or None, all device tensors.  Returns (logZ float32 [n], token int32 [n]) as device tensors:
        nothing of the population crosses the host in either direction (the call's only D2H copy is the handful of counts
        the host needs to size the forward; a failed launch shows as token -2 and raises at the next call / `engine.check()`).
        No counterpart in the reference (its API takes Python lists, hf.py:346-373); this is the same evaluation."""
        if tokens.dim() != 2 or tokens.dtype != torch.int32 or lengths.dtype != torch.int32 or lengths.numel() != tokens.shape[0]:
            raise ValueError("tokens must be int32 [n, cap] and lengths int32 [n]")
        if not tokens.is_contiguous():
            tokens = tokens.contiguous()
        n, cap = tokens.shape
        if n == 0:
            return (torch.zeros(0, dtype=torch.float32, device=self.device), torch.zeros(0, dtype=torch.int32, device=self.device))
        key = (n, cap)
        if getattr(self, "_row_starts", (None,))[0] != key:
            self._row_starts = (key, torch.arange(n, device=self.device, dtype=torch.int64) * cap)
        mid_d = None
        if self._mask_kind != MASK_NONE:
            mid_d = torch.zeros(n, dtype=torch.int32, device=self.device) if mask_ids is None else mask_ids
        return self._batch_step(tokens.view(-1), self._row_starts[1], lengths, n, mid_d, l_max=None)

    def _batch_step(self, tok_d, st_d, ln_d, n, mid_d, l_max):
        """The batched step on a ragged batch that is on the device already.  l_max: the longest context if the host
        knows it (else it rides on the call's D2H copy).  Returns device tensors."""
        eng, dev = self.engine, self.device
        group_of, rep, ng = eng.group_contexts(tok_d, st_d, ln_d)
        P = self._prefix_table()
        base, pref = None, None
        head = [ng[0]]
        used_d = None
        auto = self._auto_kv if (self._auto_kv is not None and not P["n"]) else None
        if P["n"]:
            pref, base = eng.match_prefixes(tok_d, st_d, ln_d, P["tokens"], P["starts"], P["lengths"])
            head.append((ln_d - base).max().to(torch.int32))
            # which cached prefixes this call uses (they count as recently used, like walk_cache's touch)
            used_d = torch.zeros(P["n"] + 1, dtype=torch.int32, device=dev).index_fill_(0, (pref + 1).long(), 1)[1:]
        elif l_max is None and auto is None:
            head.append(ln_d.max().to(torch.int32))
        if mid_d is not None:  # may the mask ids go per logits row? (they do when the mask is a function of the context)
            row_mid = mid_d[rep.long().clamp(0, n - 1)]  # entries of `rep` past the group count are unspecified
            head.append((row_mid[group_of.long()] == mid_d).all().to(torch.int32))
        if auto is not None:
            # contexts find the KV rows of their first L - 1 tokens (autokv.AutoKV): one token per context is fed; the error
            # word of the calls so far rides on the call's one D2H copy
            logits, row_of_group, group_of_row, U, extra = auto.logits(tok_d, st_d, ln_d, group_of, rep, ng,
                                                                       head[1:] + [eng.error_word()[0]])
            eng.raise_if_failed(extra[-1])
            by_row = bool(extra[-2]) if mid_d is not None else False
            if mid_d is not None:
                row_mid = row_mid[group_of_row]
            return self._finish_batch_step(logits, row_of_group[group_of.long()], group_of, U, n, mid_d,
                                           row_mid if mid_d is not None else None, by_row)
        n_head = len(head)
        head = torch.stack(head) if used_d is None else torch.cat([torch.stack(head), used_d])
        head = head.cpu().tolist()  # the call's one D2H copy before the forward
        if used_d is not None:
            for node, u in zip(P["nodes"], head[n_head:]):
                if u:
                    self._kv_lru.touch(node)
            head = head[:n_head]
        U = head[0]
        l_max = head[1] if (P["n"] or l_max is None) else l_max
        by_row = bool(head[-1]) if mid_d is not None else False
        p_max = P["p_max"] if P["n"] else 0
        pad_id = getattr(self.tokenizer, "pad_token_id", None) if self.tokenizer is not None else None
        ids, am, pos, last = eng.gather_padded(tok_d, st_d, ln_d, rep, U, base, 0 if pad_id is None else pad_id, p_max, l_max)
        cache = None
        if P["n"]:
            from transformers import DynamicCache

            kv0 = P["kvs"][0]
            pref_u = pref[rep[:U].long()].contiguous()
            data = [tuple(eng.gather_kv_padded(P["ptrs"][l][j], P["lengths"], pref_u, kv0.heads, kv0.head_dim, p_max,
                                               kv0.dtype) for j in range(2)) for l in range(len(kv0.layers))]
            cache = DynamicCache(ddp_cache_data=data)
        hidden = self._body(input_ids=ids, attention_mask=am, position_ids=pos, past_key_values=cache,
                            use_cache=cache is not None).last_hidden_state
        logits = self._lm_head(hidden[torch.arange(U, device=dev), last.long()])  # [U, V]
        return self._finish_batch_step(logits, group_of, group_of, U, n, mid_d, row_mid if mid_d is not None else None, by_row)

    def _finish_batch_step(self, logits, row_of, group_of, U, n, mid_d, row_mid, by_row):
        """The fused step over a batch's logits rows (row_of: logits row of every context; group_of: its dedup group -
        the order parity-mode noise is dealt in).  Returns (logZ, token) device tensors."""
        eng, dev = self.engine, self.device
        V = logits.shape[-1]
        kw = self.step_masks(logits.dtype)
        if kw:
            if by_row:
                kw["row_mask_id"] = row_mid[:U].contiguous()
            else:
                kw["mask_id"] = mid_d
        if self._rng_mode == RNG_NOISE:
            # Exp(1) rows in the reference's resolution order: by dedup group, duplicates contiguous (hf.py:285-288)
            order = torch.argsort(group_of.to(torch.int64), stable=True)
            slot = torch.empty(n, dtype=torch.int32, device=dev)
            slot[order] = torch.arange(n, dtype=torch.int32, device=dev)
            kw["noise"] = self._noise_rows(V).rows(n, row_slot=slot)
        logZ, _, tok = eng.step(logits, vocab=V, row_of=row_of, rng_mode=self._rng_mode, seed=self._rng_seed,
                                offset=self._batch_counter, want_lse=False, **kw)
        self._batch_counter += 1
        self.stats["batches"] += 1
        self.stats["queries"] += n
        self.stats["unique"] += U
        self.stats["rows"] += U
        return logZ, tok

    async def batch_next_token_step(self, contexts, mask_ids=None):
        """Awaitable form of `batch_next_token_step_sync` (the evaluation itself blocks the loop, like
        `batch_evaluate_queries` does in the reference, hf.py:307-308)."""
        return self.batch_next_token_step_sync(contexts, mask_ids)

    @torch.no_grad()
    def _batch_logprobs(self, token_ids_list):
        """`batch_next_token_logprobs` without one coroutine / future per context: cache walk per context, ONE batched
        evaluation of the misses, trie update, rows stacked on the device."""

        class _Slot:
            __slots__ = ("value",)

            def __init__(self):
                self.value = None

            def done(self):
                return self.value is not None

            def set_result(self, v):
                self.value = v

            def set_exception(self, e):
                raise e

        pending, refs = [], [None] * len(token_ids_list)  # refs[i]: (tensor, index) of context i's row (held: the budget may evict)
        for i, token_ids in enumerate(token_ids_list):
            if not token_ids:
                raise ValueError("Token ids must not be empty")
            node, nti, past, base = self.walk_cache(token_ids)
            if nti == len(token_ids):
                refs[i] = node.row_ref()
            else:
                pending.append((i, node, nti, base, Query(token_ids[base:], _Slot(), past, first_new=nti - base)))
        if pending:
            self._evaluate([q for *_, q in pending])
            for i, node, nti, base, q in pending:
                slab, row0, first = q.future.value
                refs[i] = node.extend_cache_lazy(nti, token_ids_list[i], slab, base + first - row0, store=self._rows)
        # rows of one slab leave in one gather
        by_slab = {}
        for i, (t, idx) in enumerate(refs):
            ent = by_slab.get(id(t))
            if ent is None:
                ent = by_slab[id(t)] = (t, [], [])
            ent[1].append(i)
            ent[2].append(idx)
        dev = self.device
        out = None
        for t, pos, idx in by_slab.values():
            if idx[0] < 0:  # a row of its own
                got = t[None].expand(len(pos), -1)
            else:
                got = t.index_select(0, torch.from_numpy(np.asarray(idx, np.int64)).to(dev))
            if len(by_slab) == 1:
                return got  # positions 0 .. n-1 in order
            if out is None:
                out = torch.empty((len(refs), got.shape[-1]), dtype=got.dtype, device=dev)
            out[torch.from_numpy(np.asarray(pos, np.int64)).to(dev)] = got
        return out

    async def batch_next_token_logprobs(self, token_ids_list):
        """base.py:47-60 (one batched evaluation instead of a gather over per-context coroutines)"""
        return self._batch_logprobs(token_ids_list)

    def batch_next_token_logprobs_sync(self, token_ids_list):
        """base.py:62-73"""
        return self._batch_logprobs(token_ids_list)

    # ---- sampling (base.py:110-179): a device-resident multi-token loop -----------------------------------------------
    @torch.no_grad()
    def batch_sample_sync(self, prompt_token_ids_list, max_tokens, eos_token_ids, temperature=1.0, seed=None,
                          sync_every=4):
        """base.py:148-179.  All sequences advance together on the device (sis.DeviceSampler): per-sequence KV slabs,
        softmax(logits / temperature) -> one draw per sequence per forward by the fused step, stop on `eos_token_ids`.
        With a seed every sequence reproduces torch.multinomial under its own torch.Generator().manual_seed(seed)
        (base.py:125-141); without one the draws come from the in-kernel Philox stream.  `sync_every`: how often the
        loop reads the number of unfinished sequences back (the reference reads every token of every sequence)."""
        from .sis import DeviceSampler

        if not prompt_token_ids_list:
            return []
        if any(len(p) == 0 for p in prompt_token_ids_list):
            raise ValueError("Token ids must not be empty")
        if max_tokens <= 0:
            return [[] for _ in prompt_token_ids_list]
        smp = DeviceSampler(self, prompt_token_ids_list, max_tokens, eos_token_ids, temperature, seed, sync_every=sync_every)
        return [[int(t) for t in row] for row in smp.generate()]

    async def batch_sample(self, prompt_token_ids_list, max_tokens, eos_token_ids, temperature=1.0, seed=None):
        """base.py:148-179"""
        return self.batch_sample_sync(prompt_token_ids_list, max_tokens, eos_token_ids, temperature, seed)

    async def sample(self, prompt_token_ids, max_tokens, eos_token_ids, temperature=1.0, seed=None):
        """base.py:110-146"""
        return self.batch_sample_sync([prompt_token_ids], max_tokens, eos_token_ids, temperature, seed)[0]


def load_model_by_name(name, backend=None, llm_opts=None):
    """llm/__init__.py:10-43.  `backend` may be None / "amd" / "hf" (all served by AsyncAmdLM on the
    MI355X); the CUDA / Apple engines of the reference ("vllm", "sgl", "mlx") do not exist here."""
    if llm_opts is None:
        llm_opts = {}
    if backend in (None, "amd", "hf"):
        return AsyncAmdLM.from_name(name, **llm_opts)
    if backend in ("vllm", "sgl", "mlx", "mock"):
        raise ValueError(f"Backend {backend!r} is not part of the MI355X hot-path build")
    raise ValueError(f"Invalid backend: {backend}")
