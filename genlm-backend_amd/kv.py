"""Device-resident KV state of the hot path.

* `SlabKV` — per-particle KV in preallocated slabs `[n, heads, cap, head_dim]`, one pair per layer, handed to the
  HuggingFace forward as a `Cache`.  The reference keeps KV as per-query tuples that are zero-padded and concatenated
  for every batch (hf.py:33-53,247-271) or as per-token slices on trie nodes (cache.py:103-191, mlx.py:177-318); here a
  particle owns row i of every slab, the new token's K/V is written in place at its own position (glb_kv_append:
  ragged lengths, no torch.cat regrow) and a resampling step is one gather launch over all layers
  (glb_kv_gather_rows) into the second slab set.
* `PrefixLRU` — byte-budgeted least-recently-used store for the prompt prefixes `cache_kv` pins (hf.py:155-164);
  the eviction policy of cache.py:103-191 (`DynamicTokenTrie`), applied to whole prefix slabs.
"""
from collections import OrderedDict

import torch
from transformers.cache_utils import Cache, CacheLayerMixin


# ---- the attention kernels of the path behind transformers' attention interface -----------------------------------------
# Registered under one name; AsyncAmdLM points the configuration of its SHADOW of the model (fuse.shadow_model - never the
# caller's own configuration) at it.  Two forwards of the hot path are served by this library's kernels - the in-place
# one-token forward over KV slab rows (glb_slab_attention: append fused, no mask tensor) and the padded batches of short
# contexts (glb_short_attention: a dozen tokens per row, where the library SDPA kernels spend a fifth of the step) - and
# everything else (long sequences, dropout, training / autograd, exotic masks, CPU runs) goes to transformers' own SDPA
# path exactly as before.  No module-level state: the engine rides on the shadow configuration (`_glb_engine`), the slab
# layer of a running in-place forward on the key tensor `Cache.update` returned (`_glb_layer`).
_ATTN_NAME = "glb"
SHORT_ATTENTION_MAX = 4096  # q_len * k_len up to which glb_short_attention is used


def _glb_attention_forward(module, query, key, value, attention_mask, dropout=0.0, scaling=None, **kwargs):
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    layer = getattr(key, "_glb_layer", None)
    if layer is not None and layer._new_k is not None:
        # an in-place forward whose append is this kernel's job: the new token's K / V are only in the stash
        if key is not layer.keys or value is not layer.values or query.shape[2] != 1:
            raise RuntimeError("glb_slab_attention: the attention module changed the key / value tensors the KV slab handed "
                               "out (or feeds more than one token): the new token was never appended - run this model with "
                               "SlabForward(fused_attention=False)")
        k_new, v_new, layer._new_k, layer._new_v = layer._new_k, layer._new_v, None, None
        o = layer.owner
        scale = scaling if scaling is not None else query.shape[-1] ** -0.5
        return o.engine.slab_attention(query, k_new, v_new, layer.keys, layer.values, o.pos, scale), None
    eng = getattr(getattr(module, "config", None), "_glb_engine", None)
    Lq, Lk = query.shape[2], key.shape[2]
    # the kernels write into fresh tensors outside autograd: a forward that may be differentiated keeps the SDPA path
    wants_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
    if (eng is not None and query.is_cuda and not dropout and not wants_grad and not getattr(module, "training", False)
            and Lq * Lk <= SHORT_ATTENTION_MAX and Lk >= Lq
            and eng.slab_attention_supports(query.dtype, query.shape[-1]) and key.dtype == query.dtype == value.dtype
            and query.stride(3) == 1 and key.stride(3) == 1 and value.stride(3) == 1
            and kwargs.get("is_causal", True) is not False and not kwargs.get("softcap") and not kwargs.get("sliding_window")
            and getattr(module, "is_causal", True)
            and (attention_mask is None or (attention_mask.dtype == torch.bool and attention_mask.dim() == 4
                                            and attention_mask.shape[1] == 1 and attention_mask.shape[-1] == Lk
                                            and attention_mask.stride(3) == 1))
            and all((st * query.element_size()) % 16 == 0 for t in (query, key, value) for st in t.stride()[:3])
            and all(t.data_ptr() % 16 == 0 for t in (query, key, value))):
        scale = scaling if scaling is not None else query.shape[-1] ** -0.5
        return eng.short_attention(query, key, value, attention_mask, scale), None
    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, **kwargs)


def _register_attention():
    from transformers.masking_utils import ALL_MASK_ATTENTION_FUNCTIONS
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS

    if _ATTN_NAME not in ALL_ATTENTION_FUNCTIONS:
        ALL_ATTENTION_FUNCTIONS.register(_ATTN_NAME, _glb_attention_forward)
        ALL_MASK_ATTENTION_FUNCTIONS.register(_ATTN_NAME, ALL_MASK_ATTENTION_FUNCTIONS["sdpa"])


def use_glb_attention(shadow, engine):
    """Point a SHADOW of a transformers model (fuse.shadow_model: its configuration is a private copy) whose attention goes
    through the attention interface as "sdpa" at this library's attention kernels; returns whether it was done."""
    cfg = getattr(shadow, "config", None)
    if cfg is not None and getattr(cfg, "_attn_implementation", None) == _ATTN_NAME:
        return getattr(cfg, "_glb_engine", None) is engine
    if (cfg is None or not hasattr(engine, "short_attention") or getattr(cfg, "_attn_implementation", None) != "sdpa"
            or getattr(cfg, "attn_logit_softcapping", None) or getattr(cfg, "sliding_window", None)
            or getattr(cfg, "scale_attn_by_inverse_layer_idx", False)):
        return False
    _register_attention()
    cfg._attn_implementation = _ATTN_NAME
    cfg._glb_engine = engine
    return True


class _SlabLayer(CacheLayerMixin):
    """One layer's K / V slabs.  `update` appends one token per row at `owner.pos` and returns the whole slabs; which
    positions a row may attend to is the 2-D attention mask's business (`SlabKV.attention_mask`)."""

    is_compileable = False
    is_sliding = False

    def __init__(self, owner, idx):
        super().__init__()
        self.owner, self.idx = owner, idx
        self._new_k = self._new_v = None

    def lazy_initialization(self, key_states, value_states):
        o = self.owner
        shape_k = (o.n, key_states.shape[1], o.cap, key_states.shape[-1])
        shape_v = (o.n, value_states.shape[1], o.cap, value_states.shape[-1])
        self.keys = torch.zeros(shape_k, dtype=key_states.dtype, device=key_states.device)
        self.values = torch.zeros(shape_v, dtype=value_states.dtype, device=value_states.device)
        self.is_initialized = True

    def update(self, key_states, value_states, *args, **kwargs):
        if not self.is_initialized:
            self.lazy_initialization(key_states, value_states)
        o = self.owner
        if key_states.shape[0] != o.n or key_states.shape[-2] != 1:
            raise ValueError("SlabKV takes one new token for every particle per forward")
        if o.fused_attention:  # glb_slab_attention appends: it gets the new K / V where the projection left them
            self._new_k, self._new_v = key_states, value_states
            self.keys._glb_layer = self  # (how the attention entry finds this layer: no module-level state)
            return self.keys, self.values
        o.engine.kv_append(self.keys, key_states, o.pos)
        o.engine.kv_append(self.values, value_states, o.pos)
        return self.keys, self.values

    def get_mask_sizes(self, query_length):
        return self.owner.cap, 0

    def get_seq_length(self):
        return self.owner.cap - 1  # the query sits "after" every slot: causality is expressed by the 2-D mask alone

    def get_max_length(self):
        return self.owner.cap


class SlabKV(Cache):
    def __init__(self, engine, n, cap, n_layers):
        self.engine, self.n, self.cap = engine, n, cap
        self.pos = None  # int32 [n] device: where this forward's token goes (= tokens already held by the row)
        self.fused_attention = False  # the running forward's attention appends (SlabForward sets it per call)
        super().__init__(layers=[_SlabLayer(self, i) for i in range(n_layers)])
        self._alt = None
        self._ptrs = None

    # ---- forward-side helpers ----------------------------------------------------------------------------------
    def set_forward_in_place(self, pos):
        """pos: int32 [n] device - where every row's token of the next forward goes."""
        self.pos = pos

    def attention_mask(self, pos):
        """[n, cap] 0/1: row i sees its `pos[i]` cached tokens and the one being appended at `pos[i]`."""
        ar = torch.arange(self.cap, device=pos.device, dtype=pos.dtype)
        return (ar[None, :] <= pos[:, None]).to(torch.int64)

    # ---- slab plumbing ---------------------------------------------------------------------------------------------
    def _alloc_like(self, prompt_layers):
        for layer, (k, v) in zip(self.layers, prompt_layers):
            if not layer.is_initialized:
                layer.lazy_initialization(k[:1], v[:1])

    def _tensors(self, which):
        return [t for layer in which for t in (layer.keys, layer.values)]

    def fill_rows(self, src_layers, src_row_of, len_of):
        """Rows i with src_row_of[i] >= 0 take the first len_of[i] positions of row src_row_of[i] of `src_layers`
        ([(K, V)] per layer, each [u, heads, l_src, head_dim], contiguous): fan-out of freshly encoded contexts."""
        self._alloc_like(src_layers)
        srcs = [t for kv in src_layers for t in kv]
        self.engine.kv_gather_rows(srcs, self._tensors(self.layers), src_row_of, len_of, srcs_stable=False)

    def gather(self, src_row_of, len_of):
        """Resampling: row i becomes a copy of row src_row_of[i] (first len_of[i] positions); rows with
        src_row_of[i] < 0 are left to be refilled by the caller.  Works into the second slab set, then swaps."""
        cur = self._tensors(self.layers)
        if self._alt is None:
            self._alt = [torch.zeros_like(t) for t in cur]
        self.engine.kv_gather_rows(cur, self._alt, src_row_of, len_of)
        for j, layer in enumerate(self.layers):
            layer.keys, self._alt[2 * j] = self._alt[2 * j], layer.keys
            layer.values, self._alt[2 * j + 1] = self._alt[2 * j + 1], layer.values

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._tensors(self.layers) if t is not None) * (2 if self._alt else 1)


class _SharedLayer(_SlabLayer):
    """One layer of `SharedSlabKV`: the forward's batch row u lives in slab row `owner.rows[u]`; attention reads the
    forward's rows from the staging slabs `owner.stage()` filled them into."""

    def lazy_initialization(self, key_states, value_states):
        o = self.owner
        for name, st in (("keys", key_states), ("values", value_states)):
            shape = (o.n, st.shape[1], o.cap, st.shape[-1])
            setattr(self, name, torch.zeros(shape, dtype=st.dtype, device=st.device))
            setattr(self, "stage_" + name, torch.zeros(shape, dtype=st.dtype, device=st.device))
        self.is_initialized = True

    def update(self, key_states, value_states, *args, **kwargs):
        o = self.owner
        eng = o.engine
        if o.in_place:  # the forward's batch IS the slab: row b's token goes to row b, attention reads the slabs
            if key_states.shape[0] != o.n or key_states.shape[-2] != 1:
                raise ValueError("an in-place forward takes one token for every slab row")
            if o.fused_attention:
                self._new_k, self._new_v = key_states, value_states
                self.keys._glb_layer = self
                return self.keys, self.values
            eng.kv_append(self.keys, key_states, o.pos)
            eng.kv_append(self.values, value_states, o.pos)
            return self.keys, self.values
        U = o.rows.numel()
        if key_states.shape[0] != U or key_states.shape[-2] != 1:
            raise ValueError("SharedSlabKV takes one new token for every row of the forward")
        # the new token's K / V: into the row that keeps it, and beside the gathered prefix the attention reads
        eng.kv_append(self.keys, key_states, o.pos, rows=o.rows)
        eng.kv_append(self.values, value_states, o.pos, rows=o.rows)
        eng.kv_append(self.stage_keys, key_states, o.pos, rows=o.ident[:U])
        eng.kv_append(self.stage_values, value_states, o.pos, rows=o.ident[:U])
        return self.stage_keys[:U], self.stage_values[:U]


class SharedSlabKV(SlabKV):
    """KV rows SHARED between particles (the reference's per-token KV on trie nodes, cache.py:103-191,
    restated for a population): `n` slab rows of `cap` positions, a block table outside (DeviceSIS.row_of) that maps
    particles to rows.  Particles with the same context point to ONE row and one forward row serves them all; when a
    shared row's particles draw different tokens, one keeps the row and the others get a copy of the prefix into free
    rows (copy-on-append, one gather launch over all layers); resampling re-points particles to their ancestors' rows
    and copies nothing.  A forward names its batch's rows (`set_forward`): one gather launch brings their prefixes - all
    layers, K and V - into staging slabs in batch order (the PyTorch attention wants a dense [U, H, cap, Dh]), the new
    token's K / V go into the row that keeps them and beside the gathered prefix."""

    def __init__(self, engine, n_rows, cap, n_layers):
        Cache.__init__(self, layers=[_SharedLayer(self, i) for i in range(n_layers)])
        self.engine, self.n, self.cap = engine, n_rows, cap
        self.pos = None
        self.fused_attention = False
        self.rows = None   # int32 [U]: slab row of every forward row
        self.ident = None  # int32 arange(n)
        self.in_place = False
        self._alt = None
        self._ptrs = None

    def set_forward(self, rows, pos):
        """rows, pos: int32 [U] device.  Gathers the U prefixes (pos[u] positions of row rows[u]) into the staging slabs."""
        self.rows, self.pos, self.in_place = rows, pos, False
        U = rows.numel()
        if self.ident is None:
            self.ident = torch.arange(self.n, dtype=torch.int32, device=rows.device)
        src_row_of = torch.full((self.n,), -1, dtype=torch.int32, device=rows.device)
        src_row_of[:U] = rows
        len_of = torch.zeros(self.n, dtype=torch.int32, device=rows.device)
        len_of[:U] = pos
        slabs = self._tensors(self.layers)
        stage = [t for layer in self.layers for t in (layer.stage_keys, layer.stage_values)]
        self.engine.kv_gather_rows(slabs, stage, src_row_of, len_of)

    def set_forward_in_place(self, pos):
        """The next forward runs on ALL slab rows where they lie (batch row b = slab row b, pos: int32 [n] device, the
        position row b's token is appended at): no gather.  Worth it when most rows are live - a free row costs a
        forward row whose result nobody reads, the gather costs every live row's whole prefix, read and written."""
        self.rows, self.pos, self.in_place = None, pos, True

    def copy_rows(self, src_row_of, len_of):
        """Row r with src_row_of[r] >= 0 takes the first len_of[r] positions of row src_row_of[r] (in place: sources
        are live rows, destinations free ones)."""
        cur = self._tensors(self.layers)
        self.engine.kv_gather_rows(cur, cur, src_row_of, len_of)

    def nbytes(self):
        return 2 * sum(t.numel() * t.element_size() for t in self._tensors(self.layers) if t is not None)


class SlabForward:
    """The one-token forward over ALL rows of a `SlabKV` / `SharedSlabKV` where they lie (`set_forward_in_place`).  Its
    shapes never change - batch = the slab's rows, one token each, the mask a function of the positions - so after two
    eager calls the launch sequence (the transformer body's kernels and this library's glb_kv_append launches between
    them) is captured into a hipGraph once and replayed from static input buffers: same kernels, same bits, without
    the host walking the model's Python for every step (GPT-2-small, 1024 rows: 4.1 -> 3.4 ms; Llama-3.2-1B shape, 512
    rows: 10.4 -> 7.2 ms).  A graph belongs to the slab tensors it was captured over (a resampling `gather` swaps slab
    sets: one graph each).  `graph=False` (or a CPU device) keeps every call eager."""

    def __init__(self, pkv, body, graph=True, fused_attention=True, owner=None):
        self.pkv, self.body = pkv, body
        self.owner = owner  # the AsyncAmdLM whose `refresh_weights()` drops the captured graphs (its weights_epoch moves)
        self._owner_epoch = getattr(owner, "weights_epoch", 0)
        self.graph_ok = bool(graph) and torch.cuda.is_available()
        self.calls = 0
        self.graphs = {}  # identity of the slab set -> (graph, ids, pos, hidden, the slab tensors themselves)
        # weights the shadow keeps derived copies of (fuse._merged_qkv): a captured graph reads the copy it was captured
        # with, so the graphs are dropped when one of the sources changes
        self._watched = [t for mod in body.modules() if hasattr(mod, "q_proj") and hasattr(mod, "k_proj") and hasattr(mod, "v_proj")
                         for t in (mod.q_proj.weight, mod.k_proj.weight, mod.v_proj.weight)]
        self._watched += [t for mod in body.modules() if hasattr(mod, "gate_proj") and hasattr(mod, "up_proj")
                          for t in (mod.gate_proj.weight, mod.up_proj.weight)]
        self._watched_version = self._weights_version()
        # glb_slab_attention instead of two appends + a mask + a dense SDPA call per layer: for models whose attention
        # goes through transformers' attention interface with plain softmax(q k^T * scale) v semantics on a HIP device
        cfg = getattr(body, "config", None)
        k0 = pkv.layers[0].keys if pkv.layers and getattr(pkv.layers[0], "is_initialized", False) else None
        # (the body is AsyncAmdLM's shadow of the model, already pointed at the "glb" attention entry when glb_attention is on;
        # a body that is not - the caller's own model, glb_attention=False - keeps appends + mask + SDPA)
        self.fused = bool(fused_attention and cfg is not None and k0 is not None and k0.is_cuda
                          and hasattr(pkv.engine, "slab_attention")
                          and pkv.engine.slab_attention_supports(k0.dtype, k0.shape[-1])
                          and getattr(cfg, "_attn_implementation", None) == _ATTN_NAME
                          and getattr(cfg, "_glb_engine", None) is pkv.engine)

    def _weights_version(self):
        return sum(t._version for t in self._watched)

    def _run(self, ids, pos):
        pkv = self.pkv
        pkv.set_forward_in_place(pos)
        if self.fused:
            pkv.fused_attention = True
            try:  # (no mask: the kernel attends to positions 0 .. pos[r] and nothing else)
                out = self.body(input_ids=ids, position_ids=pos.view(-1, 1).long(), attention_mask=None,
                                past_key_values=pkv, use_cache=True)
            finally:
                pkv.fused_attention = False
                stale = [i for i, ly in enumerate(pkv.layers) if ly._new_k is not None]
                for i in stale:
                    pkv.layers[i]._new_k = pkv.layers[i]._new_v = None
            if stale:  # an attention module that bypassed the attention interface: its token was never appended
                raise RuntimeError(f"glb_slab_attention was not reached in layers {stale}: the new token's K / V were not "
                                   "appended - run this model with SlabForward(fused_attention=False)")
            return out.last_hidden_state[:, 0]
        out = self.body(input_ids=ids, position_ids=pos.view(-1, 1).long(), attention_mask=pkv.attention_mask(pos),
                        past_key_values=pkv, use_cache=True)
        return out.last_hidden_state[:, 0]

    @torch.no_grad()
    def __call__(self, ids, pos):
        """ids: int64 [R, 1], pos: int32 [R] (device).  Returns the last hidden states [R, d] (valid until the next call)."""
        self.calls += 1
        if not self.graph_ok or not ids.is_cuda or self.calls <= 2:  # (allocator growth, library set-up: outside a capture)
            return self._run(ids, pos)
        if self._watched:
            ver = self._weights_version()
            if ver != self._watched_version:  # the caller changed a weight in place: captured launches would read stale copies
                self.graphs.clear()
                self._watched_version = ver
        ep = getattr(self.owner, "weights_epoch", 0)
        if ep != self._owner_epoch:  # AsyncAmdLM.refresh_weights(): writes the version counters cannot see (param.data)
            self.graphs.clear()
            self._owner_epoch = ep
        # a graph belongs to the slab tensors it was captured over: every layer's addresses, shape and dtype make the key,
        # and the entry holds the tensors, so the allocator cannot hand their addresses to another slab set meanwhile
        slabs = self.pkv._tensors(self.pkv.layers)
        key = tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in slabs)
        ent = self.graphs.get(key)
        if ent is None:
            if len(self.graphs) >= 2:  # slab sets keep changing under this forward: not worth capturing
                return self._run(ids, pos)
            s_ids, s_pos = ids.clone(), pos.clone()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (a collective's watchdog thread may be about)
                    hidden = self._run(s_ids, s_pos)
            except RuntimeError as e:
                # a forward that cannot be captured (a host-side decision on device data, a synchronising call): HIP says
                # so in a RuntimeError that names the capture - eager from now on.  Anything else (shapes, memory, an
                # assertion in kv_append) is a real error of the forward and is the caller's to see.
                msg = str(e).lower()
                if not any(w in msg for w in ("captur", "graph", "hipstreamsynchronize", "cudastreamsynchronize",
                                              "operation not permitted")):
                    raise
                self.graph_ok = False
                self.capture_error = e
                torch.cuda.synchronize()
                return self._run(ids, pos)
            ent = self.graphs[key] = (g, s_ids, s_pos, hidden, slabs)
        g, s_ids, s_pos, hidden, _held = ent
        s_ids.copy_(ids)
        s_pos.copy_(pos)
        self.pkv.set_forward_in_place(s_pos)  # (the captured launches read the static buffers)
        g.replay()
        return hidden


class PrefixLRU:
    """Least-recently-used store of cached prompt prefixes under a byte budget.  Keys are trie nodes; evicting an entry
    drops the node's `past_key_values` (the log-prob rows stay), so later queries fall back to re-encoding."""

    def __init__(self, budget_bytes, on_remove=None):
        self.budget = int(budget_bytes)
        self.used = 0
        self._od = OrderedDict()
        self.evictions = 0
        self.on_remove = on_remove  # called with the node whenever an entry leaves the store (evicted or dropped)

    def put(self, node, kv):
        self.drop(node)
        size = kv.nbytes()
        node.past_key_values = kv
        self._od[id(node)] = (node, size)
        self.used += size
        while self.used > self.budget and len(self._od) > 1:
            _, (old, sz) = self._od.popitem(last=False)
            old.past_key_values = None
            self.used -= sz
            self.evictions += 1
            if self.on_remove is not None:
                self.on_remove(old)

    def touch(self, node):
        key = id(node)
        if key in self._od:
            self._od.move_to_end(key)

    def drop(self, node):
        ent = self._od.pop(id(node), None)
        if ent is not None:
            self.used -= ent[1]
            ent[0].past_key_values = None
            if self.on_remove is not None:
                self.on_remove(ent[0])

    def clear(self):
        nodes = [node for node, _ in self._od.values()]
        for node in nodes:
            node.past_key_values = None
        self._od.clear()
        self.used = 0
        if self.on_remove is not None:
            for node in nodes:
                self.on_remove(node)

    def nodes(self):
        return [node for node, _ in self._od.values()]

    def __len__(self):
        return len(self._od)
