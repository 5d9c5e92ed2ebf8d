"""Token-id trie caching next-token log-probabilities (device rows) and prefix KV slabs.

Counterpart of the reference's `TokenTrie` (genlm/backend/cache.py:47-100).  Differences, all on
purpose: log-probability rows are float32 *device* tensors produced by the HIP log-softmax kernel
(one batched launch per evaluated batch instead of one torch.log_softmax + D2H copy per position),
and a node's `past_key_values` is a `KVPrefix` (contiguous per-layer K/V slabs the gather kernel can
address by pointer) rather than a transformers cache object.
"""
from collections import OrderedDict


class _Slab:
    """Handle of one [R, V] slab of log-probability rows: the trie's nodes point at the handle, the budget empties it."""

    __slots__ = ("t", "key")

    def __init__(self, t, key):
        self.t = t
        self.key = key


class RowLRU:
    """The log-probability rows the trie holds, under a byte budget.  Rows are (slab, index) pairs into the [R, V] slab
    their batch's log-softmax launch wrote, so the unit of eviction is a slab: least recently used first; its handle is
    emptied, every node that points at it has lost its row (`logprobs` is None: a later request for that context is a
    miss and is evaluated again), and the slab goes back to the allocator once no caller holds a view either.  The
    reference keeps CPU copies without a bound (cache.py:93-98; its bounded store, OutputCache, cache.py:6-44, serves
    the vLLM adapter only)."""

    def __init__(self, budget_bytes):
        self.budget = int(budget_bytes)
        self.used = 0
        self.evictions = 0
        self._od = OrderedDict()  # storage key -> (bytes, [handles of the views handed over])
        self._last = (None, None)  # a batch hands over the same slab a thousand times

    def add(self, rows):
        """The handle of `rows` (a slab, or a view into one: views of one storage are accounted once and leave together);
        a new storage is accounted, older ones leave if the budget says so."""
        if self._last[0] is rows:
            return self._last[1]
        st = rows.untyped_storage()
        key = (st.data_ptr(), st.nbytes())
        ent = self._od.get(key)
        if ent is None:
            ent = self._od[key] = (key[1], [])
            self.used += key[1]
        h = next((x for x in ent[1] if x.t is rows), None)
        if h is None:
            h = _Slab(rows, key)
            ent[1].append(h)
        self._od.move_to_end(key)
        while self.used > self.budget and len(self._od) > 1:
            _, (nb, old) = self._od.popitem(last=False)
            for x in old:
                x.t = None
            self.used -= nb
            self.evictions += 1
        self._last = (rows, h)
        return h

    def touch(self, node):
        r = node._rows
        if type(r) is _Slab and r.t is not None and r.key in self._od:
            self._od.move_to_end(r.key)

    def clear(self):
        for _, hs in self._od.values():
            for h in hs:
                h.t = None
        self._od.clear()
        self.used = 0
        self._last = (None, None)

    def __len__(self):
        return len(self._od)


class KVPrefix:
    """Key/value states of one cached prompt: per layer a (K, V) pair of contiguous [H, P, Dh] tensors."""

    def __init__(self, layers):
        self.layers = [(k.contiguous(), v.contiguous()) for k, v in layers]
        k0 = self.layers[0][0]
        self.heads, self.length, self.head_dim = k0.shape
        self.dtype = k0.dtype

    @classmethod
    def from_hf_cache(cls, cache, batch_index=0):
        """transformers>=5 DynamicCache (layers[i].keys / .values are [B, H, P, Dh])."""
        return cls([(l.keys[batch_index], l.values[batch_index]) for l in cache.layers])

    def __len__(self):
        return self.length

    def nbytes(self):
        return sum(k.numel() * k.element_size() + v.numel() * v.element_size() for k, v in self.layers)


class TokenTrie:
    """cache.py:47-100.  `logprobs` is the next-token log-probability row after the path to this node.

    A batch that materialises every position of a thousand new contexts would make ten thousand nodes per step; the
    nodes below the first new one are made when somebody looks (`children`): until then the node carries the rest of the
    path as `_tail` = (tokens, position, rows, index of the first row)."""

    __slots__ = ("_kids", "_rows", "_idx", "past_key_values", "_tail")

    def __init__(self, parent=None, logprobs=None, idx=-1):
        self._kids = {}
        # the row is kept as (tensor or slab handle, index) and cut out on access: a batch hands over thousands of rows
        # of one slab, and a torch view per row costs more than the node itself
        self._rows = logprobs
        self._idx = idx
        self.past_key_values = None
        self._tail = None

    @property
    def children(self):
        if self._tail is not None:
            toks, p, rows, idx = self._tail
            self._tail = None
            child = TokenTrie(self, rows, idx)
            if p + 1 < len(toks):
                child._tail = (toks, p + 1, rows, idx + 1)
            self._kids[toks[p]] = child
        return self._kids

    def row_ref(self):
        """(tensor, index) of this node's row - index -1: the tensor is the row - or None if it has none (any more)."""
        r = self._rows
        if type(r) is _Slab:
            r = r.t
            if r is None:  # the byte budget took the slab
                self._rows = None
        return None if r is None else (r, self._idx)

    def has_row(self):
        return self.row_ref() is not None

    @property
    def logprobs(self):
        ref = self.row_ref()
        if ref is None:
            return None
        return ref[0] if ref[1] < 0 else ref[0][ref[1]]

    @logprobs.setter
    def logprobs(self, value):
        self._rows = value
        self._idx = -1

    def __repr__(self):
        inner = ", ".join(f"{t}: {n!r}" for t, n in self.children.items())
        return f"{'*' if self.past_key_values is not None else ''}[{inner}]"

    def clear_kv_cache(self):
        self.past_key_values = None
        for node in self._kids.values():  # nodes not made yet hold no KV
            node.clear_kv_cache()

    def has_token(self, token_id):
        return token_id in self.children

    def get_token(self, token_id):
        return self.children[token_id]

    def add_token(self, token_id, logprobs=None, idx=-1):
        # like the reference (cache.py:86-88) an existing child is replaced - unless it only lost its row to the byte
        # budget (RowLRU): then it gets the row back and keeps what hangs below it.  idx >= 0: the row is logprobs[idx].
        kids = self.children
        old = kids.get(token_id)
        if old is not None and not old.has_row():
            old._rows, old._idx = logprobs, idx
            return old
        node = TokenTrie(self, logprobs, idx)
        kids[token_id] = node
        return node

    def extend_cache_rows(self, next_token_index, token_ids, logprob_rows, first_row_index, store=None):
        """Create nodes for token_ids[next_token_index:]; logprob_rows[j - first_row_index] is the already normalised row
        for position j.  `store` (RowLRU): the rows are accounted under its byte budget.  Returns the last node."""
        rows = store.add(logprob_rows) if store is not None and next_token_index < len(token_ids) else logprob_rows
        node = self
        for j in range(next_token_index, len(token_ids)):
            node = node.add_token(token_ids[j], rows, j - first_row_index)
        return node

    def extend_cache_lazy(self, next_token_index, token_ids, logprob_rows, first_row_index, store=None):
        """`extend_cache_rows` that makes the first new node only and leaves the path below it as that node's `_tail`
        (made when a walk gets there).  Returns (tensor, index) of the LAST position's row - what the caller came for."""
        n = len(token_ids)
        if next_token_index >= n:
            return self.row_ref()
        rows = store.add(logprob_rows) if store is not None else logprob_rows
        node = self
        j = next_token_index
        while j < n:
            node = node.add_token(token_ids[j], rows, j - first_row_index)
            j += 1
            if j < n and not node._kids and node._tail is None:  # a fresh node: the rest waits
                node._tail = (tuple(token_ids[j:]), 0, rows, j - first_row_index)
                break
        return logprob_rows, n - 1 - first_row_index

    def extend_cache(self, next_token_index, token_ids, logits, base, engine=None, store=None, out_dtype=None):
        """cache.py:90-100 signature: `logits[j - base]` are raw logits of position j; they are
        normalised by the HIP log-softmax kernel in one launch (`out_dtype`: float32 by default, or the logits' own)."""
        if engine is None:
            raise RuntimeError("TokenTrie.extend_cache needs the HIP engine that normalises the rows")
        lo = next_token_index - base
        kw = {} if out_dtype is None else {"out_dtype": out_dtype}
        rows = engine.log_softmax_rows(logits[lo:len(token_ids) - base], **kw)
        engine.check()  # (cache_kv is a set-up call: one synchronising look at the error word; no NaN row reaches the trie)
        return self.extend_cache_rows(next_token_index, token_ids, rows, next_token_index, store=store)
