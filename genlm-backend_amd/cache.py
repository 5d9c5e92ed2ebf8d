"""Token-id trie caching next-token log-probabilities (device rows) and prefix KV slabs.

Counterpart of the reference's `TokenTrie` (genlm/backend/cache.py:47-100).  Differences, all on
purpose: log-probability rows are float32 *device* tensors produced by the HIP log-softmax kernel
(one batched launch per evaluated batch instead of one torch.log_softmax + D2H copy per position),
and a node's `past_key_values` is a `KVPrefix` (contiguous per-layer K/V slabs the gather kernel can
address by pointer) rather than a transformers cache object.
"""
from collections import OrderedDict

import torch


class RowLRU:
    """The log-probability rows the trie holds, under a byte budget.  Rows are views into the [R, V] slab their batch's
    log-softmax launch wrote, so the unit of eviction is a slab: least recently used first, every node that points
    into it loses its row (`logprobs = None`: a later request for that context is a miss and is evaluated again), and
    the slab goes back to the allocator once no caller holds a view either.  The reference keeps CPU copies without a
    bound (cache.py:93-98; its bounded store, OutputCache, cache.py:6-44, serves the vLLM adapter only)."""

    def __init__(self, budget_bytes):
        self.budget = int(budget_bytes)
        self.used = 0
        self.evictions = 0
        self._od = OrderedDict()  # slab key -> [bytes, nodes]

    @staticmethod
    def _key(rows):
        st = rows.untyped_storage()
        return st.data_ptr(), st.nbytes()

    def add(self, rows, nodes):
        if not nodes:
            return
        key, nbytes = self._key(rows)
        ent = self._od.get(key)
        if ent is None:
            ent = self._od[key] = [nbytes, []]
            self.used += nbytes
        ent[1].extend(nodes)
        for nd in nodes:
            nd.slab = key
        self._od.move_to_end(key)
        while self.used > self.budget and len(self._od) > 1:
            _, (nb, old) = self._od.popitem(last=False)
            for nd in old:
                if nd.slab is not None:
                    nd._rows = None
                    nd.slab = None
            self.used -= nb
            self.evictions += 1

    def touch(self, node):
        if node.slab is not None and node.slab in self._od:
            self._od.move_to_end(node.slab)

    def clear(self):
        self._od.clear()
        self.used = 0

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
    """cache.py:47-100.  `logprobs` is the next-token log-probability row after the path to this node."""

    __slots__ = ("children", "_rows", "_idx", "past_key_values", "slab")

    def __init__(self, parent=None, logprobs=None, idx=-1):
        self.children = {}
        # the row is kept as (tensor, index) and cut out on access: a batch hands over thousands of rows of one slab,
        # and a torch view per row costs more than the node itself
        self._rows = logprobs
        self._idx = idx
        self.past_key_values = None
        self.slab = None  # RowLRU key of the slab `logprobs` points into

    @property
    def logprobs(self):
        r = self._rows
        if r is None or self._idx < 0:
            return r
        return r[self._idx]

    @logprobs.setter
    def logprobs(self, value):
        self._rows = value
        self._idx = -1

    def __repr__(self):
        inner = ", ".join(f"{t}: {n!r}" for t, n in self.children.items())
        return f"{'*' if self.past_key_values is not None else ''}[{inner}]"

    def clear_kv_cache(self):
        self.past_key_values = None
        for node in self.children.values():
            node.clear_kv_cache()

    def has_token(self, token_id):
        return token_id in self.children

    def get_token(self, token_id):
        return self.children[token_id]

    def add_token(self, token_id, logprobs=None, idx=-1):
        # like the reference (cache.py:86-88) an existing child is replaced - unless it only lost its row to the byte
        # budget (RowLRU): then it gets the row back and keeps what hangs below it.  idx >= 0: the row is logprobs[idx].
        old = self.children.get(token_id)
        if old is not None and old._rows is None:
            old._rows, old._idx = logprobs, idx
            return old
        node = TokenTrie(self, logprobs, idx)
        self.children[token_id] = node
        return node

    def extend_cache_rows(self, next_token_index, token_ids, logprob_rows, first_row_index, store=None):
        """Create nodes for token_ids[next_token_index:]; logprob_rows[j - first_row_index] is the
        already normalised row for position j.  `store` (RowLRU): the rows are accounted under its byte budget."""
        node = self
        made = []
        for j in range(next_token_index, len(token_ids)):
            node = node.add_token(token_ids[j], logprob_rows, j - first_row_index)
            made.append(node)
        if store is not None and made:
            store.add(logprob_rows, made)
        return node

    def extend_cache(self, next_token_index, token_ids, logits, base, engine=None, store=None):
        """cache.py:90-100 signature: `logits[j - base]` are raw logits of position j; they are
        normalised by the HIP log-softmax kernel in one launch."""
        if engine is None:
            raise RuntimeError("TokenTrie.extend_cache needs the HIP engine that normalises the rows")
        lo = next_token_index - base
        rows = engine.log_softmax_rows(logits[lo:len(token_ids) - base])
        return self.extend_cache_rows(next_token_index, token_ids, rows, next_token_index, store=store)
