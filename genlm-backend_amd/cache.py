"""Token-id trie caching next-token log-probabilities (device rows) and prefix KV slabs.

Counterpart of the reference's `TokenTrie` (genlm/backend/cache.py:47-100).  Differences, all on
purpose: log-probability rows are float32 *device* tensors produced by the HIP log-softmax kernel
(one batched launch per evaluated batch instead of one torch.log_softmax + D2H copy per position),
and a node's `past_key_values` is a `KVPrefix` (contiguous per-layer K/V slabs the gather kernel can
address by pointer) rather than a transformers cache object.
"""
import torch


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

    __slots__ = ("children", "logprobs", "past_key_values")

    def __init__(self, parent=None, logprobs=None):
        self.children = {}
        self.logprobs = logprobs
        self.past_key_values = None

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

    def add_token(self, token_id, logprobs=None):
        # like the reference (cache.py:86-88) an existing child is replaced
        node = TokenTrie(self, logprobs)
        self.children[token_id] = node
        return node

    def extend_cache_rows(self, next_token_index, token_ids, logprob_rows, first_row_index):
        """Create nodes for token_ids[next_token_index:]; logprob_rows[j - first_row_index] is the
        already normalised row for position j."""
        node = self
        for j in range(next_token_index, len(token_ids)):
            node = node.add_token(token_ids[j], logprob_rows[j - first_row_index])
        return node

    def extend_cache(self, next_token_index, token_ids, logits, base, engine=None):
        """cache.py:90-100 signature: `logits[j - base]` are raw logits of position j; they are
        normalised by the HIP log-softmax kernel in one launch."""
        if engine is None:
            raise RuntimeError("TokenTrie.extend_cache needs the HIP engine that normalises the rows")
        lo = next_token_index - base
        rows = engine.log_softmax_rows(logits[lo:len(token_ids) - base])
        return self.extend_cache_rows(next_token_index, token_ids, rows, next_token_index)
