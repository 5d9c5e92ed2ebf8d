"""A private view of the caller's HuggingFace model for this backend's forwards.

The reference leaves the model it is handed alone (hf.py:114-140).  Earlier rounds rewrote activation modules of
`hf_model` and re-pointed `hf_model.config._attn_implementation` in place; now `AsyncAmdLM` runs its forwards on a
**shadow** of the module tree: every module is a shallow copy (`copy.copy`) that SHARES its `_parameters` / `_buffers`
dictionaries with the original - the same weights, also after the caller replaces them (`model.to(...)`,
`load_state_dict`) - but has its own `_modules` table and its own copy of the configuration.  Whatever is swapped below
happens in the shadow only; the caller's model, its configuration and its forward are never touched, so two backends over
one model, or a foreign thread running `hf_model(...)`, see exactly the model they built.

What the shadow swaps (PyTorch ops only - no kernels of this library inside the transformer body except attention):
  * GPT-2's `gelu_new` spelled as eight elementwise ops      -> `torch.nn.GELU(approximate="tanh")`     (one kernel)
  * Llama-family RMSNorm spelled as pow / mean / add / rsqrt / mul / two casts / mul (hf modeling_llama.py:52-67)
                                                             -> `torch.nn.functional.rms_norm`          (one kernel)
  * Llama-family rotary embedding spelled as slice / neg / cat / mul / mul / add per tensor (modeling_llama.py:130-160)
                                                             -> roll + mul + addcmul, queries and keys in one pass
  * the q / k / v projections (three GEMMs)                  -> one GEMM on a derived [q; k; v] weight (a copy of those
                                                                rows, rebuilt when the caller's weights change)
  * Llama's gate_proj + up_proj, for few-token forwards       -> one GEMM on a derived [gate; up] weight (a second copy of
                                                                those two matrices: `merge_mlp=False` keeps the memory)
  * the attention interface                                  -> kv.py's "glb" entry (glb_short_attention /
                                                                glb_slab_attention where they apply, SDPA otherwise)
Same functions, different rounding (float32 inside the fused ops, one rounding at the end): the reference's goldens hold
within 1e-4 with identical tokens (tests/test_host_cpu.py, tests/test_host_gpu.py).
"""
import copy
import types

import torch

RMSNORM_CLASSES = ("LlamaRMSNorm", "MistralRMSNorm", "Qwen2RMSNorm", "Qwen3RMSNorm")  # x * rsqrt(mean(x^2) + eps) * weight
ROPE_ATTENTION_CLASSES = ("LlamaAttention",)


def shadow_model(model):
    """Shallow structural copy of a module tree: new module objects and `_modules` tables, shared parameter / buffer
    tables, one private copy of the configuration (modules that referenced the original configuration reference the copy)."""
    cfg = getattr(model, "config", None)
    new_cfg = copy.copy(cfg) if cfg is not None else None
    memo = {}

    def walk(mod):
        got = memo.get(id(mod))
        if got is not None:
            return got
        new = copy.copy(mod)
        memo[id(mod)] = new
        new._modules = {name: (walk(child) if child is not None else None) for name, child in mod._modules.items()}
        if cfg is not None and new.__dict__.get("config") is cfg:
            new.__dict__["config"] = new_cfg
        return new

    return walk(model)


class FusedRMSNorm(torch.nn.Module):
    """`weight * (x * rsqrt(mean(x^2) + eps))` as one `rms_norm` call; shares the original module's parameter table."""

    def __init__(self, src):
        super().__init__()
        self._parameters = src._parameters
        self.eps = float(src.variance_epsilon)

    def forward(self, hidden_states):
        return torch.nn.functional.rms_norm(hidden_states, (hidden_states.shape[-1],), self.weight, self.eps)

    def extra_repr(self):
        return f"{tuple(self.weight.shape)}, eps={self.eps} (fused)"


def _rope(x, rolled, cos, sin_signed):
    """x, rolled [B, T, H, D] (the projection's own layout; rolled = x with the halves of D swapped), cos / sin_signed
    [B, T, 1, D].  x * cos + rotate_half(x) * sin with rotate_half(x) * sin = roll(x, D/2) * (sin with its first half
    negated): three kernels (roll, mul, addcmul) instead of six (slice, neg, cat, mul, mul, add)."""
    return torch.addcmul(x * cos, rolled, sin_signed)


def _merged_qkv(attn):
    """One weight [q; k; v] for the three projections (one GEMM instead of three, two of them a quarter as wide, and the
    rotary embedding applied to queries and keys in ONE pass over the joint output).  A derived copy of the caller's
    weights (q + k + v rows: 12.6 MB a layer at Llama-3.2-1B's shape), rebuilt whenever one of them is replaced or changed
    in place (`Tensor._version`)."""
    ws = (attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight)
    bs = (attn.q_proj.bias, attn.k_proj.bias, attn.v_proj.bias)
    key = tuple((id(t), t._version, t.data_ptr()) for t in ws + bs if t is not None)
    ent = attn.__dict__.get("_glb_qkv")
    if ent is None or ent[0] != key:
        with torch.no_grad():
            w = torch.cat(ws, 0)
            b = torch.cat(bs, 0) if bs[0] is not None else None
        # (the entry holds the source tensors: their ids cannot be handed to other tensors while it lives)
        ent = attn.__dict__["_glb_qkv"] = (key, w, b, ws + bs)
    return ent[1], ent[2]


MERGE_MLP_MAX_TOKENS = 2048  # up to here one GEMM for gate + up wins (tools/dbg/gemm_merge_probe.py); beyond, two do


def _merged_gate_up(mlp):
    """One weight [gate; up] for a SwiGLU MLP's two input projections, as `_merged_qkv`: a derived copy (at Llama-3.2-1B's
    shape 67 MB a layer), rebuilt when the caller's weights change."""
    ws = (mlp.gate_proj.weight, mlp.up_proj.weight)
    bs = (mlp.gate_proj.bias, mlp.up_proj.bias)
    key = tuple((id(t), t._version, t.data_ptr()) for t in ws + bs if t is not None)
    ent = mlp.__dict__.get("_glb_gate_up")
    if ent is None or ent[0] != key:
        with torch.no_grad():
            w = torch.cat(ws, 0)
            b = torch.cat(bs, 0) if bs[0] is not None else None
        ent = mlp.__dict__["_glb_gate_up"] = (key, w, b, ws + bs)
    return ent[1], ent[2]


def _llama_mlp_forward(self, x):
    """modeling_llama.py:174-176.  For the few-token forwards of the path (one token per KV row: M = 512 to 1024) gate_proj and
    up_proj are ONE GEMM on a derived [gate; up] weight - at M = 512 two GEMMs of N = 8192 leave three quarters of the chip's
    CUs without a tile (65.6 -> 48.3 us a layer at Llama-3.2-1B's shape, 135 -> 114 at Llama-3-8B's); the big re-encoding
    batches keep two GEMMs (the strided halves cost the elementwise ops more than the GEMM gains there)."""
    if (x.shape[:-1].numel() <= MERGE_MLP_MAX_TOKENS
            and not (torch.is_grad_enabled() and (x.requires_grad or self.gate_proj.weight.requires_grad))):
        w, b = _merged_gate_up(self)
        n = self.gate_proj.weight.shape[0]
        h = torch.nn.functional.linear(x, w, b)
        return self.down_proj(self.act_fn(h[..., :n]) * h[..., n:])
    return self.down_proj(self.act_fn(self.gate_proj(x)) * self.up_proj(x))


def weights_version(net):
    """A number that changes when a weight the shadow keeps a derived copy of changes (SlabForward drops its hipGraphs then:
    a captured launch would keep reading the stale copy)."""
    v = 0
    for mod in net.modules():
        ent = mod.__dict__.get("_glb_qkv")
        if ent is not None:
            for t in (mod.q_proj.weight, mod.k_proj.weight, mod.v_proj.weight):
                v += t._version + (id(t) & 0xFFFF)
    return v


def _llama_attention_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None,
                             **kwargs):
    """modeling_llama.py:243-281 with q / k / v from one GEMM and the rotary embedding applied to queries and keys together,
    before the head transpose (see `_rope`).  A forward that may be differentiated runs the module's own code."""
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    from transformers.models.llama.modeling_llama import eager_attention_forward

    if torch.is_grad_enabled() and (hidden_states.requires_grad or self.q_proj.weight.requires_grad):
        return type(self).forward(self, hidden_states, position_embeddings=position_embeddings, attention_mask=attention_mask,
                                  past_key_values=past_key_values, **kwargs)
    input_shape = hidden_states.shape[:-1]
    D = self.head_dim
    w, b = _merged_qkv(self)
    qkv = torch.nn.functional.linear(hidden_states, w, b).view(*input_shape, -1, D)  # [B, T, Hq + 2 Hkv, D]
    n_kv = self.k_proj.weight.shape[0] // D
    n_q = qkv.shape[-2] - 2 * n_kv
    cos, sin = position_embeddings
    ss = getattr(sin, "_glb_signed", None)
    half = D // 2
    if ss is None:
        ss = torch.cat((-sin[..., :half], sin[..., half:]), dim=-1)
    rolled = torch.roll(qkv, half, -1)  # (the whole joint tensor: a roll of a strided slice would copy it first)
    qk = _rope(qkv[..., :n_q + n_kv, :], rolled[..., :n_q + n_kv, :], cos.unsqueeze(2), ss.unsqueeze(2))
    q = qk[..., :n_q, :].transpose(1, 2)
    k = qk[..., n_q:, :].transpose(1, 2)
    v = qkv[..., n_q + n_kv:, :].transpose(1, 2)
    if past_key_values is not None:
        k, v = past_key_values.update(k, v, self.layer_idx)
    attention_interface = ALL_ATTENTION_FUNCTIONS.get_interface(self.config._attn_implementation, eager_attention_forward)
    attn_output, attn_weights = attention_interface(self, q, k, v, attention_mask,
                                                    dropout=0.0 if not self.training else self.attention_dropout,
                                                    scaling=self.scaling, **kwargs)
    attn_output = attn_output.reshape(*input_shape, -1).contiguous()
    return self.o_proj(attn_output), attn_weights


def _rotary_forward_signed(self, x, position_ids):
    """The rotary module's own forward, its sine tagged with the half-negated copy `_rope` multiplies the rolled tensor
    by (made once per forward instead of once per layer)."""
    cos, sin = type(self).forward(self, x, position_ids)
    half = sin.shape[-1] // 2
    sin._glb_signed = torch.cat((-sin[..., :half], sin[..., half:]), dim=-1)
    return cos, sin


def fuse_shadow(shadow, activations=True, merge_mlp=True):
    """Swap the decomposed activations / norms / rotary embedding of a SHADOW tree (never call this on a caller's model).
    Returns the names of what was swapped."""
    done = []
    if not activations:
        return done
    for mod in list(shadow.modules()):
        for name, child in list(mod._modules.items()):
            kind = type(child).__name__
            if kind == "NewGELUActivation":
                mod._modules[name] = torch.nn.GELU(approximate="tanh")
                done.append("gelu_new")
            elif kind in RMSNORM_CLASSES and hasattr(child, "variance_epsilon") and "weight" in child._parameters:
                mod._modules[name] = FusedRMSNorm(child)
                done.append("rms_norm")
            elif kind in ROPE_ATTENTION_CLASSES:
                child.forward = types.MethodType(_llama_attention_forward, child)
                done.append("rope")
            elif kind == "LlamaRotaryEmbedding":
                child.forward = types.MethodType(_rotary_forward_signed, child)
            elif kind == "LlamaMLP" and merge_mlp:
                child.forward = types.MethodType(_llama_mlp_forward, child)
                done.append("gate_up")
    return done
