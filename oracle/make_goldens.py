"""Generates tests/golden/*.npz by running the REFERENCE itself (genlm/genlm-backend, imported
read-only from /root/reference) and torch-CPU on seeded inputs.  The fixtures are data only (inputs
and expected outputs); this script is the committed recipe that made them.

Run (in the build container, never on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python -B /root/repo/oracle/make_goldens.py

Container-only shims needed to import / run the reference with transformers 5.x and no numba
(SURVEY.md §8c): a stub `numba` module (the hot path never calls it), `DynamicCache.from_legacy_cache`
re-added, KV stored as legacy tuples for `cache_kv`, and `decode_vocab` patched (no tokenizer files
exist offline).  None of this code ships; nothing under genlm-backend_amd/ imports it.
"""
import asyncio
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ------------------------------------------------------------------------------------------ shims
def install_shims():
    class _Any:  # numba.float64[:] etc. appear in annotations of code the hot path never calls
        def __getitem__(self, k):
            return self

        def __call__(self, *a, **k):
            return self

    class _NumbaStub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Any()

    nb = _NumbaStub("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    nb.njit = nb.jit
    typed = types.ModuleType("numba.typed")
    typed.List = list
    nb.typed = typed
    sys.modules["numba"] = nb
    sys.modules["numba.typed"] = typed
    for name in ("vllm", "sglang", "mlx", "mlx_lm"):
        pass  # the reference guards these imports itself
    from transformers import DynamicCache

    if not hasattr(DynamicCache, "from_legacy_cache"):
        @classmethod
        def from_legacy_cache(cls, past):
            if past is None:
                return cls()
            return cls(ddp_cache_data=[(k, v) for k, v in past])

        DynamicCache.from_legacy_cache = from_legacy_cache


class FakeTokenizer:
    pad_token_id = None
    eos_token_id = 0
    is_fast = False
    name_or_path = "synthetic"

    def __len__(self):
        return 0


TINY = dict(vocab_size=1000, n_positions=64, n_embd=64, n_layer=2, n_head=4, bos_token_id=0, eos_token_id=0)


def tiny_model(seed=0):
    from transformers import GPT2Config, GPT2LMHeadModel

    torch.manual_seed(seed)
    return GPT2LMHeadModel(GPT2Config(**TINY)).eval()


def reference_llm(model, batch_size=64, timeout=0.02):
    import genlm.backend.llm.base as base

    base.decode_vocab = lambda tok: ([], [])
    from genlm.backend.llm.hf import AsyncTransformer

    return AsyncTransformer(model, FakeTokenizer(), batch_size=batch_size, timeout=timeout)


def legacy_cache_kv(llm, prompt_tokens):
    """hf.py:155-164 with the KV kept as the legacy tuple structure Query.__init__ subscripts."""
    with torch.no_grad():
        result = llm.model(torch.tensor([prompt_tokens]).to(llm.device))
    node = llm.cache.extend_cache(0, prompt_tokens, result.logits[0], 0)
    node.past_key_values = tuple((l.keys, l.values) for l in result.past_key_values.layers)


# ------------------------------------------------------------------------------------------ README SIS
def make_masks(V, max_token_length_mod=3):
    eos = 0
    valid = torch.tensor([(i == eos) or (i % max_token_length_mod != 1) for i in range(V)], dtype=torch.float).log()
    eos_one_hot = torch.nn.functional.one_hot(torch.tensor(eos), V).log()
    return valid, eos_one_hot


class Particle:  # README.md:72-91, verbatim semantics
    def __init__(self, llm, mask_function, prompt_ids):
        self.context = []
        self.prompt_ids = prompt_ids
        self.log_weight = 0.0
        self.active = True
        self.llm = llm
        self.mask_function = mask_function

    async def extend(self):
        logps = await self.llm.next_token_logprobs(self.prompt_ids + self.context)
        masked_logps = logps + self.mask_function(self.context).to(logps.device)
        logZ = masked_logps.logsumexp(dim=-1)
        self.log_weight += logZ
        next_token_id = torch.multinomial((masked_logps - logZ).exp(), 1).item()
        if next_token_id == self.llm.tokenizer.eos_token_id:
            self.active = False
        else:
            self.context.append(next_token_id)


async def autobatched_sis(n_particles, llm, masking_function, prompt_ids):  # README.md:94-98
    particles = [Particle(llm, masking_function, prompt_ids) for _ in range(n_particles)]
    steps = 0
    while any(p.active for p in particles):
        await asyncio.gather(*[p.extend() for p in particles if p.active])
        steps += 1
    return particles, steps


def golden_hotpath():
    model = tiny_model(0)
    V = TINY["vocab_size"]
    out = {"config_json": np.frombuffer(repr(TINY).encode(), dtype=np.uint8)}
    for k, v in model.state_dict().items():
        out["w::" + k] = v.numpy()

    # (1) batched async == what the reference returns, incl. a duplicate and ragged lengths (test_hf_llm.py:17-42)
    prompts = [[5, 17, 250, 3, 77, 901], [44, 8, 19], [5, 17, 250, 3, 77, 901], [600, 2, 2, 9, 31, 7, 7, 12], [999]]
    llm = reference_llm(model)
    lps = asyncio.run(llm.batch_next_token_logprobs(prompts))
    out["lp_prompts"] = np.array([p + [-1] * (8 - len(p)) for p in prompts], np.int32)
    out["lp_values"] = lps.numpy()
    unc = torch.stack([llm.next_token_logprobs_uncached(p) for p in prompts])
    out["lp_uncached"] = unc.numpy()

    # (2) README SIS loop, config 1 of BASELINE.json scaled to the tiny model: 16 particles, prompt len 8,
    #     <= 10 tokens, torch.manual_seed(1234)
    llm = reference_llm(model, batch_size=64)
    valid, eos1 = make_masks(V)
    max_tokens = 10
    prompt = list(range(100, 108))
    torch.manual_seed(1234)
    parts, steps = asyncio.run(autobatched_sis(16, llm, lambda c: eos1 if len(c) >= max_tokens else valid, prompt))
    ctx = np.full((16, max_tokens), -1, np.int32)
    for i, p in enumerate(parts):
        ctx[i, :len(p.context)] = p.context
    out["sis_prompt"] = np.array(prompt, np.int32)
    out["sis_contexts"] = ctx
    out["sis_log_weights"] = np.array([float(p.log_weight) for p in parts], np.float32)
    out["sis_steps"] = np.array([steps], np.int32)
    out["sis_masks"] = torch.stack([valid, eos1]).numpy()
    lw = torch.tensor([float(p.log_weight) for p in parts])
    out["sis_probs"] = torch.exp(lw - lw.logsumexp(dim=-1)).numpy()  # README.md:108-110

    # (3) prefix KV cache: cache_kv(prompt) then queries extending it + one unrelated (hf.py:155-164,334-342)
    llm = reference_llm(model)
    pre = [7, 8, 9, 10, 11]
    legacy_cache_kv(llm, pre)
    qs = [pre + [20], pre + [21, 22], [300, 301, 302], pre + [20]]
    kv = asyncio.run(llm.batch_next_token_logprobs(qs))
    out["kv_prefix"] = np.array(pre, np.int32)
    out["kv_queries"] = np.array([q + [-1] * (8 - len(q)) for q in qs], np.int32)
    out["kv_values"] = kv.numpy()
    out["kv_uncached"] = torch.stack([llm.next_token_logprobs_uncached(q) for q in qs]).numpy()

    # (4) seeded sampling (base.py:110-146): self-consistency in the reference's tests (test_hf_llm.py:260-283);
    #     here the actual ids are pinned
    llm = reference_llm(model)
    ids = asyncio.run(llm.sample([3, 1, 4, 1, 5], max_tokens=12, eos_token_ids=[0], temperature=0.5, seed=80808))
    out["sample_prompt"] = np.array([3, 1, 4, 1, 5], np.int32)
    out["sample_ids"] = np.array(ids, np.int32)
    ids2 = asyncio.run(llm.batch_sample([[3, 1, 4, 1, 5], [9, 9]], max_tokens=6, eos_token_ids=[], temperature=1.0, seed=7))
    out["batch_sample_ids"] = np.array(ids2, np.int32)

    np.savez_compressed(os.path.join(OUT, "ref_hotpath_tiny.npz"), **out)
    print("ref_hotpath_tiny.npz:", {k: v.shape for k, v in out.items() if not k.startswith("w::")})


# ------------------------------------------------------------------------------------------ kernel-level
def golden_kernel():
    """torch-CPU results of the exact op sequence the reference runs per particle (cache.py:96,
    README.md:84-87) on the build's synthetic logits (tests/synth.py).  Only small summaries are kept."""
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from tests import synth

    out = {}
    for tag, B, V, dt in [("gpt2_f32", 32, 50257, torch.float32), ("llama_bf16", 16, 128256, torch.bfloat16),
                          ("small_f16", 8, 4099, torch.float16)]:
        x = torch.from_numpy(synth.logits(11, B, V)).to(dt)
        masks = torch.from_numpy(synth.binary_masks(11, 2, V))
        mid = torch.arange(B) % 2
        logps = torch.log_softmax(x, -1)                       # cache.py:96 (dtype follows the logits)
        lp32 = torch.log_softmax(x.float(), -1)
        masked = logps.float() + masks[mid]                    # README.md:84 (fp32 upcast for 16-bit logits)
        logZ = masked.logsumexp(-1)                            # README.md:85
        g = torch.Generator()
        g.manual_seed(1234)
        tok = torch.multinomial((masked - logZ[:, None]).exp(), 1, generator=g).flatten()  # README.md:87
        out[f"{tag}::shape"] = np.array([B, V], np.int64)
        out[f"{tag}::logZ"] = logZ.numpy()
        out[f"{tag}::token"] = tok.numpy().astype(np.int32)
        out[f"{tag}::lse32"] = torch.logsumexp(x.float(), -1).numpy()
        out[f"{tag}::lp32_head"] = lp32[:, :64].numpy()
        out[f"{tag}::lp32_rowsum"] = lp32.double().sum(-1).numpy()
        # second-best margin of the exponential race, to document how far each draw is from a tie
        g.manual_seed(1234)
        q = torch.empty(B, V).exponential_(1, generator=g)
        r = (masked - logZ[:, None]).exp() / q
        top2 = r.topk(2, -1).values
        out[f"{tag}::race_margin"] = ((top2[:, 0] - top2[:, 1]) / top2[:, 0]).numpy()
    # torch's CPU exponential stream itself (the RNG contract of GLB_RNG_NOISE)
    g = torch.Generator()
    g.manual_seed(99)
    out["mt::seed99_first4096"] = torch.empty(4096).exponential_(1, generator=g).numpy()
    np.savez_compressed(os.path.join(OUT, "torch_kernel_ops.npz"), **out)
    print("torch_kernel_ops.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    golden_kernel()
    golden_hotpath()
