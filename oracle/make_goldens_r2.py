"""Round-2 fixtures, again produced by running the REFERENCE itself (read-only import from /root/reference) in the build
container: tests/golden/ref_round2.npz (+ two small tokenizer data files the tests rebuild their tokenizers from).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python -B /root/repo/oracle/make_goldens_r2.py

What is pinned:
  trie::*    TokenCharacterTrie (trie/base.py) on the reference's own known-answer vocabulary (tests/test_trie.py:19-85)
             and on a 600-token synthetic vocabulary: node numbering, leaves, prefixes, weight_sum / weight_max.  numba
             is absent here; the stub of make_goldens.install_shims makes `@numba.jit` the identity, so the two update
             loops (base.py:346-393) run as the plain Python they are.
  bv::*      get_byte_vocab (tokenization/bytes.py:15-57) on an in-memory byte-level BPE tokenizer (the default GPT-2
             byte-decoder path) and on a SentencePiece model trained in memory (the sp_model path).  The reference
             fetches the GPT-2 alphabet from the hub (`AutoTokenizer.from_pretrained("gpt2")`, bytes.py:222); offline
             that call is answered by an object carrying transformers' own `bytes_to_unicode` table.
  c3::*      BASELINE config 3 shape on the tiny GPT-2: K in {1, 8, 64} distinct ragged-length prompts shared by the
             particles, README SIS loop under torch.manual_seed: tokens, log-weights, unique contexts per step.  Run
             WITHOUT cache_kv for K > 1: the reference's dedup key is the token suffix after the cached prefix
             (hf.py:214-220, the ":216 XXX" note), so two particles of different cached prompts that generated the same
             tokens would be merged into one forward row; with one cached prompt (K = 1) cache_kv is exercised too.
  parity1024::*  torch-CPU log_softmax + mask + logsumexp + multinomial at the full headline size (1024 x 50257 fp32)
  llama::*   a tiny LlamaConfig (RoPE, grouped-query attention) through the reference's hf path: batched log-probs of
             ragged prompts, the uncached values, and a README SIS loop.
"""
import asyncio
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402

OUT = MG.OUT


# ------------------------------------------------------------------------------------------------ trie
def synthetic_vocab(n, seed):
    rs = np.random.default_rng(seed)
    alphabet = [b"a", b"b", b"c", b" ", b"\xc3", b"\xa9", b"e", b"t"]
    words, seen = [], set()
    while len(words) < n:
        L = int(rs.integers(1, 7))
        w = b"".join(alphabet[int(i)] for i in rs.integers(0, len(alphabet), L))
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


def golden_trie(out):
    from genlm.backend.tokenization import Token
    from genlm.backend.trie.base import TokenCharacterTrie

    cases = {"kat": [b"a", b"b", b"ab", b"<eos>"], "syn": synthetic_vocab(600, 3)}
    for tag, words in cases.items():
        decode = [Token(i, w) for i, w in enumerate(words)]
        trie = TokenCharacterTrie(decode=decode)
        n = len(trie.children)
        rs = np.random.default_rng(17)
        if tag == "kat":
            ws = np.array([[0.1, 0.2, 0.2, 0.5]], np.float32)  # tests/test_trie.py:28
        else:
            ws = rs.random((5, len(words))).astype(np.float32)
            ws[1] = 0.0
            ws[2, ::7] = 0.0
            ws /= np.maximum(ws.sum(-1, keepdims=True), 1e-30)
        out[f"trie::{tag}::words"] = np.frombuffer(b"\x00".join(words), dtype=np.uint8)
        out[f"trie::{tag}::ws"] = ws
        out[f"trie::{tag}::n_nodes"] = np.array([n], np.int64)
        out[f"trie::{tag}::root"] = np.array([trie.root], np.int64)
        out[f"trie::{tag}::idx_to_leaf"] = np.asarray(trie.idx_to_leaf, np.int32)
        # children as (parent, symbol or -1 - token index for a leaf edge, child) triples in dict order
        tri = []
        for x, ch in enumerate(trie.children):
            for sym, y in ch.items():
                tri.append((x, -1 - sym[1] if isinstance(sym, tuple) else int(sym), y))
        out[f"trie::{tag}::edges"] = np.array(tri, np.int64)
        out[f"trie::{tag}::prefix_len"] = np.array([len(trie.node2prefix[i]) for i in range(n)], np.int32)
        out[f"trie::{tag}::sum"] = np.stack([trie.weight_sum(torch.from_numpy(w)) for w in ws])
        out[f"trie::{tag}::max"] = np.stack([trie.weight_max(torch.from_numpy(w)) for w in ws])


# ------------------------------------------------------------------------------------------------ byte vocab
CORPUS = ["Montreal is a city in Canada.", "the quick brown fox jumps over the lazy dog", "naïve café ☕ 東京",
          "def f(x):\n    return x + 1\n", "’•¶∂ƒ˙∆£Ħ爨ൠᅘ∰፨"] * 20


def build_bpe_json():
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(CORPUS, trainers.BpeTrainer(vocab_size=420, special_tokens=["<|endoftext|>"],
                                                        initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                                        show_progress=False))
    return tok.to_str()


def build_spm_bytes():
    import sentencepiece as spm

    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(CORPUS), model_writer=buf, vocab_size=400, model_type="bpe",
                                   byte_fallback=True, character_coverage=0.9995, num_threads=1,
                                   user_defined_symbols=["<sep>"], minloglevel=2)
    return buf.getvalue()


class SpTokenizer:
    """The slice of a slow SentencePiece tokenizer get_byte_vocab reads: sp_model, get_added_vocab, __len__."""

    def __init__(self, model_bytes, added=("<extra_0>",)):
        import sentencepiece as spm

        self.sp_model = spm.SentencePieceProcessor(model_proto=model_bytes)
        n = self.sp_model.get_piece_size()
        self._added = {t: n + i for i, t in enumerate(added)}

    def get_added_vocab(self):
        return dict(self._added)

    def __len__(self):
        return self.sp_model.get_piece_size() + len(self._added)


def golden_byte_vocab(out):
    from tokenizers import Tokenizer
    from transformers import AutoTokenizer, PreTrainedTokenizerFast
    from transformers.convert_slow_tokenizer import bytes_to_unicode

    import genlm.backend.tokenization.bytes as RB

    class _Gpt2Alphabet:  # what bytes.py:222 needs from the hub copy of the gpt2 tokenizer
        byte_decoder = {c: b for b, c in bytes_to_unicode().items()}

    real = AutoTokenizer.from_pretrained
    RB.AutoTokenizer.from_pretrained = staticmethod(lambda name, **kw: _Gpt2Alphabet() if name == "gpt2" else real(name, **kw))
    bpe_json = build_bpe_json()
    fast = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(bpe_json), eos_token="<|endoftext|>")
    bv = RB.get_byte_vocab(fast)
    out["bv::bpe::lens"] = np.array([len(b) for b in bv], np.int32)
    out["bv::bpe::bytes"] = np.frombuffer(b"".join(bv), dtype=np.uint8)
    spm_bytes = build_spm_bytes()
    sp = SpTokenizer(spm_bytes)
    bv = RB.get_byte_vocab(sp)
    out["bv::spm::lens"] = np.array([len(b) for b in bv], np.int32)
    out["bv::spm::bytes"] = np.frombuffer(b"".join(bv), dtype=np.uint8)
    with open(os.path.join(OUT, "bpe_tokenizer.json"), "w") as f:
        f.write(bpe_json)
    with open(os.path.join(OUT, "spm_tiny.model"), "wb") as f:
        f.write(spm_bytes)


# ------------------------------------------------------------------------------------------------ config 3 + llama
def ragged_prompts(K, seed, V):
    rs = np.random.default_rng(seed)
    return [[int(t) for t in rs.integers(1, V, int(rs.integers(3, 9)))] for _ in range(K)]


class MultiParticle(MG.Particle):
    pass


async def sis_multi(llm, masking_function, prompts_per_particle):
    parts = [MG.Particle(llm, masking_function, p) for p in prompts_per_particle]
    steps, uniq = 0, []
    while any(p.active for p in parts):
        before = None
        await asyncio.gather(*[p.extend() for p in parts if p.active])
        steps += 1
    return parts, steps


def run_sis(model, prompts_per_particle, V, max_tokens, seed, cache_prompts=()):
    # batch_size above the population: the batch fires on the 20 ms timer and the particles resume (and draw) in the
    # order their futures were resolved - dedup-group order.  A batch that fills exactly lets the last coroutine run
    # on before the others (hf.py:307-308), a different draw order.
    llm = MG.reference_llm(model, batch_size=2 * len(prompts_per_particle))
    for p in cache_prompts:
        MG.legacy_cache_kv(llm, p)
    valid, eos1 = MG.make_masks(V)
    torch.manual_seed(seed)
    parts, steps = asyncio.run(sis_multi(llm, lambda c: eos1 if len(c) >= max_tokens else valid, prompts_per_particle))
    ctx = np.full((len(parts), max_tokens), -1, np.int32)
    for i, p in enumerate(parts):
        ctx[i, :len(p.context)] = p.context
    return ctx, np.array([float(p.log_weight) for p in parts], np.float32), steps


def golden_config3(out):
    model = MG.tiny_model(0)
    V = MG.TINY["vocab_size"]
    N, max_tokens = 64, 6
    for K in (1, 8, 64):
        prompts = ragged_prompts(K, 100 + K, V)
        per = [prompts[i % K] for i in range(N)]
        ctx, lw, steps = run_sis(model, per, V, max_tokens, seed=4321 + K)
        out[f"c3::K{K}::prompts"] = np.array([p + [-1] * (8 - len(p)) for p in prompts], np.int32)
        out[f"c3::K{K}::contexts"] = ctx
        out[f"c3::K{K}::log_weights"] = lw
        out[f"c3::K{K}::steps"] = np.array([steps], np.int32)
        if K == 1:  # with the prompt's KV cached (hf.py:155-164): one cached prefix, no cross-prefix suffix collisions
            ctx2, lw2, _ = run_sis(model, per, V, max_tokens, seed=4321 + K, cache_prompts=[prompts[0]])
            out["c3::K1::contexts_kv"] = ctx2
            out["c3::K1::log_weights_kv"] = lw2
    out["c3::masks"] = torch.stack(MG.make_masks(V)).numpy()


LLAMA_TINY = dict(vocab_size=500, hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=6,
                  num_key_value_heads=2, head_dim=8, max_position_embeddings=64, rope_theta=10000.0, rms_norm_eps=1e-5,
                  tie_word_embeddings=False, bos_token_id=0, eos_token_id=0, attention_bias=False, mlp_bias=False)


def golden_llama(out):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(5)
    model = LlamaForCausalLM(LlamaConfig(**LLAMA_TINY)).eval()
    out["llama::config_json"] = np.frombuffer(repr(LLAMA_TINY).encode(), dtype=np.uint8)
    for k, v in model.state_dict().items():
        out["llama::w::" + k] = v.numpy()
    V = LLAMA_TINY["vocab_size"]
    prompts = [[5, 17, 250, 3, 77, 401], [44, 8, 19], [5, 17, 250, 3, 77, 401], [300, 2, 2, 9, 31, 7, 7, 12], [499]]
    llm = MG.reference_llm(model)
    out["llama::lp_prompts"] = np.array([p + [-1] * (8 - len(p)) for p in prompts], np.int32)
    out["llama::lp_values"] = asyncio.run(llm.batch_next_token_logprobs(prompts)).numpy()
    out["llama::lp_uncached"] = torch.stack([llm.next_token_logprobs_uncached(p) for p in prompts]).numpy()
    prompts3 = ragged_prompts(3, 77, V)
    per = [prompts3[i % 3] for i in range(24)]
    ctx, lw, steps = run_sis(model, per, V, 6, seed=999)
    out["llama::sis_prompts"] = np.array([p + [-1] * (8 - len(p)) for p in prompts3], np.int32)
    out["llama::sis_contexts"] = ctx
    out["llama::sis_log_weights"] = lw
    out["llama::sis_steps"] = np.array([steps], np.int32)
    out["llama::sis_masks"] = torch.stack(MG.make_masks(V)).numpy()


def golden_parity_full(out):
    """torch-CPU results of the reference's per-particle op sequence (cache.py:96, README.md:84-87) at the FULL headline
    size, 1024 x 50257 fp32 (tests/synth.py logits, two masks): sampled ids, logZ and the exponential race's margins."""
    from tests import synth

    B, V = 1024, 50257
    x = torch.from_numpy(synth.logits(21, B, V))
    masks = torch.from_numpy(synth.binary_masks(21, 2, V))
    mid = torch.arange(B) % 2
    masked = torch.log_softmax(x, -1) + masks[mid]
    logZ = masked.logsumexp(-1)
    g = torch.Generator()
    g.manual_seed(2024)
    p = (masked - logZ[:, None]).exp()
    tok = torch.multinomial(p, 1, generator=g).flatten()
    g.manual_seed(2024)
    q = torch.empty(B, V).exponential_(1, generator=g)
    top2 = (p / q).topk(2, -1).values
    out["parity1024::logZ"] = logZ.numpy()
    out["parity1024::token"] = tok.numpy().astype(np.int32)
    out["parity1024::margin"] = ((top2[:, 0] - top2[:, 1]) / top2[:, 0]).numpy()


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    MG.install_shims()
    out = {}
    golden_trie(out)
    golden_byte_vocab(out)
    golden_config3(out)
    golden_llama(out)
    golden_parity_full(out)
    np.savez_compressed(os.path.join(OUT, "ref_round2.npz"), **out)
    print("ref_round2.npz:", {k: v.shape for k, v in out.items() if "::w::" not in k})
