"""byte_vocab boundary (tokenization/vocab.py:9-59, bytes.py:15-115, token.py:9-90 of the reference) on an
in-memory byte-level BPE tokenizer (no hub access), and the README's mask builders on top of it."""
import asyncio

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def tokenizer():
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    corpus = ["Montreal is a city in Canada.", "the quick brown fox jumps over the lazy dog", "naïve café ☕ 東京",
              "def f(x):\n    return x + 1\n"] * 20
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=400, special_tokens=["<|endoftext|>"],
                                                        initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    return PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|endoftext|>")


def test_token_semantics():
    from genlm_backend_amd.tokenization import Token

    a, b, c = Token(1, b"ab"), Token(2, b"ab"), Token(1, b"zz")
    assert a != b and a == c and hash(a) == hash(c) and a < b  # identity is the token id (token.py:9-90)
    assert bytes(a) == b"ab" and b"".join([a, b]) == b"abab" and len(a) == 2
    assert a.byte_string == b"ab" and Token.as_bytes(a) == b"ab" and not Token.is_plain_bytes(a)
    import pickle
    assert pickle.loads(pickle.dumps(a)).token_id == 1
    with pytest.raises(TypeError):
        Token("1", b"x")


def test_byte_vocab_roundtrip(tokenizer):
    from genlm_backend_amd.tokenization import decode_vocab

    byte_vocab, str_vocab = decode_vocab(tokenizer)
    assert len(byte_vocab) == len(tokenizer) == len(str_vocab)
    assert all(t.token_id == i for i, t in enumerate(byte_vocab))
    for text in ["Montreal is", "naïve café ☕", "東京 fox\n", "x + 1"]:
        ids = tokenizer.encode(text)
        assert b"".join(byte_vocab[i] for i in ids).decode("utf-8") == text  # test_vocabulary.py:30-83 property
    assert byte_vocab[tokenizer.eos_token_id].byte_string == b"<|endoftext|>"
    with pytest.raises(ValueError):
        decode_vocab(tokenizer, byte2str_fallback="nope")


def test_readme_mask_builders_and_sis(tokenizer):
    """README.md:57-115 end to end on the CPU test engine: byte-length masks, SIS, weight normalisation."""
    from transformers import GPT2Config, GPT2LMHeadModel

    from genlm_backend_amd.llm import AsyncAmdLM
    from genlm_backend_amd.sis import autobatched_sis, make_masking_function
    from tests.cpu_engine import CpuOracleEngine

    V = len(tokenizer)
    torch.manual_seed(0)
    model = GPT2LMHeadModel(GPT2Config(vocab_size=V, n_positions=64, n_embd=32, n_layer=1, n_head=2)).eval()
    llm = AsyncAmdLM(model, tokenizer, batch_size=64, engine=CpuOracleEngine())
    assert len(llm.byte_vocab) == V
    sel = make_masking_function(llm, max_token_length=3, max_tokens=4)
    assert llm._mask_kind == 1  # {0,-inf} masks were packed to bit rows
    llm.set_rng("philox", 7)
    eos = tokenizer.eos_token_id
    parts = asyncio.run(autobatched_sis(8, llm, sel, tokenizer.encode("Montreal is"), eos_id=eos))
    for p in parts:
        assert len(p.context) <= 4 and not p.active
        assert all(len(llm.byte_vocab[t]) <= 3 for t in p.context)  # the mask was honoured
    lw = torch.tensor([p.log_weight for p in parts])
    probs = torch.exp(lw - lw.logsumexp(-1))
    assert abs(float(probs.sum()) - 1.0) < 1e-5
