"""Round-2 fixtures made by the REFERENCE (oracle/make_goldens_r2.py -> tests/golden/ref_round2.npz): the token->byte
trie and its masses, byte vocabularies, BASELINE config 3 (K distinct ragged prompts, dedup + prefix KV) and a tiny
Llama (RoPE, grouped-query attention) through the backend.  CPU: the HIP engine is the oracle-backed test double."""
import ast
import os

import numpy as np
import pytest
import torch

from tests.cpu_engine import CpuOracleEngine

GD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GD, "ref_round2.npz"))


def _strip(row):
    return [int(t) for t in row if t >= 0]


# ------------------------------------------------------------------------------------------------ trie
def _words(gold, tag):
    return bytes(gold[f"trie::{tag}::words"]).split(b"\x00")


@pytest.mark.parametrize("tag", ["kat", "syn"])
def test_trie_structure_and_masses_match_reference(gold, oracle, tag):
    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    words = _words(gold, tag)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)])
    # same node numbering, leaves, edges (in the reference's dict order) and prefixes
    assert len(trie) == int(gold[f"trie::{tag}::n_nodes"][0]) and trie.root == int(gold[f"trie::{tag}::root"][0])
    assert np.array_equal(trie.idx_to_leaf, gold[f"trie::{tag}::idx_to_leaf"])
    tri = [(x, -1 - sym[1] if isinstance(sym, tuple) else int(sym), y) for x, ch in enumerate(trie.children)
           for sym, y in ch.items()]
    assert np.array_equal(np.array(tri, np.int64), gold[f"trie::{tag}::edges"])
    assert [len(trie.node2prefix[i]) for i in range(len(trie))] == list(gold[f"trie::{tag}::prefix_len"])
    assert trie.leaf2word[trie.word2leaf[(words[1], 1)]] == (words[1], 1)
    # masses: the oracle's restatement of the kernel contract against the reference's own numbers
    ws = gold[f"trie::{tag}::ws"]
    for op, key in ((0, "sum"), (1, "max")):
        got = oracle.trie_reduce(ws, trie.flat(), op)
        assert np.abs(got - gold[f"trie::{tag}::{key}"]).max() < 1e-6
    lp = np.log(np.maximum(ws, 1e-30)).astype(np.float32)
    got = oracle.trie_reduce(lp, trie.flat(), 0, from_logprobs=True)
    assert np.abs(got - gold[f"trie::{tag}::sum"]).max() < 1e-5
    with pytest.raises(ValueError):
        TokenByteTrie([Token(0, b"a"), Token(0, b"a")])


def test_trie_known_answers(gold, oracle):
    """tests/test_trie.py:26-85 of the reference, by prefix."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    trie = TokenByteTrie([Token(0, b"a"), Token(1, b"b"), Token(2, b"ab"), Token(3, b"<eos>")])
    ws = np.array([[0.1, 0.2, 0.2, 0.5]], np.float32)
    s, m = oracle.trie_reduce(ws, trie.flat(), 0)[0], oracle.trie_reduce(ws, trie.flat(), 1)[0]
    leaf = {b"a": 0.1, b"b": 0.2, b"ab": 0.2, b"<eos>": 0.5}
    want_s = {b"": 1, b"a": 0.3, b"b": 0.2, b"ab": 0.2, b"<": 0.5, b"<e": 0.5, b"<eo": 0.5, b"<eos": 0.5, b"<eos>": 0.5}
    want_m = dict(want_s)
    want_m.update({b"": 0.5, b"a": 0.2})
    for node, prefix in trie.node2prefix.items():
        p = bytes(prefix)
        if node in trie.leaf2word:
            assert np.isclose(s[node], leaf[p]) and np.isclose(m[node], leaf[p])
        else:
            assert np.isclose(s[node], want_s[p], rtol=1e-5) and np.isclose(m[node], want_m[p], rtol=1e-5)


# ------------------------------------------------------------------------------------------------ byte vocabulary
def _split(gold, tag):
    lens, raw = gold[f"bv::{tag}::lens"], bytes(gold[f"bv::{tag}::bytes"])
    out, at = [], 0
    for n in lens:
        out.append(raw[at:at + int(n)])
        at += int(n)
    return out


def test_byte_vocab_matches_reference_on_bpe_and_sentencepiece(gold):
    import sentencepiece as spm
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.tokenization import decode_vocab, get_byte_vocab

    fast = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_file(os.path.join(GD, "bpe_tokenizer.json")),
                                   eos_token="<|endoftext|>")
    assert get_byte_vocab(fast) == _split(gold, "bpe")
    bv, sv = decode_vocab(fast)
    assert [t.byte_string for t in bv] == _split(gold, "bpe") and len(sv) == len(bv)

    class SpTok:  # the slice of a slow SentencePiece tokenizer the decoder reads
        def __init__(self, path):
            self.sp_model = spm.SentencePieceProcessor(model_file=path)
            self._added = {"<extra_0>": self.sp_model.get_piece_size()}

        def get_added_vocab(self):
            return dict(self._added)

        def __len__(self):
            return self.sp_model.get_piece_size() + 1

    assert get_byte_vocab(SpTok(os.path.join(GD, "spm_tiny.model"))) == _split(gold, "spm")


def test_default_table_knows_whitespace_and_the_sentencepiece_space():
    """Pieces of fast tokenizers without byte_decoder / sp_model may carry U+2581 and literal whitespace (Llama-2 /
    Mistral / T5 style): the default table maps them (bytes.py:214-231).  A table that cannot reproduce the probe
    string is refused (bytes.py:118-187), here as in the reference."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from genlm_backend_amd.tokenization import ByteVocabError, _via_char_table, default_char_table, get_byte_vocab

    table = default_char_table()
    assert table["\u2581"] == 32 and table[" "] == 32 and table["\n"] == 10 and table["\t"] == 9 and table["\r"] == 13
    vocab = {"<unk>": 0, "\u2581the": 1, "\u2581": 2, "a b": 3, "x\n": 4}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>")
    bv = _via_char_table(fast, table)
    assert bv[1] == b" the" and bv[2] == b" " and bv[3] == b"a b" and bv[4] == b"x\n"
    with pytest.raises(ByteVocabError):  # this toy vocabulary cannot spell the probe string: not a byte-level table
        get_byte_vocab(fast)


# ------------------------------------------------------------------------------------------------ config 3 / llama
class Tok:
    pad_token_id = None
    eos_token_id = 0


def _gpt2_tiny():
    from transformers import GPT2Config, GPT2LMHeadModel

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.llm import AsyncAmdLM

    g = np.load(os.path.join(GD, "ref_hotpath_tiny.npz"))
    cfg = ast.literal_eval(bytes(g["config_json"]).decode())
    model = GPT2LMHeadModel(GPT2Config(**cfg)).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")})
    m = AsyncAmdLM(model, None, batch_size=128, engine=CpuOracleEngine())
    m.tokenizer = Tok()
    return m


@pytest.mark.parametrize("K", [1, 8, 64])
@pytest.mark.parametrize("mode", ["plain", "prefix", "pkv"])
def test_config3_shared_ragged_prompts(gold, K, mode):
    """BASELINE config 3: particles over K distinct shared prompts of ragged length; dedup + (prefix | per-particle) KV.
    Tokens equal the reference's run, weights within 1e-4."""
    from genlm_backend_amd.sis import DeviceSIS

    m = _gpt2_tiny()
    m.register_masks(torch.from_numpy(gold["c3::masks"]))
    prompts = [_strip(r) for r in gold[f"c3::K{K}::prompts"]]
    per = [prompts[i % K] for i in range(64)]
    sis = DeviceSIS(m, 64, per, max_tokens=6, eos_id=0, seed=4321 + K, rng="torch", use_prefix_kv=mode == "prefix",
                    use_particle_kv=mode == "pkv")
    sis.run()
    ctx, lw = sis.results()
    assert [list(map(int, c)) for c in ctx] == [_strip(r) for r in gold[f"c3::K{K}::contexts"]]
    assert np.abs(lw - gold[f"c3::K{K}::log_weights"]).max() < TOL
    if K == 1:  # the reference's own cache_kv run agrees with its plain run
        assert np.array_equal(gold["c3::K1::contexts_kv"], gold["c3::K1::contexts"])
        assert np.abs(lw - gold["c3::K1::log_weights_kv"]).max() < TOL
    assert m.stats["unique"] == 0  # DeviceSIS bypasses the queue; dedup happened on the "device"
    assert sis.last_stats["n_unique"] <= 64


def _llama_tiny(gold, engine=None, device="cpu", **kw):
    from transformers import LlamaConfig, LlamaForCausalLM

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.llm import AsyncAmdLM

    cfg = ast.literal_eval(bytes(gold["llama::config_json"]).decode())
    model = LlamaForCausalLM(LlamaConfig(**cfg)).eval()
    model.load_state_dict({k[len("llama::w::"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("llama::w::")})
    m = AsyncAmdLM(model.to(device), None, batch_size=64, engine=engine or CpuOracleEngine(), **kw)
    m.tokenizer = Tok()
    return m


def test_llama_shaped_model_matches_reference(gold):
    """BASELINE config 4's model family (RoPE, grouped-query attention) through the backend: batched log-probs of
    ragged prompts, and the SIS loop over three shared prompts with every KV variant."""
    import asyncio

    from genlm_backend_amd.sis import DeviceSIS

    m = _llama_tiny(gold)
    prompts = [_strip(r) for r in gold["llama::lp_prompts"]]
    got = asyncio.run(m.batch_next_token_logprobs(prompts)).numpy()
    assert np.abs(got - gold["llama::lp_values"]).max() < TOL
    assert np.abs(got - gold["llama::lp_uncached"]).max() < TOL
    m.register_masks(torch.from_numpy(gold["llama::sis_masks"]))
    p3 = [_strip(r) for r in gold["llama::sis_prompts"]]
    per = [p3[i % 3] for i in range(24)]
    for kw in (dict(), dict(use_prefix_kv=True), dict(use_particle_kv=True)):
        sis = DeviceSIS(m, 24, per, max_tokens=6, eos_id=0, seed=999, rng="torch", **kw)
        sis.run()
        ctx, lw = sis.results()
        assert [list(map(int, c)) for c in ctx] == [_strip(r) for r in gold["llama::sis_contexts"]]
        assert np.abs(lw - gold["llama::sis_log_weights"]).max() < TOL
    # the same loop through the stateless API, with and without KV rows that follow the contexts (autokv.AutoKV)
    for kw in (dict(), dict(auto_kv_rows=40, auto_kv_cap=24)):
        a = _llama_tiny(gold, **kw)
        a.register_masks(torch.from_numpy(gold["llama::sis_masks"]))
        a.set_rng("torch", 999)
        gen, lw, active = [[] for _ in range(24)], np.zeros(24, np.float64), [True] * 24
        while any(active):
            idx = [i for i in range(24) if active[i]]
            logZ, tok = a.batch_next_token_step_sync([per[i] + gen[i] for i in idx], [1 if len(gen[i]) >= 6 else 0 for i in idx])
            for i, z, t in zip(idx, logZ, tok):
                lw[i] += z
                if t == 0 or t < 0:
                    active[i] = False
                else:
                    gen[i].append(int(t))
        assert gen == [_strip(r) for r in gold["llama::sis_contexts"]]
        assert np.abs(lw.astype(np.float32) - gold["llama::sis_log_weights"]).max() < TOL
        if kw:
            assert a._auto_kv.stats["encoded_rows"] == 3 and a._auto_kv.stats["one_token_rows"] > 24


def test_trie_accepts_plain_iterables_and_refuses_duplicates():
    """tests/test_trie.py:288-325 of the reference: plain bytes (with a deprecation warning) and other iterables of
    symbols beside `Token`s; tokens with the same bytes but different ids get leaves of their own (test_duplicates.py);
    a word that occurs twice is an error."""
    import warnings

    import genlm_backend_amd  # noqa: F401
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    with pytest.warns(DeprecationWarning, match="plain bytes"):
        trie = TokenByteTrie([Token(0, b"hello"), b"world", Token(2, b"test"), b"data", ("e", "o", "s")])
    assert (b"hello", 0) in trie.word2leaf and b"world" in trie.word2leaf and (b"test", 2) in trie.word2leaf
    assert b"data" in trie.word2leaf and ("e", "o", "s") in trie.word2leaf
    assert len(trie.idx_to_leaf) == 5 and len(set(trie.idx_to_leaf[:, 1])) == 5
    dup = TokenByteTrie([Token(0, b"ab"), Token(1, b"ab"), Token(2, b"a")])  # one byte string, two token ids
    assert dup.word2leaf[(b"ab", 0)] != dup.word2leaf[(b"ab", 1)]
    c = dup.compact()
    assert c["n_nodes"] < len(dup) and len(set(c["leaf_node"])) == 3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        with pytest.raises(ValueError, match="Duplicate word in vocabulary"):
            TokenByteTrie([b"hello", b"world", b"hello"])
    with pytest.raises(ValueError, match="Duplicate word in vocabulary"):
        TokenByteTrie([Token(0, b"test"), Token(1, b"other"), Token(0, b"test")])


def _run_plan(pl, ws, op, sel_nodes=None):
    """What glb_trie_rows does with a plan (glb_trie.hip), in numpy: per part the leaves, then depth by depth - children
    consecutive, ascending, in double, stored as float32 -, then the top from the parts' roots.  Returns [B, n_nodes], or
    with `sel_nodes` the selected nodes' values [B, len(sel_nodes)] the way part_write finds them (by slot)."""
    B = ws.shape[0]
    out = np.full((B, pl["n_nodes"]), np.nan, np.float32)
    sel_slot = None if sel_nodes is None else pl["slot_of"][np.asarray(sel_nodes, np.int64)].astype(np.int64)
    out_sel = None if sel_nodes is None else np.full((B, len(sel_nodes)), np.nan, np.float32)
    cut = np.full((B, max(pl["n_cut"], 1)), np.nan, np.float32)
    parts = list(range(pl["n_parts"])) + ([pl["n_parts"]] if pl["n_top"] else [])
    for p in parts:
        d = pl["desc"][p]
        top = p == pl["n_parts"]
        n_local, n_roots, n_depths = int(d[1]), int(d[2]), int(d[3])
        ds = pl["depth_start"][d[4]:d[4] + n_depths + 1]
        assert d[5] % 2 == 0 and d[11] % 2 == 0  # (the kernel copies the 16-bit tables word by word)
        cp = pl["cptr16"][d[5]:d[5] + n_local + 1].astype(np.int64)
        ins = pl["inode16"][d[11]:d[11] + d[12]].astype(np.int64)
        idp = pl["idepth"][d[13]:d[13] + n_depths + 1]
        src, loc = pl["leaf_src"][d[6]:d[6] + d[7]], pl["leaf_local"][d[6]:d[6] + d[7]]
        assert ds[0] == 0 and ds[-1] == n_local and cp[-1] == n_local and (np.diff(cp) >= 0).all()
        assert idp[0] == 0 and idp[-1] == len(ins) == int((np.diff(cp) > 0).sum()) and len(set(ins)) == len(ins)
        swept = bool(pl.get("sweep")) and not top  # (the kernel that reads a row front to back: round 5)
        if swept:
            # its tables say what the gathered plan's do: every token has a slot (its own, or a word of the 32-word slack behind
            # the values), the internal nodes are inode16's with their children's range
            V = pl["vocab"]
            vp = (V + 7) & ~7
            tl = pl["tok_local16"][p * vp:(p + 1) * vp].astype(np.int64)
            mine = np.zeros(vp, bool)
            mine[src] = True
            assert np.array_equal(tl[src], loc) and ((tl[~mine] >= n_local) & (tl[~mine] < n_local + 32)).all()
            e = pl["inode64"][d[11]:d[11] + d[12]].astype(np.uint64)
            assert np.array_equal((e & np.uint64(0xffff)).astype(np.int64), ins)
            assert np.array_equal(((e >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64), cp[ins])
            assert np.array_equal((e >> np.uint64(32)).astype(np.int64), cp[ins + 1] - cp[ins])
            assert 4 * n_local + 128 <= pl["lds_bytes"] <= 160 * 1024 and n_depths <= 30 and n_local + 32 < 65536
        else:
            assert pl["lds_bytes" if not (top and pl.get("sweep")) else "lds_top_bytes"] >= 4 * n_local + 2 * (n_local + 1) + 2 * len(ins) + 4 * (n_depths + 1) and n_depths <= 30
        for r in range(B):
            if swept:
                val = np.full(n_local + 32, np.nan, np.float32)
                val[tl[:V]] = ws[r, :V]  # (every token is stored)
                val = val[:n_local]
            else:
                val = np.full(n_local, np.nan, np.float32)
                val[loc] = cut[r, src] if top else ws[r, src]
            for k in range(n_depths - 2, -1, -1):
                for s in ins[idp[k]:idp[k + 1]]:
                    assert ds[k] <= s < ds[k + 1] and cp[s] >= ds[k + 1] and cp[s + 1] <= ds[k + 2]  # children sit one depth down
                    acc = 0.0
                    for v in val[cp[s]:cp[s + 1]].astype(np.float64):
                        acc = acc + v if op == 0 else max(acc, v)
                    val[s] = np.float32(acc)
            assert not np.isnan(val).any()
            nl = pl["pn_local16"][d[9]:d[9] + d[10]].astype(np.int64)
            runs = pl["run_tab"][d[14]:d[14] + d[15]]
            nd = np.concatenate([np.arange(lo, lo + cnt) for lo, cnt in runs]) if len(runs) else np.zeros(0, np.int64)
            assert np.array_equal(nd, pl["pn_node"][d[9]:d[9] + d[10]]) and (top or len(runs) <= int(d[2]))  # (a run per subtree)
            out[r, nd] = val[nl]
            if sel_slot is not None:
                base = int(d[0])
                if top:
                    mine = sel_slot >= base
                    out_sel[r, mine] = val[pl["top_local"][sel_slot[mine] - base]]
                else:
                    mine = (sel_slot >= base) & (sel_slot < base + n_local)
                    out_sel[r, mine] = val[sel_slot[mine] - base]
            if not top:
                cut[r, d[8]:d[8] + n_roots] = val[:n_roots]
    return out if sel_nodes is None else out_sel


@pytest.mark.parametrize("cap", [60, 300, 20000])
def test_selection_plans_give_the_reference_masses(oracle, cap):
    """TokenByteTrie._build_plan for a SELECTION of nodes (round 5): only the subtrees below the selection's maximal nodes
    are planned - fewer leaves read, fewer nodes reduced -, and the selected nodes' values are still the oracle's, bit for
    bit: leaves only, nested selections, a selection whose subtrees need a top of their own (small caps), the root."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(8)
    words, seen = [], set()
    while len(words) < 1500:
        w = bytes(rs.integers(97, 103, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)])
    c = trie.compact()
    kids, size, root = trie._tree()
    ws = rs.random((3, len(words))).astype(np.float32)
    n = len(trie)
    depth1 = sorted(trie.children[trie.root].values())
    for sel in (rs.choice(n, 200, replace=False), np.asarray(depth1[:2] + [int(trie.jump[depth1[0]][0])]),
                trie.idx_to_leaf[:50, 1].astype(np.int64), np.asarray(depth1)):
        sel = np.asarray(sel, np.int64)
        slots = np.unique(c["slot_of"].astype(np.int64)[sel])
        roots, lo = [], None
        for s_ in slots[::-1]:
            if lo is not None and s_ > lo:
                continue
            roots.append(int(s_))
            lo = int(s_) - int(size[s_])
        # no root below another
        for a in roots:
            assert not any(b != a and a - size[a] < b <= a for b in roots)
        pl = trie._build_plan(cap, roots)
        if pl is None:
            continue
        needed = sum(int(size[r]) for r in roots)
        assert pl["n_slots"] == needed <= c["n_nodes"] and (pl["slot_of"][sel] >= 0).all()
        for op in (0, 1):
            want = oracle.trie_reduce(ws, trie.flat(), op)[:, sel]
            got = _run_plan(pl, ws, op, sel_nodes=sel)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (cap, op)


@pytest.mark.parametrize("cap", [40, 300, 20000])
def test_trie_plan_gives_the_reference_masses(gold, oracle, cap):
    """TokenByteTrie.plan (the folded trie cut into LDS-sized parts for glb_trie_rows): every node's value lands in
    exactly one part, a node's children are consecutive and one depth down, and the planned arithmetic equals the
    oracle's level-synchronous loop bit for bit - with a cut below the root (small caps) and without one."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(5)
    words, seen = [], set()
    while len(words) < 1200:
        w = bytes(rs.integers(97, 103, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    for vocab in (words, _words(gold, "kat")):
        trie = TokenByteTrie([Token(i, w) for i, w in enumerate(vocab)])
        pl = trie.plan(cap)
        if pl is None:  # (a node with more children than a part holds: the level-synchronous kernels serve such a trie)
            assert cap == 40
            continue
        assert pl["max_local"] <= cap and pl["n_slots"] == trie.compact()["n_nodes"]
        assert (pl["n_top"] == 0) == (pl["n_slots"] <= cap)
        assert sorted(pl["slot_of"][np.unique(pl["slot_of"], return_index=True)[1]]) == list(range(pl["n_slots"]))
        assert np.array_equal(np.sort(np.concatenate([pl["pn_node"]])), np.arange(len(trie)))
        ws = rs.random((3, len(vocab))).astype(np.float32)
        for op in (0, 1):
            got = _run_plan(pl, ws, op)
            assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(ws, trie.flat(), op).view(np.uint32))
        # the same cut for the kernel that reads a row front to back (round 5): its two tables restate the gathered plan's
        sw = trie.plan(cap, sweep=True)
        assert sw is not None and sw["sweep"] and sw["n_parts"] == pl["n_parts"] and np.array_equal(sw["slot_of"], pl["slot_of"])
        for op in (0, 1):
            got = _run_plan(sw, ws, op)
            assert np.array_equal(got.view(np.uint32), oracle.trie_reduce(ws, trie.flat(), op).view(np.uint32))
        assert trie.slot_plan() is trie.plan(sweep=True) and trie.plan(sweep=True)["n_parts"] == 1  # (small tries: one part, no top)


def test_sweep_cap_packs_the_trie_into_the_parts_it_promises():
    """TokenByteTrie.sweep_cap / _count_parts (round 5): the part count predicted from the cut and the packing alone is the
    one `_build_plan` makes, and the cap chosen for the sweep plan gives the fewest parts the LDS limit allows, each within
    it - on a vocabulary big enough for several parts when the limit is lowered."""
    from genlm_backend_amd.tokenization import Token
    from genlm_backend_amd.trie import TokenByteTrie

    rs = np.random.default_rng(12)
    words, seen = [], set()
    while len(words) < 6000:
        w = bytes(rs.integers(97, 110, int(rs.integers(1, 7))).astype(np.uint8))
        if w not in seen:
            seen.add(w)
            words.append(w)
    trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)])
    n_slots = int(trie.compact()["n_nodes"])
    for cap in (500, 1500, 4000, n_slots):
        pl = trie.plan(cap, sweep=True)
        assert pl is not None and trie._count_parts(cap) == pl["n_parts"] and pl["max_local"] <= cap
    trie.SWEEP_LOCAL_MAX = 3000  # (as if the LDS held 3000 slots)
    cap = trie.sweep_cap()
    pl = trie.plan(sweep=True)
    assert cap <= 3000 and pl["cap"] == cap and pl["max_local"] <= cap
    assert pl["n_parts"] <= -(-n_slots // 3000) + 1  # (the fewest parts, give or take one for the packing)
    assert pl["lds_bytes"] == 4 * pl["max_local"] + 128 and pl["tok_local16"].shape == (pl["n_parts"] * ((len(words) + 7) & ~7),)
