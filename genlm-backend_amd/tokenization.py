"""Byte-level view of a HuggingFace tokenizer's vocabulary (host side, runs once per model).

Boundary counterpart of the reference's `decode_vocab` / `Token` (genlm/backend/tokenization/
vocab.py:9-59, bytes.py:15-115, token.py:9-90): `AsyncLM.byte_vocab[i]` is a `Token` whose bytes
are the byte string token id `i` decodes to.  Unlike the reference this never goes back to the
hub for a slow tokenizer; it works from the tokenizer object it is given.
"""
import re


class Token(bytes):
    """bytes subclass that also knows its token id; Tokens compare / hash by id (two ids may share
    a byte string), and behave as plain bytes towards everything else (token.py:9-90)."""

    def __new__(cls, token_id, byte_string):
        if not isinstance(token_id, int):
            raise TypeError(f"token_id must be an int, got {type(token_id)}")
        if not isinstance(byte_string, bytes):
            raise TypeError(f"byte_string must be bytes, got {type(byte_string)}")
        self = super().__new__(cls, byte_string)
        self.token_id = token_id
        return self

    @property
    def byte_string(self):
        return bytes(self)

    def __repr__(self):
        return f"Token(token_id={self.token_id}, byte_string={bytes(self)!r})"

    def _key(self, other):
        return other.token_id if isinstance(other, Token) else None

    def __eq__(self, other):
        return self.token_id == other.token_id if isinstance(other, Token) else NotImplemented

    def __ne__(self, other):
        return self.token_id != other.token_id if isinstance(other, Token) else NotImplemented

    def __lt__(self, other):
        return self.token_id < other.token_id if isinstance(other, Token) else NotImplemented

    def __le__(self, other):
        return self.token_id <= other.token_id if isinstance(other, Token) else NotImplemented

    def __gt__(self, other):
        return self.token_id > other.token_id if isinstance(other, Token) else NotImplemented

    def __ge__(self, other):
        return self.token_id >= other.token_id if isinstance(other, Token) else NotImplemented

    def __hash__(self):
        return hash(self.token_id)

    def __reduce__(self):
        return (Token, (self.token_id, bytes(self)))

    @staticmethod
    def as_bytes(x):
        return x.byte_string if isinstance(x, Token) else x

    @staticmethod
    def is_plain_bytes(x):
        return isinstance(x, bytes) and not isinstance(x, Token)


def gpt2_unicode_to_byte():
    """Inverse of the GPT-2 byte-level BPE alphabet: printable latin-1 code points stand for
    themselves, the remaining byte values are mapped, in order, to code points 256, 257, ..."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table = {chr(b): b for b in keep}
    nxt = 256
    for b in range(256):
        if b not in keep:
            table[chr(nxt)] = b
            nxt += 1
    return table


class ByteVocabError(ValueError):
    pass


def _added(tokenizer):
    try:
        return {i: t for t, i in tokenizer.get_added_vocab().items()}
    except Exception:
        return {}


def default_char_table():
    """The fallback character -> byte table: the GPT-2 byte-level alphabet plus the literal whitespace and the
    SentencePiece space marker some fast tokenizers keep in their pieces (bytes.py:214-231)."""
    table = gpt2_unicode_to_byte()
    table.update({" ": 32, "\n": 10, "\r": 13, "\t": 9, "\u2581": 32})
    return table


_PROBE = "\u2019\u2022\u00b6\u2202\u0192\u02d9\u2206\u00a3\u0126\u7228\u0d60\u1158\u2230\u1368"


def _table_is_usable(tokenizer, table):
    """A character table is accepted only if (a) it covers every character of every ordinary piece of the vocabulary
    and (b) a string of unusual code points survives tokenise -> pieces -> bytes (bytes.py:118-187's two checks)."""
    special = set(_added(tokenizer).values())
    try:
        vocab = tokenizer.get_vocab()
    except Exception:
        return False
    for piece in vocab:
        if piece in special:
            continue
        for ch in piece:
            if ch not in table:
                return False
    try:
        ids = tokenizer(_PROBE, add_special_tokens=False)["input_ids"]
        raw = b"".join(bytes(table[ch] for ch in tokenizer.convert_ids_to_tokens(i)) for i in ids)
        bos = getattr(tokenizer, "bos_token", None)
        if bos and raw.startswith(bos.encode()):
            raw = raw[len(bos):]
        return raw.decode() == _PROBE
    except Exception:
        return False


def _via_char_table(tokenizer, table):
    added = _added(tokenizer)
    out = []
    for i in range(len(tokenizer)):
        if i in added:
            out.append(added[i].encode())
            continue
        piece = tokenizer.convert_ids_to_tokens(i)
        if piece is None:
            out.append(b"")
            continue
        try:
            out.append(bytes(table[ch] for ch in piece))
        except KeyError as e:
            raise ByteVocabError(f"token {i} ({piece!r}) is not in the byte-level alphabet") from e
    return out


def _via_sentencepiece(tokenizer):
    added = _added(tokenizer)
    out = []
    for i in range(len(tokenizer)):
        if i in added:
            raw = added[i].encode()
        else:
            raw = tokenizer.sp_model.id_to_piece(i).encode()
            raw = re.sub(rb"<0x(..)>", lambda m: bytes.fromhex(m[1].decode()), raw)
        out.append(raw.replace("\u2581".encode(), b" "))
    return out


def get_byte_vocab(tokenizer):
    """list[bytes] indexed by token id (bytes.py:15-57): the tokenizer's own `byte_decoder` if it passes the checks,
    else its SentencePiece model, else the default byte-level table (if THAT passes the checks)."""
    table = getattr(tokenizer, "byte_decoder", None)
    if table and _table_is_usable(tokenizer, table):
        return _via_char_table(tokenizer, table)
    if hasattr(tokenizer, "sp_model"):
        return _via_sentencepiece(tokenizer)
    table = default_char_table()
    if not _table_is_usable(tokenizer, table):
        raise ByteVocabError("Could not decode vocabulary by falling back to GPT2 byte decoder.")
    return _via_char_table(tokenizer, table)


def _alternate_tokenizer(tokenizer, use_fast):
    name = getattr(tokenizer, "name_or_path", None)
    if not name:
        return None
    try:
        from transformers import AutoTokenizer

        return AutoTokenizer.from_pretrained(name, use_fast=use_fast)
    except Exception:  # offline / in-memory tokenizers: there is no other flavour to load
        return None


def decode_vocab(tokenizer, byte2str_fallback="tokenizer"):
    """(byte_vocab: list[Token], str_vocab: list[str]); token id == list index."""
    if byte2str_fallback not in ("latin1", "tokenizer", "replace"):
        raise ValueError(f"Unknown byte2str_fallback strategy: {byte2str_fallback}")
    # vocab.py:31-49: prefer the slow flavour of a fast tokenizer, fall back to the fast one
    raw = None
    first = _alternate_tokenizer(tokenizer, use_fast=False) if getattr(tokenizer, "is_fast", False) else None
    for cand in (first, tokenizer, _alternate_tokenizer(tokenizer, use_fast=True) if first is not None else None):
        if cand is None:
            continue
        try:
            raw = get_byte_vocab(cand)
            tokenizer = cand
            break
        except ByteVocabError:
            continue
    if raw is None:
        raise ValueError(f"Could not decode byte representation of token vocabuary from tokenizer "
                         f"{getattr(tokenizer, 'name_or_path', '?')}")
    byte_vocab = [Token(i, b) for i, b in enumerate(raw)]
    str_vocab = []
    for i, b in enumerate(raw):
        try:
            s = b.decode("utf-8")
        except UnicodeDecodeError:
            if byte2str_fallback == "latin1":
                s = b.decode("latin1")
            elif byte2str_fallback == "replace":
                s = b.decode("utf-8", errors="replace")
            else:
                s = tokenizer.convert_ids_to_tokens(i)
        str_vocab.append(s)
    return byte_vocab, str_vocab
