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
            raw = re.sub(rb"<0x([0-9A-Fa-f]{2})>", lambda m: bytes.fromhex(m[1].decode()), raw)
        out.append(raw.replace("▁".encode(), b" "))
    return out


def get_byte_vocab(tokenizer):
    """list[bytes] indexed by token id."""
    table = getattr(tokenizer, "byte_decoder", None)
    if table:
        try:
            return _via_char_table(tokenizer, table)
        except ByteVocabError:
            pass
    if hasattr(tokenizer, "sp_model"):
        return _via_sentencepiece(tokenizer)
    return _via_char_table(tokenizer, gpt2_unicode_to_byte())


def decode_vocab(tokenizer, byte2str_fallback="tokenizer"):
    """(byte_vocab: list[Token], str_vocab: list[str]); token id == list index."""
    if byte2str_fallback not in ("latin1", "tokenizer", "replace"):
        raise ValueError(f"Unknown byte2str_fallback strategy: {byte2str_fallback}")
    raw = get_byte_vocab(tokenizer)
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
