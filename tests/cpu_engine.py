"""Test double for `HipEngine` built on the CPU oracle, so the host logic (queueing, dedup order,
trie, sharding) is testable without a GPU.  TEST INFRASTRUCTURE: lives under tests/, is never
imported by the product package."""
import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    return None if t is None else t.detach().cpu().contiguous().numpy()


def _logits_np(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.detach().cpu().contiguous().numpy()


class _Prepared:
    """stand-in for engine.PreparedMasks: the oracle works on the plain bit rows"""

    def __init__(self, bits):
        self.bits = bits


class _OracleNoise:
    """stand-in for engine.DeviceRng: the oracle's serial MT19937 -> exponential_ stream dealt out by row slots"""

    def __init__(self, seed, vocab):
        self.seed, self.vocab = seed, vocab
        self.reset()

    def reset(self):
        self.state = None

    def prefetch(self):  # (DeviceRng's side-stream generation: nothing to get ahead of here)
        pass

    def rows(self, n_out, row_slot=None, n_draw=None, max_draw=None, out=None):
        V = self.vocab
        slot = np.arange(n_out) if row_slot is None else row_slot.cpu().numpy()
        n = int(n_out if max_draw is None else max_draw) if n_draw is None else int(n_draw)
        assert (slot < n).all()
        block, self.state = O.mt_exponential(self.seed, n * V, self.state)
        block = block.reshape(n, V)
        res = np.ones((n_out, V), np.float32)
        res[slot >= 0] = block[slot[slot >= 0]]
        return torch.from_numpy(res)


class CpuOracleEngine:
    device = torch.device("cpu")

    def noise_rng(self, seed, vocab, ahead=False):
        return _OracleNoise(seed, vocab)

    def prepare_masks(self, bits, vocab, logits_dtype=torch.float32):
        return _Prepared(bits.clone())  # (a snapshot, as the device's prepared form is: later writes to `bits` do not reach it)

    def update_prepared_masks(self, prepared, bits, rows):
        idx = rows.long()
        idx = idx[(idx >= 0) & (idx < bits.shape[0])]
        prepared.bits[idx] = bits[idx]  # (only the named rows, as glb_mask_prepare_rows)
        return prepared

    def step(self, logits, vocab=None, row_of=None, mask_kind=0, mask=None, mask_id=None, rng_mode=0, noise=None,
             seed=0, offset=0, particle_base=0, logit_scale=1.0, want_lse=True, out=None, row_mask_id=None,
             out_margin=None):
        V = logits.shape[1] if vocab is None else vocab
        x = _logits_np(logits[:, :V])
        if isinstance(mask, _Prepared):
            mask, mask_kind = mask.bits, 1
        m = _np(mask)
        if mask_kind == 1:
            m = m.view(np.uint32)
        if noise is not None and noise.shape[0] == 1:  # one noise row shared by every particle
            n_p = logits.shape[0] if row_of is None else row_of.numel()
            noise = noise.expand(n_p, noise.shape[1])
        if row_mask_id is not None:  # ids per logits row -> per particle for the oracle
            rm = _np(row_mask_id)
            mask_id = torch.from_numpy(rm if row_of is None else rm[_np(row_of)])
        logZ, lse, tok = O.step(x, row_of=_np(row_of), mask_kind=mask_kind, mask=m, mask_id=_np(mask_id),
                                rng_mode=rng_mode, noise=_np(noise), seed=seed, offset=offset,
                                particle_base=particle_base, logit_scale=logit_scale)
        res = (torch.from_numpy(logZ), torch.from_numpy(lse) if want_lse else None,
               torch.from_numpy(tok) if rng_mode else None)
        if out is not None:
            for o, r in zip(out, res):
                if o is not None and r is not None:
                    o.copy_(r)
            return out
        return res

    def error_word(self):
        return torch.zeros(1, dtype=torch.int32)

    def raise_if_failed(self, err_count=None, tokens=None, lse=None, what=""):
        assert not err_count and (tokens is None or not (np.asarray(tokens) == -2).any())

    def check(self):
        pass

    def log_softmax_rows(self, logits, vocab=None, logit_scale=1.0, out=None, want_lse=False, out_dtype=torch.float32):
        V = logits.shape[1] if vocab is None else vocab
        lp, lse = O.log_softmax_rows(_logits_np(logits[:, :V]), logit_scale)
        lp = torch.from_numpy(lp).to(out_dtype)  # (torch rounds float32 -> bf16 / f16 to nearest even)
        if out is not None:
            out.copy_(lp)
            lp = out
        return (lp, torch.from_numpy(lse)) if want_lse else lp

    def mask_to_bits(self, mask):
        if mask.dim() == 1:
            mask = mask[None]
        bits, nb = O.mask_f32_to_bits(_np(mask.to(torch.float32)))
        return torch.from_numpy(bits.view(np.int32)), torch.tensor([int(nb)], dtype=torch.int32)

    @staticmethod
    def _ctxs(tokens, starts, lengths):
        t, s, l = _np(tokens), _np(starts), _np(lengths)
        return [list(t[s[i]:s[i] + l[i]]) for i in range(len(l))]

    @staticmethod
    def _hash(ctx):
        """The library's context hash (glb_api.hip ctx_hash_step): an FNV-style fold, one token at a time."""
        M = (1 << 64) - 1
        h = 0xcbf29ce484222325
        for t in ctx:
            h ^= int(t) & 0xffffffff
            h = (h * 0x100000001b3) & M
            h ^= h >> 29
        return h

    def hash_contexts(self, tokens, starts, lengths):
        hs = [self._hash(c) for c in self._ctxs(tokens, starts, lengths)]
        return torch.from_numpy(np.array(hs, np.uint64).view(np.int64).copy())

    def group_contexts(self, tokens, starts, lengths, hashes=None):
        if hashes is not None:  # hashes kept by the caller must be the hashes of the contexts it passes
            want = self.hash_contexts(tokens, starts, lengths)
            assert torch.equal(hashes.cpu(), want), "stale context hashes"
        g, rep, ng = O.group_contexts(self._ctxs(tokens, starts, lengths))
        rep_full = np.zeros(len(g), np.int32)
        rep_full[:ng] = rep
        return torch.from_numpy(g), torch.from_numpy(rep_full), torch.tensor([ng], dtype=torch.int32)

    def match_prefixes(self, tokens, starts, lengths, ptok, pst, pln):
        pre = [] if pln is None else self._ctxs(ptok, pst, pln)
        p, b = O.match_prefixes(self._ctxs(tokens, starts, lengths), pre)
        return torch.from_numpy(p), torch.from_numpy(b)

    def gather_padded(self, tokens, starts, lengths, sel, n_sel, base, pad_id, p_max, l_max):
        t, s, l = _np(tokens), _np(starts), _np(lengths)
        b = _np(base)
        sel_np = np.arange(n_sel) if sel is None else _np(sel)[:n_sel]
        ids = np.full((n_sel, l_max), pad_id, np.int64)
        am = np.zeros((n_sel, p_max + l_max), np.int64)
        pos = np.zeros((n_sel, l_max), np.int64)
        last = np.zeros(n_sel, np.int32)
        for u, c in enumerate(sel_np):
            bb = 0 if b is None else int(b[c])
            ln = int(l[c]) - bb
            ids[u, :ln] = t[s[c] + bb:s[c] + bb + ln]
            pos[u, :ln] = np.arange(bb, bb + ln)
            am[u, :bb] = 1
            am[u, p_max:p_max + ln] = 1
            last[u] = ln - 1
        return torch.from_numpy(ids), torch.from_numpy(am), torch.from_numpy(pos), torch.from_numpy(last)

    def gather_kv_padded(self, slab_ptrs, slab_len, prefix_of, heads, head_dim, p_max, dtype):
        import ctypes as C

        ptrs = [int(p) for p in slab_ptrs.tolist()]
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        n_rows = prefix_of.numel()
        out = torch.empty((n_rows, heads, p_max, head_dim), dtype=dtype)
        sl, po = _np(slab_len), _np(prefix_of)
        O.lib().orc_gather_kv_padded(arr, sl.ctypes.data_as(C.c_void_p), C.c_int64(len(ptrs)),
                                     po.ctypes.data_as(C.c_void_p), C.c_int64(n_rows), C.c_int64(heads),
                                     C.c_int64(head_dim), C.c_int64(p_max), C.c_int32(out.element_size()),
                                     C.c_void_p(out.data_ptr()))
        return out

    def particles_advance(self, contexts, lengths, active, log_weights, logZ, token, eos_id, max_len, hashes=None):
        c, l, a, w = _np(contexts), _np(lengths), _np(active), _np(log_weights)
        l_before = l.copy()
        O.particles_advance(c, l, a, w, _np(logZ), _np(token), eos_id, max_len)
        if hashes is not None:
            M = (1 << 64) - 1
            hv = hashes.numpy().view(np.uint64)
            for i in np.nonzero(l != l_before)[0]:
                h = int(hv[i]) ^ (int(c[i, l_before[i]]) & 0xffffffff)
                h = (h * 0x100000001b3) & M
                hv[i] = h ^ (h >> 29)
        contexts.copy_(torch.from_numpy(c))
        lengths.copy_(torch.from_numpy(l))
        active.copy_(torch.from_numpy(a))
        log_weights.copy_(torch.from_numpy(w))

    def normalize_weights(self, log_weights):
        p, s = O.normalize_weights(_np(log_weights))
        return torch.from_numpy(p), torch.from_numpy(s)

    # ---- device-resident particle state (torch / oracle stand-ins) ---------------------------------------------
    def kv_append(self, slab, new_rows, pos, rows=None):
        n = new_rows.shape[0]
        r = torch.arange(n) if rows is None else rows.long()
        slab[r, :, pos.long()] = new_rows[:, :, 0]

    def kv_gather_rows(self, srcs, dsts, src_row_of, len_of, srcs_stable=True):
        sr, ln = _np(src_row_of), _np(len_of)
        for s_, d_ in zip(srcs, dsts):
            for i in range(d_.shape[0]):
                if sr[i] >= 0:
                    L = min(int(ln[i]), s_.shape[2], d_.shape[2])
                    d_[i, :, :L] = s_[sr[i], :, :L]

    def gather_rows_i32(self, src, row_of, out=None):
        res = src[row_of.long()].contiguous()
        if out is not None:
            out.copy_(res)
            return out
        return res

    def match_rows(self, tokens, starts, lengths, rep, n_groups, row_tok, row_len, row_hash):
        ctxs = self._ctxs(tokens, starts, lengths)
        U = int(n_groups.item())
        old, gh = O.match_rows(ctxs, _np(rep), U, _np(row_tok), _np(row_len), _np(row_hash).view(np.uint64))
        n = len(ctxs)
        old_f, gh_f = np.full(n, -1, np.int32), np.zeros(n, np.uint64)
        old_f[:U], gh_f[:U] = old, gh
        return torch.from_numpy(old_f), torch.from_numpy(gh_f.view(np.int64).copy())

    def kv_plan(self, group_of, rep, n_groups, old_row, lengths, n_rows, cap, by_context=False, stamps=None, call_no=0,
                table=None):
        U = int(n_groups.item())
        rep_h, old_h = _np(rep), _np(old_row)
        old_g = old_h[rep_h[:U]] if by_context else old_h[:U]
        st = None if stamps is None else stamps.numpy()  # (updated in place)
        out = O.kv_plan(_np(group_of), rep_h, U, old_g, _np(lengths), n_rows, cap, stamps=st, call_no=call_no)
        out.pop("n_valid")
        if table is not None:
            row_tok, row_len, row_hash, group_hash, tokens, starts = table
            t, s, gh = _np(tokens), _np(starts), _np(group_hash)
            for u in range(U):
                r = int(out["group_row"][u])
                if r >= 0:
                    L = int(_np(lengths)[rep_h[u]])
                    row_tok[r] = 0
                    row_tok[r, :L] = torch.from_numpy(t[s[rep_h[u]]:s[rep_h[u]] + L].copy())
                    row_len[r] = L
                    row_hash[r] = int(gh[u])
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}

    def resample_systematic(self, log_weights, seed, offset):
        anc, lse = O.resample_systematic(_np(log_weights), seed, offset)
        return torch.from_numpy(anc), torch.tensor([lse], dtype=torch.float32)
