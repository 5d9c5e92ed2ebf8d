"""KV rows that follow the contexts of a stateless API.

`AsyncAmdLM.batch_next_token_step` is handed plain token lists, step after step; a population that grows by one token
per step asks for contexts whose first L - 1 tokens were evaluated the step before.  The reference re-encodes them in
full unless the caller pinned a prompt with `cache_kv` (hf.py:155-164); its MLX backend keeps per-token KV on the
trie's nodes instead (mlx.py:177-318, cache.py:103-191).  `AutoKV` is that idea on slab rows: R rows of `cap` positions
(`kv.SharedSlabKV`), a device table of the context every row holds (tokens, length, hash), and per call

  * a lookup of every distinct context - the row that holds exactly it, else the row that holds its first L - 1 tokens
    (sorted hashes + binary search, then the tokens are compared: a hash never decides alone);
  * contexts with a row feed ONE token (the first to claim a row keeps it, the others get a copy of the prefix in a row
    nobody used for the longest time: copy-on-append); contexts without one are encoded from their tokens and their
    KV kept if a row is to be had;
  * the table rows of everything that now holds a context are rewritten.

The block table is decided on the host from the call's one D2H copy, like `DeviceSIS._step_shared_kv`; rows move on the
device.
"""
import numpy as np
import torch


class AutoKV:
    def __init__(self, llm, rows, cap=64, in_place=0.75, graph=True):
        self.llm, self.eng, self.dev = llm, llm.engine, llm.device
        self.R, self.cap, self.in_place, self.graph = int(rows), int(cap), in_place, graph
        self.pkv = None
        self._slab_fwd = None
        self.reset()

    def reset(self):
        dev, R, cap = self.dev, self.R, self.cap
        self.row_tok = torch.zeros((R, cap), dtype=torch.int32, device=dev)
        self.row_len = torch.zeros(R, dtype=torch.int32, device=dev)       # 0: the row holds nothing
        self.row_hash = torch.zeros(R, dtype=torch.int64, device=dev)
        self.stamp = np.zeros(R, np.int64)                                   # host: call in which the row was last used
        self.t = 0
        self.stats = dict(calls=0, forward_rows=0, one_token_rows=0, encoded_rows=0, copied_rows=0, unkept_rows=0,
                          in_place_calls=0)

    # ------------------------------------------------------------------------------------------------------------
    def _lookup(self, sorted_h, order, q_hash, q_len, ctx_pad):
        """Row holding a context of hash q_hash / length q_len whose tokens equal ctx_pad[:, :q_len]; -1 if none.
        (sorted_h, order) = torch.sort(self.row_hash)."""
        idx = torch.searchsorted(sorted_h, q_hash).clamp_(max=self.R - 1)
        cand = order[idx]
        ar = torch.arange(self.cap, device=self.dev, dtype=torch.int32)
        same = ((self.row_tok[cand] == ctx_pad) | (ar[None, :] >= q_len[:, None])).all(dim=1)
        ok = (sorted_h[idx] == q_hash) & (self.row_len[cand] == q_len) & (q_len > 0) & same
        return torch.where(ok, cand, torch.full_like(cand, -1))

    @torch.no_grad()
    def logits(self, tok_d, st_d, ln_d, group_of, rep, ng, extra_head=()):
        """Next-token logits rows of the call's distinct contexts.  tok_d / st_d / ln_d: the ragged batch on the device;
        group_of, rep, ng: glb_group_contexts' output.  Returns (logits [U, V], row_of_group int32 [U] device: the
        logits row of dedup group g, group_of_row int64 [U]: its inverse, U, extra: the host values of `extra_head`'s
        device scalars - they ride on the call's one D2H copy)."""
        eng, llm, dev, R, cap = self.eng, self.llm, self.dev, self.R, self.cap
        n = ln_d.numel()
        rep_l = rep.long().clamp(0, n - 1)       # (entries past the group count are unspecified)
        L_d = ln_d[rep_l]
        ar = torch.arange(cap, device=dev, dtype=torch.int64)
        valid = ar[None, :] < L_d[:, None]
        src = (st_d[rep_l][:, None] + ar[None, :]).clamp_(max=tok_d.numel() - 1)
        ctx_pad = torch.where(valid, tok_d[src], torch.zeros((), dtype=torch.int32, device=dev))  # [n, cap] by group
        h_full = eng.hash_contexts(tok_d, st_d, ln_d)[rep_l]
        h_par = eng.hash_contexts(tok_d, st_d, (ln_d - 1).clamp_(min=0))[rep_l]
        fits = L_d <= cap
        sorted_h, by_hash = torch.sort(self.row_hash)
        exact = self._lookup(sorted_h, by_hash, h_full, torch.where(fits, L_d, torch.zeros_like(L_d)), ctx_pad)
        parent = self._lookup(sorted_h, by_hash, h_par, torch.where(fits, L_d - 1, torch.zeros_like(L_d)), ctx_pad)
        old_d = torch.where(exact >= 0, exact, parent).to(torch.int32)
        head = torch.cat([torch.stack([ng[0].to(torch.int32), *[e.to(torch.int32) for e in extra_head]]), rep.to(torch.int32),
                          L_d.to(torch.int32), old_d]).cpu().numpy()  # the call's one D2H copy before the forward
        ne = 1 + len(extra_head)
        U = int(head[0])
        extra = [int(v) for v in head[1:ne]]
        rep_h, L, old = head[ne:ne + U], head[ne + n:ne + n + U], head[ne + 2 * n:ne + 2 * n + U].copy()
        self.t += 1
        # ---- block table: who keeps its row, who gets a copy, who is encoded (host)
        has = old >= 0
        idx_has = np.nonzero(has)[0]
        _, first = np.unique(old[idx_has], return_index=True)
        keep = idx_has[np.sort(first)]
        copies = np.setdiff1d(idx_has, keep)
        fresh = np.nonzero(~has)[0]
        storable = L <= cap
        grp_row = np.full(U, -1, np.int32)
        grp_row[keep] = old[keep]
        live = np.zeros(R, bool)
        live[grp_row[keep]] = True
        free = np.nonzero(~live)[0]
        free = free[np.argsort(self.stamp[free], kind="stable")]            # longest unused first
        need = np.concatenate([copies, fresh[storable[fresh]]])               # copies first: cheaper than an encoding
        k = min(len(need), len(free))
        grp_row[need[:k]] = free[:k]
        copied = copies[grp_row[copies] >= 0]
        in_a = np.zeros(U, bool)
        in_a[keep] = True
        in_a[copied] = True
        A, B = np.nonzero(in_a)[0], np.nonzero(~in_a)[0]
        order = np.concatenate([A, B])
        inv = np.empty(U, np.int32)
        inv[order] = np.arange(U, dtype=np.int32)
        self.stamp[grp_row[grp_row >= 0]] = self.t
        st = self.stats
        st["calls"] += 1
        st["forward_rows"] += U
        st["one_token_rows"] += len(A)
        st["encoded_rows"] += len(B)
        st["copied_rows"] += len(copied)
        st["unkept_rows"] += int((grp_row[B] < 0).sum())
        to_dev = lambda a, dt=torch.int32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt, non_blocking=True)
        parts = []
        if len(A):
            if len(copied):
                src_full, len_full = np.full(R, -1, np.int32), np.zeros(R, np.int32)
                src_full[grp_row[copied]] = old[copied]
                len_full[grp_row[copied]] = L[copied] - 1
                self.pkv.copy_rows(to_dev(src_full), to_dev(len_full))
            rows_a = grp_row[A]
            if self.in_place is not None and len(A) >= self.in_place * R:
                # rows outside this forward still hold contexts the table knows: their dummy token is appended BEHIND what
                # they hold (a full row has no such place: the table forgets it)
                is_a = np.zeros(R, bool)
                is_a[rows_a] = True
                pos_full, grp_full = np.zeros(R, np.int32), np.zeros(R, np.int64)
                pos_full[rows_a] = L[A] - 1
                grp_full[rows_a] = A
                is_a_d = to_dev(is_a, torch.bool)
                self.row_len = torch.where((~is_a_d) & (self.row_len >= cap), torch.zeros_like(self.row_len), self.row_len)
                pos_d = torch.where(is_a_d, to_dev(pos_full), self.row_len.clamp(max=cap - 1))
                ids = ctx_pad[to_dev(grp_full, torch.int64), pos_d.long().clamp(max=cap - 1)].view(-1, 1).long()
                if self._slab_fwd is None or self._slab_fwd.pkv is not self.pkv:
                    from .kv import SlabForward

                    self._slab_fwd = SlabForward(self.pkv, llm._body, graph=self.graph)
                hidden = self._slab_fwd(ids, pos_d)
                parts.append(llm._lm_head(hidden[to_dev(rows_a, torch.int64)]))
                st["in_place_calls"] += 1
            else:
                pos_a = to_dev(L[A] - 1)
                ids = ctx_pad[to_dev(A, torch.int64), pos_a.long()].view(-1, 1).long()
                self.pkv.set_forward(to_dev(rows_a), pos_a)
                out = llm._body(input_ids=ids, position_ids=pos_a.view(-1, 1).long(),
                                attention_mask=self.pkv.attention_mask(pos_a), past_key_values=self.pkv, use_cache=True)
                parts.append(llm._lm_head(out.last_hidden_state[:, 0]))
        if len(B):
            sel = to_dev(rep_h[B])
            l_max = int(L[B].max())
            pad_id = getattr(llm.tokenizer, "pad_token_id", None) if llm.tokenizer is not None else None
            ids, am, pos, last = eng.gather_padded(tok_d, st_d, ln_d, sel, len(B), None, 0 if pad_id is None else pad_id, 0,
                                                   l_max)
            stored = B[grp_row[B] >= 0]
            out = llm._body(input_ids=ids, attention_mask=am, position_ids=pos, use_cache=len(stored) > 0)
            parts.append(llm._lm_head(out.last_hidden_state[torch.arange(len(B), device=dev), last.long()]))
            if len(stored):
                srcs = [(ly.keys.contiguous(), ly.values.contiguous()) for ly in out.past_key_values.layers]
                if self.pkv is None:
                    from .kv import SharedSlabKV

                    self.pkv = SharedSlabKV(eng, R, cap, len(srcs))
                src_full, len_full = np.full(R, -1, np.int32), np.zeros(R, np.int32)
                where_b = np.full(U, -1, np.int32)
                where_b[B] = np.arange(len(B), dtype=np.int32)
                src_full[grp_row[stored]] = where_b[stored]
                len_full[grp_row[stored]] = L[stored]
                self.pkv.fill_rows(srcs, to_dev(src_full), to_dev(len_full))
        # ---- the table rows of everything that holds a context now
        held = np.nonzero(grp_row >= 0)[0]
        if len(held):
            rows_d, grp_d = to_dev(grp_row[held], torch.int64), to_dev(held, torch.int64)
            self.row_tok[rows_d] = ctx_pad[grp_d]
            self.row_len[rows_d] = L_d[grp_d].to(torch.int32)
            self.row_hash[rows_d] = h_full[grp_d]
        logits = parts[0] if len(parts) == 1 else torch.cat(parts)
        return logits, to_dev(inv), to_dev(order, torch.int64), U, extra
