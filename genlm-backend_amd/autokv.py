"""KV rows that follow the contexts of a stateless API.

`AsyncAmdLM.batch_next_token_step` is handed plain token lists, step after step; a population that grows by one token
per step asks for contexts whose first L - 1 tokens were evaluated the step before.  The reference re-encodes them in
full unless the caller pinned a prompt with `cache_kv` (hf.py:155-164); its MLX backend keeps per-token KV on the
trie's nodes instead (mlx.py:177-318, cache.py:103-191).  `AutoKV` is that idea on slab rows: R rows of `cap` positions
(`kv.SharedSlabKV`), a device table of the context every row holds (tokens, length, hash), and per call

  * a lookup of every distinct context - the row that holds exactly it, else the row that holds its first L - 1 tokens
    (glb_match_rows: hashes select candidates, every candidate's tokens are compared: a hash never decides alone);
  * contexts with a row feed ONE token (the first to claim a row keeps it, the others get a copy of the prefix in a row
    nobody used for the longest time: copy-on-append); contexts without one are encoded from their tokens and their
    KV kept if a row is to be had;
  * the table rows of everything that now holds a context are rewritten.

Lookup and block table run on the device (glb_match_rows, glb_kv_plan: one launch each), like `DeviceSIS._step_shared_kv`;
the host reads the numbers of rows of each kind from the call's one D2H copy and launches the forwards.
"""
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
        self.stamp = torch.zeros(R, dtype=torch.int64, device=dev)          # call in which the row was last used
        self.t = 0
        self.stats = dict(calls=0, forward_rows=0, one_token_rows=0, encoded_rows=0, copied_rows=0, unkept_rows=0,
                          in_place_calls=0)

    @torch.no_grad()
    def logits(self, tok_d, st_d, ln_d, group_of, rep, ng, extra_head=()):
        """Next-token logits rows of the call's distinct contexts.  tok_d / st_d / ln_d: the ragged batch on the device;
        group_of, rep, ng: glb_group_contexts' output.  Returns (logits [U, V], row_of_group int32 [n] device: the
        logits row of dedup group g, group_of_row int64 [U]: its inverse, U, extra: the host values of `extra_head`'s
        device scalars - they ride on the call's one D2H copy)."""
        eng, llm, dev, R, cap = self.eng, self.llm, self.dev, self.R, self.cap
        # the row that holds every distinct context, or its first L - 1 tokens (every candidate's tokens are compared:
        # a hash never decides alone), then the block table - both on the device, one launch each
        old, gh = eng.match_rows(tok_d, st_d, ln_d, rep, ng, self.row_tok, self.row_len, self.row_hash)
        self.t += 1
        plan = eng.kv_plan(group_of, rep, ng, old, ln_d, R, cap, stamps=self.stamp, call_no=self.t,
                           table=(self.row_tok, self.row_len, self.row_hash, gh, tok_d, st_d))
        head = torch.cat([plan["head"][:6], *[e.to(torch.int32).view(1) for e in extra_head]]).cpu().tolist()  # the one D2H copy
        U, nA, nB, n_copied, n_unkept, l_max_b = head[:6]
        extra = head[6:]
        st = self.stats
        st["calls"] += 1
        st["forward_rows"] += U
        st["one_token_rows"] += nA
        st["encoded_rows"] += nB
        st["copied_rows"] += n_copied
        st["unkept_rows"] += n_unkept
        parts = []
        if nA:
            if n_copied:
                self.pkv.copy_rows(plan["copy_src"], plan["copy_len"])
            if self.in_place is not None and nA >= self.in_place * R:
                # rows outside this forward still hold contexts the table knows: their dummy token is appended BEHIND what
                # they hold (a full row has no such place: the table forgets it); rows being filled from an encoding take
                # theirs at position 0, which the fill overwrites
                ctx_r = plan["ctx_of_row"]
                is_a, idle = ctx_r >= 0, ctx_r == -1
                self.row_len.masked_fill_(idle & (self.row_len >= cap), 0)
                zero = torch.zeros_like(ctx_r)
                pos_d = torch.where(is_a, plan["pos_of_row"], torch.where(idle, self.row_len.clamp(max=cap - 1), zero))
                at = st_d[ctx_r.clamp_min(0).long()] + torch.where(is_a, pos_d, zero).long()
                ids = tok_d[at].view(-1, 1).long()
                if self._slab_fwd is None or self._slab_fwd.pkv is not self.pkv:
                    from .kv import SlabForward

                    self._slab_fwd = SlabForward(self.pkv, llm._body, graph=self.graph, owner=llm)
                hidden = self._slab_fwd(ids, pos_d)
                parts.append(llm._lm_head(hidden.index_select(0, plan["rows_a"][:nA].long())))
                st["in_place_calls"] += 1
            else:
                pos_a = plan["pos_a"][:nA].contiguous()
                ids = tok_d[st_d[plan["ctx_a"][:nA].long()] + pos_a.long()].view(-1, 1).long()
                self.pkv.set_forward(plan["rows_a"][:nA].contiguous(), pos_a)
                out = llm._body(input_ids=ids, position_ids=pos_a.view(-1, 1).long(),
                                attention_mask=self.pkv.attention_mask(pos_a), past_key_values=self.pkv, use_cache=True)
                parts.append(llm._lm_head(out.last_hidden_state[:, 0]))
        if nB:
            sel = plan["ctx_b"][:nB].contiguous()
            pad_id = getattr(llm.tokenizer, "pad_token_id", None) if llm.tokenizer is not None else None
            ids, am, pos, last = eng.gather_padded(tok_d, st_d, ln_d, sel, nB, None, 0 if pad_id is None else pad_id, 0,
                                                   l_max_b)
            stored = nB > n_unkept
            out = llm._body(input_ids=ids, attention_mask=am, position_ids=pos, use_cache=stored)
            parts.append(llm._lm_head(out.last_hidden_state[torch.arange(nB, device=dev), last.long()]))
            if stored:
                srcs = [(ly.keys.contiguous(), ly.values.contiguous()) for ly in out.past_key_values.layers]
                if self.pkv is None:
                    from .kv import SharedSlabKV

                    self.pkv = SharedSlabKV(eng, R, cap, len(srcs))
                rows_b = plan["rows_b"][:nB].long()
                slot = torch.where(rows_b >= 0, rows_b, torch.full_like(rows_b, R))  # (rows nobody keeps: a slot past the end)
                src_full = torch.full((R + 1,), -1, dtype=torch.int32, device=dev)
                len_full = torch.zeros(R + 1, dtype=torch.int32, device=dev)
                src_full[slot] = torch.arange(nB, dtype=torch.int32, device=dev)
                len_full[slot] = ln_d[sel.long()]
                src_full[R] = -1
                self.pkv.fill_rows(srcs, src_full[:R].contiguous(), len_full[:R].contiguous())
        logits = parts[0] if len(parts) == 1 else torch.cat(parts)
        row_of_group = plan["logits_row"]
        group_of_row = torch.empty(U, dtype=torch.int64, device=dev)
        group_of_row[row_of_group[:U].long()] = torch.arange(U, device=dev)
        return logits, row_of_group, group_of_row, U, extra
