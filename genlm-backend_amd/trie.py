"""Token -> byte trie with batched mass propagation on the MI355X.

Counterpart of the reference's `TokenCharacterTrie` / `ParallelTokenCharacterTrie` (genlm/backend/trie/base.py:10-213,
trie/parallel.py:33-145): a trie over the byte strings of the vocabulary with one extra leaf per token;
`weight_sum` / `weight_max` give, for every node, the sum / maximum of the weights of the tokens below it.  The
structure (node numbering, `children`, `word2leaf`, `leaf2word`, `node2prefix`, `jump`) is the reference's, so node ids
mean the same thing; the propagation runs for a whole batch of weight rows at once on the device (glb_trie_reduce: one launch per
tree level) instead of a numba loop per row (base.py:346-393) or a sparse matmul (parallel.py:92-145).
"""
import numpy as np
import torch

from .tokenization import Token


class TokenByteTrie:
    def __init__(self, decode, engine=None):
        """decode: the vocabulary - `Token`s from `decode_vocab`, plain bytes, or other iterables of symbols; the weight
        of token k is column k of a weight row.  engine: a `HipEngine` (needed for the weight_* methods)."""
        self.decode = decode
        self.engine = engine
        # -- build: nodes in creation order, every node keeps its out-edges in insertion order; a token ends in a leaf
        #    of its own hanging off the node of its last byte through the edge (None, k)          (base.py:13-62)
        edges = [[]]          # per node: [(symbol, child)]
        index = [{}]          # per node: symbol -> child
        leaf_of, keys = [], []
        warned = False
        for k, item in enumerate(decode):
            if isinstance(item, Token):
                word, key = item.byte_string, (item.byte_string, item.token_id)
            else:
                if isinstance(item, (bytes, bytearray)) and not warned:  # (base.py:34-42: once per trie)
                    import warnings

                    warnings.warn("Passing plain bytes to TokenByteTrie is deprecated. Use Token objects from "
                                  "decode_vocab() instead.", DeprecationWarning, stacklevel=2)
                    warned = True
                word, key = item, item
            at = 0
            for sym in word:
                nxt = index[at].get(sym)
                if nxt is None:
                    nxt = len(edges)
                    index[at][sym] = nxt
                    edges[at].append((sym, nxt))
                    edges.append([])
                    index.append({})
                at = nxt
            leaf = len(edges)
            edges[at].append(((None, k), leaf))
            edges.append([])
            index.append({})
            leaf_of.append(leaf)
            keys.append(key)
        if len(set(keys)) != len(keys):
            seen = set()
            dup = next(x for x in keys if x in seen or seen.add(x))
            raise ValueError(f"Duplicate word in vocabulary: {dup}")
        # -- renumber in post-order (children before their parent, edges in insertion order; the root comes last), the
        #    reference's memory-locality numbering (base.py:72-79,225-243)
        n = len(edges)
        new_id = np.empty(n, np.int64)
        counter = 0
        stack = [(0, 0)]
        while stack:
            node, i = stack.pop()
            if i < len(edges[node]):
                stack.append((node, i + 1))
                stack.append((edges[node][i][1], 0))
            else:
                new_id[node] = counter
                counter += 1
        self.root = int(new_id[0])
        self.children = [None] * n
        for old in range(n):
            self.children[new_id[old]] = {sym: int(new_id[c]) for sym, c in edges[old]}
        self.word2leaf = {key: int(new_id[leaf]) for key, leaf in zip(keys, leaf_of)}
        self.leaf2word = {v: k for k, v in self.word2leaf.items()}
        self.idx_to_leaf = np.array([(k, new_id[leaf]) for k, leaf in enumerate(leaf_of)], dtype=np.int32).reshape(-1, 2)
        self.jump = [np.array(sorted(c.values()), dtype=np.int32) for c in self.children]
        # prefixes: parents have larger ids than their children, so a descending sweep sees a parent first
        self.node2prefix = {self.root: []}
        for x in range(n - 1, -1, -1):
            for sym, y in self.children[x].items():
                is_leaf_edge = isinstance(sym, tuple) and sym[0] is None
                self.node2prefix[y] = self.node2prefix[x] if is_leaf_edge else self.node2prefix[x] + [sym]
        self._flat = None
        self._dev = None
        self._compact = None
        self._cdev = None
        self._plans = {}
        self._pdevs = {}
        self._tree_cache = None
        self._sel_plans = {}

    def __len__(self):
        return len(self.children)

    # ---- flattened form for the kernel --------------------------------------------------------------------------
    def flat(self):
        """Host arrays of glb_trie_reduce: leaf of every token, internal nodes by level (level 0 = all children are
        leaves; a node sits one level above its highest internal child) and the CSR of ascending children."""
        if self._flat is None:
            n = len(self.children)
            counts = np.fromiter((len(j) for j in self.jump), np.int64, n)
            child_ptr = np.zeros(n + 1, np.int64)
            np.cumsum(counts, out=child_ptr[1:])
            child_idx = np.concatenate(self.jump) if n else np.zeros(0, np.int32)
            level = np.full(n, -1, np.int64)
            for x in range(n):  # ascending ids = children first
                if counts[x]:
                    level[x] = 1 + max(-1, int(level[self.jump[x]].max()))
            internal = np.nonzero(level >= 0)[0]
            order = internal[np.argsort(level[internal], kind="stable")]
            n_levels = int(level.max()) + 1
            level_start = np.searchsorted(level[order], np.arange(n_levels + 1))
            self._flat = dict(vocab=len(self.decode), n_nodes=n, n_levels=n_levels,
                              leaf_node=self.idx_to_leaf[:, 1].astype(np.int32), level_start=level_start.astype(np.int32),
                              level_nodes=order.astype(np.int32), child_ptr=child_ptr.astype(np.int32),
                              child_idx=child_idx.astype(np.int32))
        return self._flat

    def compact(self):
        """The same trie with its one-child nodes folded away, for the kernel.  A node with exactly one child has that
        child's value - the sum or the maximum of one term, bit for bit - so it needs neither storage nor a pass of its
        own: `slot_of[node]` names the slot (a leaf or a node with two children and more, numbered in ascending node
        order) whose value it shares, and the arrays describe the tree over the slots (children in the original
        ascending order, so every sum adds the same numbers in the same order).  A byte trie of a BPE vocabulary is
        mostly such chains - 194 k nodes, 68 k slots for the 50 257 synthetic tokens of tools/tbench.py - and the
        propagation moves a third of the bytes."""
        if self._compact is None:
            n = len(self.children)
            counts = np.fromiter((len(j) for j in self.jump), np.int64, n)
            own = counts != 1
            slot_of = np.full(n, -1, np.int64)
            slot_of[own] = np.arange(int(own.sum()))
            first = np.fromiter((j[0] if len(j) else -1 for j in self.jump), np.int64, n)
            for x in np.nonzero(~own)[0]:  # ascending ids = children first
                slot_of[x] = slot_of[first[x]]
            owners = np.nonzero(own)[0]
            n_slots = len(owners)
            ccount = np.where(counts[owners] >= 2, counts[owners], 0)
            child_ptr = np.zeros(n_slots + 1, np.int64)
            np.cumsum(ccount, out=child_ptr[1:])
            kids = [slot_of[self.jump[x]] for x in owners if counts[x] >= 2]
            child_idx = np.concatenate(kids) if kids else np.zeros(0, np.int64)
            level = np.full(n_slots, -1, np.int64)
            for s_, x in enumerate(owners):  # ascending
                if counts[x] >= 2:
                    level[s_] = 1 + max(-1, int(level[slot_of[self.jump[x]]].max()))
            internal = np.nonzero(level >= 0)[0]
            order = internal[np.argsort(level[internal], kind="stable")]
            n_levels = int(level.max()) + 1 if len(internal) else 0
            level_start = np.searchsorted(level[order], np.arange(n_levels + 1))
            self._compact = dict(vocab=len(self.decode), n_nodes=n_slots, n_levels=n_levels,
                                 leaf_node=slot_of[self.idx_to_leaf[:, 1]].astype(np.int32),
                                 level_start=level_start.astype(np.int32), level_nodes=order.astype(np.int32),
                                 child_ptr=child_ptr.astype(np.int32), child_idx=child_idx.astype(np.int32),
                                 slot_of=slot_of.astype(np.int32))
        return self._compact

    PLAN_CAP = 9500  # slots per part: ~6.5 bytes of LDS a slot (value, child pointer, list of internal nodes); three 512-thread workgroups a CU
    # (tools/dbg/stamps_trie.py at 1024 x 50257: parts of <= 5267 / 7875 / 10481 / 23440 slots: 232 / 206 / 225 / 225 us)

    SWEEP_LOCAL_MAX = 40000  # slots per part of a sweep plan: 4 bytes of LDS a slot, one 1024-thread workgroup a CU (160 KB)

    def _count_parts(self, cap):
        """How many parts `_build_plan(cap, None)` would make (the cut and the first-fit packing alone)."""
        kids, size, root = self._tree()
        cut, stack = [], [root]
        while stack:
            s = stack.pop()
            if size[s] <= cap:
                cut.append(int(size[s]))
            else:
                stack.extend(kids[s].tolist())
        room = []
        for sz in sorted(cut, reverse=True):
            for b in range(len(room)):
                if room[b] >= sz:
                    room[b] -= sz
                    break
            else:
                room.append(cap - sz)
        return len(room)

    def sweep_cap(self):
        """Slots per part for `plan(sweep=True)`: as few parts as the LDS allows (every part reads the whole row once), of
        about equal size - the smallest cap that packs the subtrees into that many parts."""
        n_slots = int(self.compact()["n_nodes"])
        n_parts = max(1, -(-n_slots // self.SWEEP_LOCAL_MAX))
        while True:
            ideal = -(-n_slots // n_parts)
            for num in (102, 105, 110, 120, 135):
                cap = min(self.SWEEP_LOCAL_MAX, ideal * num // 100 + 64)
                if self._count_parts(cap) <= n_parts:
                    return cap
            n_parts += 1

    def plan(self, cap=None, sweep=False):
        """The folded trie (`compact()`) cut for glb_trie_rows, which keeps ONE ROW's values of a part of the trie in
        the LDS of a compute unit: subtrees of at most `cap` slots are packed into parts of at most `cap` slots; the few
        nodes above them (the root and what else is too big - `top`) form one more part whose leaves are the cut
        subtrees' roots.  Inside a part the slots are numbered breadth first over its subtrees (roots, then their
        children in the original ascending order, ...), so a node's children are consecutive - `cptr[s] .. cptr[s + 1]` -
        and a depth is a range `depth_start[d] .. depth_start[d + 1]`; every sum still adds the same numbers in the same
        order as the reference's loop (base.py:346-393).  Slots are renumbered part by part (`slot_of`: node -> new
        slot; the top's nodes come last), so a part's values are one run of a row of the slot-major output.
        Returns None when the top does not fit a part (a trie with a node of more than `cap` children).
        sweep=True: the plan of the kernel that reads a row front to back instead of gathering a part's tokens from it
        (round 5): parts as big as the LDS holds when only the VALUES live there (`sweep_cap`), and two more tables -
        `tok_local16[part, token]` (the token's local slot in that part; another part's token: one of the 32 slack words behind
        the part's values, n_local + (token // 8) % 32; rows padded to 8 tokens)
        and `inode64` (a part's internal nodes depth by depth as `inode16` lists them: slot | first child << 16 |
        children << 32), which the kernel streams from global memory."""
        cap = int(cap or (self.sweep_cap() if sweep else self.PLAN_CAP))
        key = (cap, bool(sweep))
        if key not in self._plans:
            self._plans[key] = self._build_plan(cap, None, sweep=sweep)
        return self._plans[key]

    def _tree(self):
        """The folded trie as Python sees it: children of every slot, subtree sizes, the root's slot."""
        if self._tree_cache is None:
            c = self.compact()
            n_slots = int(c["n_nodes"])
            cp, ci = c["child_ptr"].astype(np.int64), c["child_idx"].astype(np.int64)
            kids = [ci[cp[s]:cp[s + 1]] for s in range(n_slots)]
            size = np.ones(n_slots, np.int64)
            for s in range(n_slots):  # ascending slots = children first
                if len(kids[s]):
                    size[s] += size[kids[s]].sum()
            self._tree_cache = (kids, size, int(c["slot_of"][self.root]))
        return self._tree_cache

    def _build_plan(self, cap, sel_roots, sweep=False):
        """`plan()` for the forest below `sel_roots` (folded-trie slots, none an ancestor of another; None: the whole trie):
        only those subtrees are cut into parts, read and reduced - what the masses of a SELECTION of nodes need (the
        selection's maximal nodes are the roots).  Slots outside the forest get slot -1."""
        c = self.compact()
        n_slots = int(c["n_nodes"])
        kids, size, root = self._tree()
        forest = [root] if sel_roots is None else sorted(int(r) for r in sel_roots)
        # -- the cut: descend from the roots while a subtree is too big for a part
        top, cut = [], []
        stack = list(forest)
        while stack:
            s = stack.pop()
            if size[s] <= cap:
                cut.append(s)
            else:
                top.append(s)
                stack.extend(kids[s].tolist())
        top.sort()
        cut.sort()
        if cap >= 65536:
            return None
        # -- parts: first fit, biggest subtrees first
        bins, room = [], []
        for s in sorted(cut, key=lambda s: -size[s]):
            for b in range(len(bins)):
                if room[b] >= size[s]:
                    bins[b].append(s)
                    room[b] -= size[s]
                    break
            else:
                bins.append([s])
                room.append(cap - size[s])
        n_parts = len(bins)
        DESC = 16
        desc = np.zeros((n_parts + 1, DESC), np.int32)
        cptr_all, depth_all, leaf_src, leaf_local = [], [], [], []
        new_slot = np.full(n_slots, -1, np.int64)
        part_of = np.full(n_slots, -1, np.int64)
        local_of = np.full(n_slots, -1, np.int64)
        cut_index = {}
        slot_base = 0
        tok_slot = c["leaf_node"].astype(np.int64)  # compact slot of every token's leaf

        def bfs(roots, is_leaf_here):
            order = list(roots)
            cptr = []
            depth_start = [0]
            lo, nxt = 0, len(order)
            while lo < len(order):
                hi = len(order)
                for i in range(lo, hi):
                    cptr.append(len(order))
                    if not is_leaf_here(order[i]):
                        order.extend(kids[order[i]].tolist())
                depth_start.append(hi)
                lo = hi
            cptr.append(len(order))
            return order, cptr, depth_start

        for p, roots in enumerate(bins):
            roots = sorted(roots)
            order, cptr, depth_start = bfs(roots, lambda s: False)
            order = np.asarray(order, np.int64)
            part_of[order] = p
            local_of[order] = np.arange(len(order))
            new_slot[order] = slot_base + np.arange(len(order))
            for i, s in enumerate(roots):
                cut_index[s] = len(cut_index)
            desc[p, 0], desc[p, 1], desc[p, 2], desc[p, 3] = slot_base, len(order), len(roots), len(depth_start) - 1
            desc[p, 4], desc[p, 5] = sum(len(d) for d in depth_all), sum(len(x) for x in cptr_all)
            desc[p, 8] = cut_index[roots[0]]
            depth_all.append(depth_start)
            cptr_all.append(cptr)
            slot_base += len(order)
        # leaves of the parts: tokens in ascending order with their local slots
        tok_part = part_of[tok_slot]
        for p in range(n_parts):
            toks = np.nonzero(tok_part == p)[0]
            desc[p, 6], desc[p, 7] = sum(len(x) for x in leaf_src), len(toks)
            leaf_src.append(toks)
            leaf_local.append(local_of[tok_slot[toks]])
        # -- the top: one more part, its leaves are the cut roots
        top_set, T = set(top), n_parts
        n_top = len(top)
        toplocal = np.zeros(0, np.int64)
        if n_top:
            order, cptr, depth_start = bfs([s for s in forest if s in top_set], lambda s: s not in top_set)
            if len(order) > cap:  # the top and the cut roots below it have to fit a part
                return None
            order = np.asarray(order, np.int64)
            is_top = np.fromiter((s in top_set for s in order), bool, len(order))
            top_order = order[is_top]  # breadth first; new slots of the top follow this order
            new_slot[top_order] = slot_base + np.arange(n_top)
            part_of[top_order] = T
            toplocal = np.nonzero(is_top)[0]
            local_of[top_order] = toplocal
            leaves = np.nonzero(~is_top)[0]
            desc[T, 0], desc[T, 1], desc[T, 2], desc[T, 3] = slot_base, len(order), sum(s in top_set for s in forest), len(depth_start) - 1
            desc[T, 4], desc[T, 5] = sum(len(d) for d in depth_all), sum(len(x) for x in cptr_all)
            desc[T, 6], desc[T, 7] = sum(len(x) for x in leaf_src), len(leaves)
            depth_all.append(depth_start)
            cptr_all.append(cptr)
            leaf_src.append(np.asarray([cut_index[int(s)] for s in order[leaves]], np.int64))
            leaf_local.append(leaves)
        # every node's new slot; the nodes of a part in ascending order with their local slots
        node_slot = c["slot_of"].astype(np.int64)
        slot_of_new = new_slot[node_slot]
        node_part = part_of[node_slot]
        if sel_roots is not None:  # (a selection's plan writes selected nodes only: no node lists)
            node_part = np.full_like(node_part, -1)
        pn_node, pn_local = [], []
        for p in range(n_parts + 1):
            nodes = np.nonzero(node_part == p)[0]
            desc[p, 9], desc[p, 10] = sum(len(x) for x in pn_node), len(nodes)
            pn_node.append(nodes)
            pn_local.append(local_of[node_slot[nodes]])
        cat = lambda xs: (np.concatenate([np.asarray(x, np.int64) for x in xs]) if xs else np.zeros(0, np.int64)).astype(np.int32)
        # a part's nodes are a few runs of consecutive ids (post-order: a subtree is an interval, and the one-child nodes
        # folded into its root follow it): (first node, count) per run; the local slots as 16-bit words
        run_tab = []
        for p in range(n_parts + 1):
            nd = np.asarray(pn_node[p], np.int64)
            cuts = np.nonzero(np.diff(nd) != 1)[0] + 1 if len(nd) else np.zeros(0, np.int64)
            b = np.concatenate([[0], cuts, [len(nd)]]) if len(nd) else np.zeros(1, np.int64)
            desc[p, 14], desc[p, 15] = len(run_tab), len(b) - 1
            run_tab.extend((int(nd[b[k]]), int(b[k + 1] - b[k])) for k in range(len(b) - 1))
        # what the kernel keeps in LDS next to the values, as 16-bit words (a part has fewer than 65536 slots): the child
        # pointers, and the part's internal nodes depth by depth (within a depth the nodes with the most children first:
        # the lanes of a wave then run loops of about the same length)
        cptr16, inode16, idepth, inode64 = [], [], [], []
        lds_bytes = lds_top = 0
        for p in range(len(cptr_all)):
            cp = np.asarray(cptr_all[p], np.int64)
            ds = np.asarray(depth_all[p], np.int64)
            nch = np.diff(cp)
            desc[p, 5] = sum(len(x) for x in cptr16)          # (offsets in 16-bit words, even)
            cptr16.append(np.concatenate([cp, np.zeros(len(cp) & 1, np.int64)]))
            ins, idp = [], [0]
            for k in range(len(ds) - 1):
                sl = np.arange(ds[k], ds[k + 1])
                sl = sl[nch[sl] > 0]
                ins.append(sl[np.argsort(-nch[sl], kind="stable")])
                idp.append(idp[-1] + len(sl))
            ins = np.concatenate(ins) if ins else np.zeros(0, np.int64)
            desc[p, 11], desc[p, 12], desc[p, 13] = sum(len(x) for x in inode16), len(ins), sum(len(x) for x in idepth)
            inode16.append(np.concatenate([ins, np.zeros(len(ins) & 1, np.int64)]))
            inode64.append(np.concatenate([ins | (cp[ins] << 16) | (nch[ins] << 32), np.zeros(len(ins) & 1, np.int64)]))
            idepth.append(idp)
            if len(idp) > 31:  # (the kernel keeps a part's depth table in 32 words of LDS)
                return None
            resident = 4 * int(desc[p, 1]) + 2 * len(cptr16[-1]) + 2 * len(inode16[-1]) + 4 * 32
            if sweep and p < n_parts:  # (a swept part keeps its values and 32 words of slack)
                lds_bytes = max(lds_bytes, 4 * int(desc[p, 1]) + 4 * 32)
            else:
                lds_bytes = max(lds_bytes, resident)
            if p == n_parts:
                lds_top = resident
        cat16 = lambda xs: np.concatenate(list(xs) + [np.zeros(2, np.int64)]).astype(np.uint16)  # (never empty: a device pointer)
        extra = {}
        if sweep and lds_bytes > 160 * 1024:
            return None
        if sweep:
            V = len(self.decode)
            vp = (V + 7) & ~7
            tl = np.zeros((max(n_parts, 1), vp), np.uint16)
            for p in range(n_parts):  # (another part's token: a word of the 32-word slack behind the values - the kernel stores every token)
                tl[p] = (int(desc[p, 1]) + ((np.arange(vp) >> 3) & 31)).astype(np.uint16)
                tl[p, leaf_src[p]] = np.asarray(leaf_local[p], np.int64).astype(np.uint16)
            extra = dict(tok_local16=tl.reshape(-1), inode64=np.concatenate(inode64 + [np.zeros(2, np.int64)]).astype(np.uint64))
        plan = dict(n_parts=n_parts, n_top=n_top, sweep=bool(sweep), lds_top_bytes=lds_top, **extra, n_slots=n_slots if sel_roots is None else slot_base + n_top, n_cut=len(cut), cap=cap,
                    vocab=len(self.decode),
                    n_nodes=len(self.children), max_local=int(desc[:, 1].max()), top_base=slot_base,
                    lds_bytes=lds_bytes, cptr16=cat16(cptr16), inode16=cat16(inode16), idepth=cat(idepth),
                    desc=desc, depth_start=cat(depth_all), cptr=cat(cptr_all), leaf_src=cat(leaf_src),
                    leaf_local=cat(leaf_local), pn_node=cat(pn_node), pn_local=cat(pn_local),
                    run_tab=np.asarray(run_tab, np.int32).reshape(-1, 2), pn_local16=cat(pn_local).astype(np.uint16),
                    top_local=toplocal.astype(np.int32), slot_of=slot_of_new.astype(np.int32),
                    slot_compact=np.argsort(new_slot).astype(np.int32))  # new slot -> compact() slot
        return plan

    def prepare_selection(self, nodes):
        """Plan the sub-forest below a selection of nodes that will be asked for again and again (`masses_from_logits(nodes=
        this tensor)` then reads and reduces only those subtrees: 4096 nodes of a 50 k-token trie 179 -> 78 us a call).  Host
        work - a device-to-host copy of the ids (a stream synchronisation), a plan of the sub-forest in Python, some twenty
        small uploads: tens of milliseconds - which is why it is asked for, not done behind a call's back: a selection that
        changes every step (the children of each particle's current node) belongs to the per-row form (`nodes` [B, K]),
        whose need mask is made on the device.  Returns True when a pruned plan is in place (False: the selection reaches
        the root or covers half the trie - the whole trie's plan serves it)."""
        got = self.selection_plan(nodes, build=True) is not None
        if got and self.sweep:  # big batches read the rows front to back: the sub-forest's sweep plan beside the gathered one
            self.selection_plan(nodes, build=True, sweep=True)
        return got

    def selection_plan(self, nodes, build=False, sweep=False):
        """The plan of the sub-forest a selection of nodes needs (`_build_plan`), on the device, cached per selection tensor
        (its storage and version).  None: use the whole trie's plan (the selection reaches the root, has no plan, or - unless
        `build` - was never prepared: `prepare_selection`).  Ids outside the trie (negative: "none", as the kernel reads
        them) select nothing."""
        key = (nodes.data_ptr(), nodes._version, nodes.numel(), bool(sweep))
        ent = self._sel_plans.get(key)
        if ent is None and not build:
            return None
        if ent is None:
            if len(self._sel_plans) >= 16:
                self._sel_plans.pop(next(iter(self._sel_plans)))
            c = self.compact()
            kids, size, root = self._tree()
            ids = nodes.cpu().numpy().astype(np.int64).reshape(-1)
            ids = ids[(ids >= 0) & (ids < len(c["slot_of"]))]
            sel = np.unique(c["slot_of"].astype(np.int64)[ids])
            # the maximal selected slots: drop every slot that lies below another selected one (slots ascend children first,
            # so a subtree is the slot range (s - size[s], s])
            sel_desc = sel[::-1]
            roots, lo_bound = [], None
            for s_ in sel_desc:
                if lo_bound is not None and s_ > lo_bound:
                    continue  # inside the last root's subtree
                roots.append(int(s_))
                lo_bound = int(s_) - int(size[s_])
            pl = None
            if sum(int(size[r]) for r in roots) < 0.5 * int(c["n_nodes"]) and root not in roots:
                host = self._build_plan(self.sweep_cap() if sweep else self.PLAN_CAP, roots, sweep=sweep)
                if host is not None:
                    dev = self.engine.device
                    signed = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64}
                    pl = {k: (torch.from_numpy(v.view(signed.get(v.dtype, v.dtype))).to(dev) if isinstance(v, np.ndarray) else v)
                          for k, v in host.items()}
            ent = self._sel_plans[key] = (pl, nodes)  # (holds the tensor: its address cannot be handed to another selection)
        return ent[0]

    def plan_device_arrays(self, cap=None, sweep=False):
        """`plan()` on the device (None when the trie has no usable plan)."""
        pl = self.plan(cap, sweep=sweep)
        if pl is None:
            return None
        key = (pl["cap"], bool(sweep))
        if key not in self._pdevs:
            if self.engine is None:
                raise RuntimeError("TokenByteTrie needs a HipEngine to compute masses (there is no CPU path)")
            dev = self.engine.device
            signed = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64}
            self._pdevs[key] = {k: (torch.from_numpy(v.view(signed.get(v.dtype, v.dtype))).to(dev)
                                    if isinstance(v, np.ndarray) else v) for k, v in pl.items()}
        return self._pdevs[key]

    def _to_device(self, f):
        if self.engine is None:
            raise RuntimeError("TokenByteTrie needs a HipEngine to compute masses (there is no CPU path)")
        dev = self.engine.device
        d = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        d["level_start_host"] = np.ascontiguousarray(f["level_start"], dtype=np.int32)
        return d

    def device_arrays(self):
        if self._dev is None:
            self._dev = self._to_device(self.flat())
        return self._dev

    def compact_device_arrays(self):
        """`compact()` on the device; ["slot_of"]: int32 [n_nodes], the slot that holds a node's value."""
        if self._cdev is None:
            self._cdev = self._to_device(self.compact())
        return self._cdev

    # ---- masses -----------------------------------------------------------------------------------------------------
    def _rows(self, ws):
        if isinstance(ws, (list, tuple)):
            ws = torch.stack([torch.as_tensor(w, dtype=torch.float32) for w in ws])
        ws = torch.as_tensor(ws)
        if ws.dim() == 1:
            ws = ws[None]
        if ws.shape[1] != len(self.decode):
            raise ValueError(f"weight rows have {ws.shape[1]} columns, vocabulary has {len(self.decode)}")
        return ws.to(self.engine.device, torch.float32).contiguous()

    _COMPACT_ROWS = 32  # from here on the kernels keep the values node-major: the folded trie pays

    # masses of selected nodes ([len(nodes)] for every row) through a plan of only the sub-forest below them: "cached" (the
    # default) - when the caller prepared this selection (`prepare_selection(nodes)`: the host work is asked for, never
    # done behind a call's back), else the whole trie's plan; True - planned at the first call that meets a selection
    # (synchronises the stream once per new selection tensor); False - never
    prune_selection = "cached"
    sweep = True  # whole-trie masses through the plan whose parts read a row front to back (plan(sweep=True)): the slots always
    # (their numbering is that plan's: `slot_plan()`; 1024 rows 193 -> 140 us, 8 rows 17 -> 21), all nodes from SWEEP_MIN_ROWS
    # rows on (1024: 309 -> 273 us, 512: 158 -> 146; below, the gathered plan's many small workgroups fill the chip better:
    # 256 rows 63 against 68 us, one row 18 against 30 - tools/dbg/sweep_probe2.py)
    SWEEP_MIN_ROWS = 512
    resident = True  # glb_trie_rows (a row of a part of the trie in LDS) when the trie has a plan; False: the level-synchronous kernels

    def _batch(self, ws, op, from_logprobs):
        ws = self._rows(ws)
        pl = self._whole_plan_device(ws.shape[0]) if self.resident else None
        if pl is not None:
            got = self._trie_rows(ws, pl, op, from_logprobs)
            if got is not None:
                return got
        if ws.shape[0] < self._COMPACT_ROWS or self.compact()["n_levels"] == 0:
            return self.engine.trie_reduce(ws, self.device_arrays(), op, from_logprobs)
        c = self.compact_device_arrays()
        return self.engine.trie_masses(ws, c, op, from_logprobs, nodes=c["slot_of"])

    def slot_plan(self):
        """The plan whose slot numbering layout "slot_rows" uses (`["slot_of"]`: node -> column): the sweep plan when the
        trie has one and `sweep` is on, else `plan()`."""
        return (self.plan(sweep=True) if self.sweep else None) or self.plan()

    def _whole_plan_device(self, n_rows, slots=False):
        """The device plan for masses of the whole trie: the sweep plan for the slots (one numbering whatever the batch) and
        for big batches, the gathered plan otherwise."""
        if self.sweep and (slots or n_rows >= self.SWEEP_MIN_ROWS):
            pl = self.plan_device_arrays(sweep=True)
            if pl is not None:
                return pl
        return self.plan_device_arrays()

    def _trie_rows(self, ws, pl, *args, **kw):
        """glb_trie_rows, or None when the device cannot run it (a part of the plan needs more LDS than the device has - a
        build for another architecture): the level-synchronous kernels serve the trie from then on."""
        from ._lib import GLB_EHIP, GlbError

        try:
            return self.engine.trie_rows(ws, pl, *args, **kw)
        except GlbError as e:
            if e.code != GLB_EHIP:
                raise
            self.resident = False
            return None

    def batch_weight_sum_device(self, ws, from_logprobs=False):
        """[B, V] weights (or log-probabilities) -> float32 [B, n_nodes] on the device."""
        return self._batch(ws, 0, from_logprobs)

    def batch_weight_max_device(self, ws, from_logprobs=False):
        return self._batch(ws, 1, from_logprobs)

    def masses_from_logits(self, logits, lse=None, nodes=None, layout="rows", op=0, logit_scale=1.0, wide_selections=False):
        """Masses of softmax(logits * logit_scale) straight from the logits rows ([B, V] float32 / bfloat16 / float16 on
        the device) and the rows' lse (float32 [B]: the fused step's `lse` output; None: computed here, `HipEngine.row_lse`):
        what the reference gets from `batch_weight_sum(logprobs.exp())` (trie/parallel.py:92-103) without the [B, V]
        matrix of log-probabilities ever being written.  nodes: int32 device tensor - only these nodes' masses,
        [B, len(nodes)] (only the subtrees below them are read and reduced: `selection_plan`) - or int32 [B, K]: every row's
        OWN nodes, e.g. each particle's current node's children (negative entries: none), [B, K] (wide_selections=True: most
        rows' nodes lie above the parts - the root's children, after a token boundary -, so nearly every row needs nearly
        every part: the sweep plan's two parts then beat the gathered plan's nine, 168 -> 100 us at 1024 x 50257; a row
        that needs one part of nine is better off gathered: 58 against 99 us); layout "slots": node-major [n_slots, pitch] over the folded trie (`compact()`: the value of node
        n for row r is at [slot_of[n], r] - nothing is transposed back); layout "nodes": the same over all nodes,
        [n_nodes, pitch] (see HipEngine.trie_masses); layout "slot_rows": row-major [B, n_slots] over the plan's slots
        (`slot_plan()["slot_of"]`: node -> slot) - the cheapest form: the logits are read once and nothing else is written."""
        if logits.shape[1] < len(self.decode):
            raise ValueError(f"logits rows have {logits.shape[1]} columns, vocabulary has {len(self.decode)}")
        if lse is None:  # (one more reading of the rows: hand the fused step's lse over when there is one)
            lse = self.engine.row_lse(logits, vocab=len(self.decode), logit_scale=logit_scale)
        if nodes is not None and nodes.dtype != torch.int32:
            raise TypeError("nodes must be int32")
        pl = None
        if self.resident and layout in ("rows", "slot_rows"):
            if nodes is None:
                pl = self._whole_plan_device(logits.shape[0], slots=layout == "slot_rows")
            elif nodes.dim() == 2 and wide_selections and self.sweep and (self.plan(sweep=True) or {}).get("n_parts", 99) <= 62:
                pl = self.plan_device_arrays(sweep=True)
            else:
                pl = self.plan_device_arrays()
        if pl is not None and nodes is not None and nodes.dim() == 1 and layout == "rows" and self.prune_selection:
            # only the subtrees below the selected nodes are read and reduced (a plan of that sub-forest, cached per selection)
            # (SWEEP_MIN_ROWS rows and more: the sub-forest's sweep plan - persistent workgroups read every row front to back
            # once, where the gathered plan's parts each touch most of a row's sectors for their tokens)
            want_sweep = self.sweep and logits.shape[0] >= self.SWEEP_MIN_ROWS
            pl = ((self.selection_plan(nodes, build=self.prune_selection is True, sweep=True) if want_sweep else None)
                  or self.selection_plan(nodes, build=self.prune_selection is True) or pl)
        if pl is not None:
            got = self._trie_rows(logits, pl, op, True, lse=lse, logit_scale=logit_scale, nodes=nodes,
                                  layout="slots" if layout == "slot_rows" and nodes is None else "rows")
            if got is not None:
                return got
            if nodes is not None and nodes.dim() == 2:
                raise RuntimeError("per-row selections need glb_trie_rows, which this device refused")
        if layout == "slot_rows":
            raise ValueError("layout 'slot_rows' needs a plan (TokenByteTrie.plan() returned None, or resident is off)")
        if layout == "nodes" or self.compact()["n_levels"] == 0:
            return self.engine.trie_masses(logits, self.device_arrays(), op, True, lse=lse, logit_scale=logit_scale,
                                           nodes=nodes, layout=layout)
        c = self.compact_device_arrays()
        if layout == "slots" and nodes is None:
            return self.engine.trie_masses(logits, c, op, True, lse=lse, logit_scale=logit_scale, layout="nodes")
        if layout != "rows":
            raise ValueError(f"unknown layout {layout!r}")
        if nodes is None:
            sel = c["slot_of"]
        else:
            if nodes.dtype != torch.int32:
                raise TypeError("nodes must be int32")
            sel = c["slot_of"][nodes.long()]
        return self.engine.trie_masses(logits, c, op, True, lse=lse, logit_scale=logit_scale, nodes=sel)

    def batch_weight_sum(self, ws):
        """base.py:196-205 / parallel.py:92-103: summed weights of every node for a batch of weight rows."""
        return self.batch_weight_sum_device(ws).cpu().numpy()

    def batch_weight_max(self, ws):
        """base.py:207-216 / parallel.py:120-145"""
        return self.batch_weight_max_device(ws).cpu().numpy()

    def weight_sum(self, ws):
        """base.py:147-169"""
        return self.batch_weight_sum(self._rows(ws))[0]

    def weight_max(self, ws):
        """base.py:171-193"""
        return self.batch_weight_max(self._rows(ws))[0]


class AsyncTokenByteTrie:
    """Awaitable `weight_sum` / `weight_max` with automatic batching: the counterpart of the reference's
    `AsyncTokenCharacterTrie` (trie/async_impl.py:10-160).  Coroutines hand over one weight row each; whatever has been
    handed over by the time the event loop gets back to the drain task goes to the device as ONE batch per operation
    (glb_trie_masses on the folded trie), and every caller gets its row of the result - a float32 device tensor
    [n_nodes] (the reference returns NumPy rows; `.cpu().numpy()` gives those).  A failing batch fails every request
    in it (async_impl.py:129-134)."""

    def __init__(self, trie):
        self.trie = trie
        self._pending = []   # (weights row, future, op)
        self._task = None
        self._wake = None

    @classmethod
    def from_vocab(cls, vocab, engine=None, **kwargs):
        """async_impl.py:23-45 (one implementation here: the device one)."""
        return cls(TokenByteTrie(vocab, engine=engine, **kwargs))

    def start(self):
        """Start the drain task on the running loop (done by the first request)."""
        import asyncio

        if self._task is None or self._task.done():
            self._wake = asyncio.Event()
            self._task = asyncio.get_running_loop().create_task(self._drain())

    async def _request(self, ws, op):
        import asyncio

        self.start()
        fut = asyncio.get_running_loop().create_future()
        self._pending.append((ws, fut, op))
        self._wake.set()
        return await fut

    async def weight_sum(self, ws):
        """async_impl.py:55-67"""
        return await self._request(ws, 0)

    async def weight_max(self, ws):
        """async_impl.py:69-81"""
        return await self._request(ws, 1)

    def _fail_waiters(self, batch, exc):
        """nobody may be left waiting: the requests of `batch` and everything still queued get `exc`"""
        stranded, self._pending = list(batch) + self._pending, []
        for _, f, _ in stranded:
            if not f.done():
                f.set_exception(exc)

    async def _drain(self):
        import asyncio

        batch = []
        try:
            await self._drain_loop(batch)
        except asyncio.CancelledError:  # cleanup() / shutdown(): the requests in flight and in the queue end with it
            self._fail_waiters(batch, asyncio.CancelledError("AsyncTokenByteTrie was shut down"))
            raise

    async def _drain_loop(self, batch):
        while True:
            await self._wake.wait()
            self._wake.clear()
            batch[:], self._pending = self._pending, []
            try:
                for op in (0, 1):
                    group = [(w, f) for w, f, o in batch if o == op]
                    if not group:
                        continue
                    rows = torch.stack([torch.as_tensor(w, dtype=torch.float32).to(self.trie.engine.device) for w, _ in group])
                    out = self.trie._batch(rows, op, False)
                    for (_, f), r in zip(group, out):
                        if not f.done():
                            f.set_result(r)
            except Exception as e:  # noqa: BLE001 - whatever went wrong, nobody may be left waiting
                for _, f, _ in batch:
                    if not f.done():
                        f.set_exception(e)
            batch.clear()

    async def cleanup(self):
        """async_impl.py:136-145"""
        import asyncio

        if self._task is not None and not self._task.done():
            self._task.cancel()
            try:
                await self._task
            except asyncio.CancelledError:
                pass
        self._task = None

    def shutdown(self):
        """async_impl.py:147-156"""
        if self._task is not None:
            try:
                self._task.cancel()
            except RuntimeError:  # the loop is gone
                pass
            self._task = None
        import asyncio

        try:  # (a cancelled task that never runs again cannot tell its waiters: do it here)
            self._fail_waiters([], asyncio.CancelledError("AsyncTokenByteTrie was shut down"))
        except RuntimeError:  # the waiters' loop is gone, and they with it
            self._pending = []

    def __del__(self):
        self.shutdown()
