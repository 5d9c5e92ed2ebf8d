"""ctypes front-end of the CPU oracle (oracle/glb_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (genlm-backend_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libglb_oracle.so")

F32, BF16, F16 = 0, 1, 2
MASK_NONE, MASK_BITS, MASK_F32 = 0, 1, 2
RNG_NONE, RNG_PHILOX, RNG_NOISE = 0, 1, 2


def build(force=False):
    src = os.path.join(_HERE, "glb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, ct=C.c_void_p):
    if a is None:
        return None
    return a.ctypes.data_as(ct)


def _dtype_code(a):
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.float16:
        return F16
    if a.dtype == np.uint16:  # raw bf16 bits
        return BF16
    raise TypeError(a.dtype)


def f32_to_bf16_bits(x):
    """round-to-nearest-even float32 -> bf16 bit pattern (uint16)"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def step(logits, row_of=None, mask_kind=MASK_NONE, mask=None, mask_id=None, rng_mode=RNG_NONE,
         noise=None, seed=0, offset=0, particle_base=0, logit_scale=1.0, n_particles=None, want_margin=False,
         contract="poly", want_edge=False):
    """Layer-B particle step.  Returns (logZ, lse, token) [+ the race's tie margin with want_margin] [+ the Philox
    draws' distance from the nearest boundary of the inverse CDF with want_edge].  contract "hw": the terms of
    GLB_STEP_HW_EXP restated with exp2f (not bit for bit what v_exp_f32 gives: glb_oracle.c)."""
    logits = np.ascontiguousarray(logits)
    n_rows, ld = logits.shape
    V = ld
    if row_of is not None:
        row_of = np.ascontiguousarray(row_of, dtype=np.int32)
        n = len(row_of)
    else:
        n = n_rows if n_particles is None else n_particles
    n_masks, mask_ld = 0, 0
    if mask_kind != MASK_NONE:
        mask = np.ascontiguousarray(mask)
        n_masks, mask_ld = mask.shape
    if mask_id is not None:
        mask_id = np.ascontiguousarray(mask_id, dtype=np.int32)
    noise_ld = 0
    if noise is not None:
        noise = np.ascontiguousarray(noise, dtype=np.float32)
        noise_ld = noise.shape[1]
    logZ = np.empty(n, np.float32)
    lse = np.empty(n, np.float32)
    tok = np.full(n, -2, np.int32)
    margin = np.ones(n, np.float32) if want_margin else None
    if contract not in ("poly", "hw"):
        raise ValueError(contract)
    edge = np.ones(n, np.float32) if want_edge else None
    rc = lib().orc_step2(
        _p(logits), _dtype_code(logits), C.c_int64(n_rows), C.c_int64(V), C.c_int64(ld),
        C.c_float(logit_scale), C.c_int64(n), _p(row_of), C.c_int(mask_kind), _p(mask),
        C.c_int64(mask_ld), C.c_int64(n_masks), _p(mask_id), C.c_int(rng_mode), _p(noise),
        C.c_int64(noise_ld), C.c_uint64(seed), C.c_uint64(offset), C.c_int64(particle_base),
        _p(logZ), _p(lse), _p(tok), _p(margin), C.c_int(1 if contract == "hw" else 0), _p(edge))
    if rc:
        raise RuntimeError(f"orc_step rc={rc}")
    out = (logZ, lse, tok)
    if want_margin:
        out += (margin,)
    if want_edge:
        out += (edge,)
    return out


def log_softmax_rows(logits, logit_scale=1.0):
    logits = np.ascontiguousarray(logits)
    n, V = logits.shape
    out = np.empty((n, V), np.float32)
    lse = np.empty(n, np.float32)
    lib().orc_log_softmax_rows(_p(logits), _dtype_code(logits), C.c_int64(n), C.c_int64(V),
                               C.c_int64(V), C.c_float(logit_scale), _p(out), C.c_int64(V), _p(lse))
    return out, lse


def round_rows_16(rows32, dtype):
    """Layer B of glb_log_softmax_rows' out_dtype: the float32 log-probabilities rounded to nearest even into the
    logits' own 16-bit type - what the reference's `torch.log_softmax` on a 16-bit tensor returns up to the last bit
    of its own float32 intermediate (cache.py:96 keeps the dtype).  dtype "bf16" -> uint16 words, "f16" -> float16."""
    rows32 = np.ascontiguousarray(rows32, dtype=np.float32)
    if dtype == "f16":
        return rows32.astype(np.float16)  # IEEE round to nearest even
    u = rows32.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(rows32)
    if nan.any():  # (the integer recipe can turn a NaN into an infinity: keep it a NaN, as the hardware conversion does)
        r[nan] = ((u[nan] >> 16) | 0x0040).astype(np.uint16)
    return r


def mask_f32_to_bits(mask):
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    k, V = mask.shape
    W = (V + 31) // 32
    bits = np.zeros((k, W), np.uint32)
    nb = C.c_int32(0)
    lib().orc_mask_f32_to_bits(_p(mask), C.c_int64(k), C.c_int64(V), C.c_int64(V), _p(bits),
                               C.c_int64(W), C.byref(nb))
    return bits, bool(nb.value)


def normalize_weights(lw):
    lw = np.ascontiguousarray(lw, dtype=np.float32)
    probs = np.empty_like(lw)
    stats = np.empty(2, np.float32)
    lib().orc_normalize_weights(_p(lw), C.c_int64(len(lw)), _p(probs), _p(stats))
    return probs, stats


class MT19937(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int32)]


def mt_exponential(seed, n, state=None):
    """float32 Exp(1) variates exactly as torch CPU `exponential_` draws them."""
    st = state or MT19937()
    if state is None:
        lib().orc_mt19937_seed(C.byref(st), C.c_uint64(seed))
    out = np.empty(n, np.float32)
    lib().orc_mt19937_exponential_f32(C.byref(st), _p(out), C.c_int64(n))
    return out, st


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def ref_particle(logps, mask, E):
    """Layer-A README.md:84-87 particle math in float32.  Returns (logZ, token)."""
    logps = np.ascontiguousarray(logps, dtype=np.float32)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.float32)
    E = None if E is None else np.ascontiguousarray(E, dtype=np.float32)
    z = C.c_float()
    t = C.c_int32(-2)
    lib().orc_ref_particle(_p(logps), _p(mask), C.c_int64(len(logps)), _p(E), C.byref(z), C.byref(t))
    return z.value, t.value


def ref_log_softmax(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_ref_log_softmax(_p(x), C.c_int64(len(x)), _p(out))
    return out


def ragged(contexts):
    """list of token lists -> (tokens int32, starts int64, lengths int32)"""
    lens = np.array([len(c) for c in contexts], np.int32)
    starts = np.zeros(len(contexts), np.int64)
    if len(contexts) > 1:
        starts[1:] = np.cumsum(lens[:-1])
    tok = np.zeros(max(int(lens.sum()), 1), np.int32)
    for i, c in enumerate(contexts):
        tok[starts[i]:starts[i] + lens[i]] = c
    return tok, starts, lens


def group_contexts(contexts):
    tok, st, ln = ragged(contexts)
    n = len(contexts)
    g = np.empty(n, np.int32)
    rep = np.empty(max(n, 1), np.int32)
    ng = C.c_int32()
    lib().orc_group_contexts(_p(tok), _p(st), _p(ln), C.c_int64(n), _p(g), _p(rep), C.byref(ng))
    return g, rep[:ng.value].copy(), ng.value


def match_prefixes(contexts, prefixes):
    tok, st, ln = ragged(contexts)
    ptok, pst, pln = ragged(prefixes)
    n = len(contexts)
    pref = np.empty(n, np.int32)
    base = np.empty(n, np.int32)
    lib().orc_match_prefixes(_p(tok), _p(st), _p(ln), C.c_int64(n), _p(ptok), _p(pst), _p(pln),
                             C.c_int64(len(prefixes)), _p(pref), _p(base))
    return pref, base


def gather_padded(contexts, sel, base, pad_id, p_max, l_max):
    tok, st, ln = ragged(contexts)
    sel_a = None if sel is None else np.ascontiguousarray(sel, dtype=np.int32)
    base_a = None if base is None else np.ascontiguousarray(base, dtype=np.int32)
    n_sel = len(contexts) if sel is None else len(sel)
    ids = np.empty((n_sel, l_max), np.int64)
    am = np.empty((n_sel, p_max + l_max), np.int64)
    pos = np.empty((n_sel, l_max), np.int64)
    last = np.empty(n_sel, np.int32)
    rc = lib().orc_gather_padded(_p(tok), _p(st), _p(ln), _p(sel_a), C.c_int64(n_sel), _p(base_a),
                                 C.c_int64(pad_id), C.c_int64(p_max), C.c_int64(l_max), _p(ids),
                                 _p(am), _p(pos), _p(last))
    if rc:
        raise RuntimeError("orc_gather_padded")
    return ids, am, pos, last


def gather_kv_padded(slabs, prefix_of, p_max):
    """slabs: list of [H, P_k, D] arrays; returns [U, H, p_max, D]."""
    slabs = [np.ascontiguousarray(s) for s in slabs]
    H, _, D = slabs[0].shape
    eb = slabs[0].dtype.itemsize
    ptrs = (C.c_void_p * len(slabs))(*[s.ctypes.data for s in slabs])
    lens = np.array([s.shape[1] for s in slabs], np.int32)
    prefix_of = np.ascontiguousarray(prefix_of, dtype=np.int32)
    U = len(prefix_of)
    out = np.empty((U, H, p_max, D), slabs[0].dtype)
    lib().orc_gather_kv_padded(ptrs, _p(lens), C.c_int64(len(slabs)), _p(prefix_of), C.c_int64(U),
                               C.c_int64(H), C.c_int64(D), C.c_int64(p_max), C.c_int32(eb), _p(out))
    return out


def particles_advance(ctx, lengths, active, lw, logZ, tok, eos, max_len):
    n, ld = ctx.shape
    lib().orc_particles_advance(_p(ctx), C.c_int64(ld), _p(lengths), _p(active), _p(lw),
                                _p(np.ascontiguousarray(logZ, dtype=np.float32)),
                                _p(np.ascontiguousarray(tok, dtype=np.int32)), C.c_int64(n),
                                C.c_int32(eos), C.c_int32(max_len))


def resample_systematic(log_weights, seed=0, offset=0):
    """Layer-B systematic resampling: (ancestors int32 [n], logsumexp of the weights)."""
    lw = np.ascontiguousarray(log_weights, dtype=np.float32)
    n = len(lw)
    anc = np.empty(n, np.int32)
    stats = np.empty(1, np.float32)
    rc = lib().orc_resample_systematic(_p(lw), C.c_int64(n), C.c_uint64(seed), C.c_uint64(offset), _p(anc), _p(stats))
    if rc:
        raise RuntimeError(f"orc_resample_systematic rc={rc}")
    return anc, float(stats[0])


def trie_reduce(ws, flat, op=0, from_logprobs=False):
    """Layer-B trie masses: ws [B, V] float32, flat = dict of the flattened trie arrays (genlm_backend_amd.trie)."""
    ws = np.ascontiguousarray(ws, dtype=np.float32)
    B, V = ws.shape
    n_nodes = int(flat["n_nodes"])
    out = np.empty((B, n_nodes), np.float32)
    a = {k: np.ascontiguousarray(flat[k], dtype=np.int32) for k in ("leaf_node", "level_start", "level_nodes", "child_ptr", "child_idx")}
    rc = lib().orc_trie_reduce(_p(ws), C.c_int64(V), C.c_int64(B), C.c_int64(V), C.c_int64(n_nodes),
                               C.c_int64(len(a["level_start"]) - 1), _p(a["leaf_node"]), _p(a["level_start"]),
                               _p(a["level_nodes"]), _p(a["child_ptr"]), _p(a["child_idx"]), C.c_int(op),
                               C.c_int(1 if from_logprobs else 0), _p(out), C.c_int64(n_nodes))
    if rc:
        raise RuntimeError(f"orc_trie_reduce rc={rc}")
    return out


# ---- KV rows shared between contexts: the block table (layer B of glb_match_rows / glb_kv_plan; integer work) -------
def ctx_hash(ctx):
    """The library's context hash (glb_hash_contexts): an FNV-style fold, one token at a time."""
    M = (1 << 64) - 1
    h = 0xcbf29ce484222325
    for t in ctx:
        h ^= int(t) & 0xffffffff
        h = (h * 0x100000001b3) & M
        h ^= h >> 29
    return h


def match_rows(contexts, rep, n_groups, row_tok, row_len, row_hash):
    """For every dedup group the smallest table row that holds exactly its context, else the smallest row that holds
    its first L - 1 tokens, else -1; plus the groups' hashes.  (The reference's counterpart is the trie walk that finds
    the deepest node with KV, hf.py:314-344 / cache.py:103-191.)"""
    R, cap = row_tok.shape
    old = np.full(n_groups, -1, np.int32)
    gh = np.zeros(n_groups, np.uint64)
    for u in range(n_groups):
        c = [int(t) for t in contexts[rep[u]]]
        L = len(c)
        gh[u] = ctx_hash(c)
        if L == 0 or L > cap:
            continue
        exact = par = -1
        for r in range(R):
            rl = int(row_len[r])
            if rl <= 0:
                continue
            if rl == L and int(row_hash[r]) == int(gh[u]) and list(row_tok[r, :rl]) == c:
                exact = r if exact < 0 else exact
            elif rl == L - 1 and int(row_hash[r]) == ctx_hash(c[:-1]) and list(row_tok[r, :rl]) == c[:-1]:
                par = r if par < 0 else par
        old[u] = exact if exact >= 0 else par
    return old, gh


def kv_plan(group_of, rep, n_groups, old, lengths, n_rows, cap, stamps=None, call_no=0):
    """The block table of one step, as include/glb.h states it for glb_kv_plan (`old`: per GROUP).  Returns a dict of
    int32 arrays with the library's output names; `stamps` (int64, optional) is updated in place."""
    n, U, R = len(group_of), int(n_groups), int(n_rows)
    L = np.array([int(lengths[rep[u]]) for u in range(U)], np.int64)
    # a row index outside the table reads as "no row"; so does the row of a context that has outgrown a row's cap positions
    old = np.array([int(old[u]) if 0 <= int(old[u]) < R and L[u] <= cap else -1 for u in range(U)], np.int64)
    keeper = {}
    for u in range(U):
        if old[u] >= 0:
            keeper.setdefault(int(old[u]), u)
    keep = [u for u in range(U) if old[u] >= 0 and keeper[int(old[u])] == u]
    cand = [u for u in range(U) if old[u] >= 0 and keeper[int(old[u])] != u]
    fresh = [u for u in range(U) if old[u] < 0 and L[u] <= cap]
    live = {int(old[u]) for u in keep}
    free = [r for r in range(R) if r not in live]
    if stamps is not None:
        free.sort(key=lambda r: (int(stamps[r]), r))
    grp_row = np.full(n, -1, np.int32)
    for u in keep:
        grp_row[u] = old[u]
    for u, r in zip(cand + fresh, free):
        grp_row[u] = r
    in_a = [u for u in range(U) if old[u] >= 0 and grp_row[u] >= 0]
    in_b = [u for u in range(U) if not (old[u] >= 0 and grp_row[u] >= 0)]
    out = {k: np.zeros(n, np.int32) for k in ("logits_row", "rows_a", "ctx_a", "pos_a", "ctx_b", "rows_b", "row_of_context")}
    out["group_row"] = grp_row
    out["copy_src"], out["copy_len"] = np.full(R, -1, np.int32), np.zeros(R, np.int32)
    out["ctx_of_row"], out["pos_of_row"] = np.full(R, -1, np.int32), np.zeros(R, np.int32)
    copied = 0
    for k, u in enumerate(in_a):
        r = int(grp_row[u])
        out["logits_row"][u] = k
        out["rows_a"][k], out["ctx_a"][k], out["pos_a"][k] = r, rep[u], L[u] - 1
        out["ctx_of_row"][r], out["pos_of_row"][r] = rep[u], L[u] - 1
        if r != old[u]:
            out["copy_src"][r], out["copy_len"][r] = old[u], L[u] - 1
            copied += 1
    for k, u in enumerate(in_b):
        out["logits_row"][u] = len(in_a) + k
        out["ctx_b"][k], out["rows_b"][k] = rep[u], grp_row[u]
        if grp_row[u] >= 0:
            out["ctx_of_row"][grp_row[u]] = -2
    if stamps is not None:
        for u in range(U):
            if grp_row[u] >= 0:
                stamps[grp_row[u]] = call_no
    out["row_of_context"] = grp_row[np.asarray(group_of, np.int64)].astype(np.int32)
    out["head"] = np.array([U, len(in_a), len(in_b), copied, sum(1 for u in in_b if grp_row[u] < 0),
                            max([int(L[u]) for u in in_b], default=0), len(free), 0], np.int32)
    out["n_valid"] = dict(group_row=U, logits_row=U, rows_a=len(in_a), ctx_a=len(in_a), pos_a=len(in_a), ctx_b=len(in_b),
                          rows_b=len(in_b), row_of_context=n)
    return out
