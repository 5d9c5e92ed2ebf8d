"""Timing of glb_trie_reduce on a gpt2-sized synthetic vocabulary (HIP events around the engine call, median)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from genlm_backend_amd.tokenization import Token
from genlm_backend_amd.trie import TokenByteTrie
eng = HipEngine("cuda:0"); dev = eng.device
rs = np.random.default_rng(0)
words, seen = [], set()
while len(words) < 50257:
    w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
    if w not in seen: seen.add(w); words.append(w)
t0 = time.perf_counter(); trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=eng); f = trie.flat()
print(f"build {time.perf_counter() - t0:.2f} s: {len(trie)} nodes, {f['n_levels']} levels")
for B in (1, 64, 1024):
    ws = torch.rand((B, len(words)), device=dev); ws /= ws.sum(-1, keepdim=True)
    out = torch.empty((B, len(trie)), device=dev)
    for op in (0, 1):
        for _ in range(3): eng.trie_reduce(ws, trie.device_arrays(), op, out=out)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); eng.trie_reduce(ws, trie.device_arrays(), op, out=out); b.record()
        torch.cuda.synchronize()
        t = float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))
        print(f"trie_reduce op={op} B={B}: {t:9.1f} us  ({t / B:7.2f} us per row)", flush=True)
# the round-3 entry point: masses straight from logits + lse, selected / node-major output
B = 1024
x = torch.randn((B, len(words)), device=dev) * 3
_, lse, _ = eng.step(x, rng_mode=0)
sel = torch.from_numpy(rs.choice(len(trie), 4096, replace=False).astype(np.int32)).to(dev)
print(f"folded trie: {trie.compact()['n_nodes']} slots, {trie.compact()['n_levels']} levels")
pl = trie.plan()
print(f"plan: {pl['n_parts']} parts of at most {pl['max_local']} slots, top of {pl['n_top']} nodes over {pl['n_cut']} subtrees")
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))
# round 4: a row of a part of the trie resident in LDS (glb_trie_rows) against the level-synchronous kernels
for B in (1, 8, 64, 1024):
    xb = x[:B].contiguous(); lb = lse[:B].contiguous()
    for xx, tag in ((xb, "f32"), (xb.to(torch.bfloat16), "bf16")):
        for name, kw in (("rows", dict(layout="rows")), ("slot_rows", dict(layout="slot_rows")), ("4096 selected nodes", dict(nodes=sel))):
            res = []
            for resident, sweep in ((True, True), (True, False), (False, False)):
                trie.resident, trie.sweep = resident, sweep
                if name == "slot_rows" and not resident:
                    res.append(timed(lambda: trie.masses_from_logits(xx, lb, layout="slots")))  # (node-major: the old path's cheapest form)
                else:
                    res.append(timed(lambda: trie.masses_from_logits(xx, lb, **kw)))
            trie.sweep = True
            print(f"masses_from_logits {tag} B={B} -> {name}: in LDS {res[0]:9.1f} us (sweep plan where it is the default) "
                  f"{res[1]:9.1f} us (gathered plan)   level kernels {res[2]:9.1f} us", flush=True)
trie.resident = False
for B in (1, 8, 64, 1024):
    ws = torch.rand((B, len(words)), device=dev); ws /= ws.sum(-1, keepdim=True)
    for _ in range(3): trie.batch_weight_sum_device(ws)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); trie.batch_weight_sum_device(ws); b.record()
    torch.cuda.synchronize()
    t = float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))
    print(f"batch_weight_sum_device (folded trie) B={B}: {t:9.1f} us", flush=True)
B = 1024
trie.resident = False  # (the level-synchronous kernels, as in rounds 2-3)
for name, kw in (("rows", dict(layout="rows")), ("nodes (node-major, nothing transposed back)", dict(layout="nodes")),
                 ("slots (node-major over the folded trie)", dict(layout="slots")),
                 ("4096 selected nodes", dict(nodes=sel))):
    for xx, tag in ((x, "f32"), (x.to(torch.bfloat16), "bf16")):
        for _ in range(3): trie.masses_from_logits(xx, lse, **kw)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); trie.masses_from_logits(xx, lse, **kw); b.record()
        torch.cuda.synchronize()
        t = float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))
        print(f"masses_from_logits {tag} B={B} -> {name}: {t:9.1f} us", flush=True)

# config 5's vocabulary size: 128 256 synthetic tokens, 512 rows of bf16 logits
words2, seen2 = [], set()
while len(words2) < 128256:
    w = bytes(rs.integers(97, 123, int(rs.integers(1, 10))).astype(np.uint8))
    if w not in seen2: seen2.add(w); words2.append(w)
trie2 = TokenByteTrie([Token(i, w) for i, w in enumerate(words2)], engine=eng)
pl2 = trie2.plan()
sw2 = trie2.plan(sweep=True)
print(f"128256 tokens: {len(trie2)} nodes, {pl2['n_slots']} slots, plan: {pl2['n_parts']} parts of at most {pl2['max_local']} slots, top of {pl2['n_top']}; "
      f"sweep plan: {sw2['n_parts']} parts of at most {sw2['max_local']} slots, top of {sw2['n_top']}")
x2 = (torch.randn((512, len(words2)), device=dev) * 3).to(torch.bfloat16)
_, lse2, _ = eng.step(x2, rng_mode=0)
sel2 = torch.from_numpy(rs.choice(len(trie2), 4096, replace=False).astype(np.int32)).to(dev)
for name, kw in (("rows", dict(layout="rows")), ("slot_rows", dict(layout="slot_rows")), ("4096 selected nodes", dict(nodes=sel2))):
    res = []
    for resident, sweep in ((True, True), (True, False), (False, False)):
        trie2.resident, trie2.sweep = resident, sweep
        if name == "slot_rows" and not resident:
            res.append(timed(lambda: trie2.masses_from_logits(x2, lse2, layout="slots"), 5))
        else:
            res.append(timed(lambda: trie2.masses_from_logits(x2, lse2, **kw), 5))
    trie2.sweep = True
    print(f"masses_from_logits bf16 B=512 V=128256 -> {name}: in LDS {res[0]:9.1f} us (sweep plan where it is the default) "
          f"{res[1]:9.1f} us (gathered plan)   level kernels {res[2]:9.1f} us", flush=True)
