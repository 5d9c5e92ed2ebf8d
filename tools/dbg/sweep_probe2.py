"""glb_trie_rows, sweep plan against gathered plan by batch size and for per-row selections (times only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from genlm_backend_amd.tokenization import Token
from genlm_backend_amd.trie import TokenByteTrie

eng = HipEngine("cuda:0"); dev = eng.device
V = 50257
rs = np.random.default_rng(0)
words, seen = [], set()
while len(words) < V:
    w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
    if w not in seen: seen.add(w); words.append(w)
trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=eng)
g = trie.plan_device_arrays(); s = trie.plan_device_arrays(sweep=True)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))

x = torch.randn((1024, V), device=dev) * 3
_, lse, _ = eng.step(x, rng_mode=0)
for B in (8, 32, 64, 128, 256, 512, 1024):
    xb = x[:B].contiguous(); lb = lse[:B].contiguous()
    for layout in ("rows", "slots"):
        tg = timed(lambda: eng.trie_rows(xb, g, 0, True, lse=lb, layout=layout))
        ts = timed(lambda: eng.trie_rows(xb, s, 0, True, lse=lb, layout=layout))
        print(f"B={B} {layout}: gather {tg:8.1f} us   sweep {ts:8.1f} us", flush=True)
B = 1024
d1 = sorted(trie.children[trie.root].values())
for name, cur in (("rowsel", [d1[int(k)] for k in rs.integers(0, len(d1), B)]), ("rowsel-root", [trie.root] * B)):
    K = max(len(trie.jump[c]) for c in set(cur))
    rowsel = np.full((B, K), -1, np.int32)
    for r, c in enumerate(cur):
        rowsel[r, :len(trie.jump[c])] = trie.jump[c]
    rsd = torch.from_numpy(rowsel).to(dev)
    a = eng.trie_rows(x, g, 0, True, lse=lse, nodes=rsd); b = eng.trie_rows(x, s, 0, True, lse=lse, nodes=rsd)
    assert torch.equal(a, b)
    tg = timed(lambda: eng.trie_rows(x, g, 0, True, lse=lse, nodes=rsd))
    ts = timed(lambda: eng.trie_rows(x, s, 0, True, lse=lse, nodes=rsd))
    print(f"{name}: gather {tg:8.1f} us   sweep {ts:8.1f} us", flush=True)
sel = torch.from_numpy(rs.choice(len(trie), 4096, replace=False).astype(np.int32)).to(dev)
tp = timed(lambda: trie.masses_from_logits(x, lse, nodes=sel))
ts = timed(lambda: eng.trie_rows(x, s, 0, True, lse=lse, nodes=sel))
print(f"4096 selected: pruned gather {tp:8.1f} us   sweep (whole trie) {ts:8.1f} us")
