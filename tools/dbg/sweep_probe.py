"""glb_trie_rows: the sweep plan (a row read front to back, values only in LDS) against the gathered plan - same bits, times."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from genlm_backend_amd.tokenization import Token
from genlm_backend_amd.trie import TokenByteTrie

eng = HipEngine("cuda:0"); dev = eng.device
V = int(os.environ.get("V", 50257))
caps = [int(c) for c in os.environ.get("CAPS", "0").split(",")]
rs = np.random.default_rng(0)
words, seen = [], set()
while len(words) < V:
    w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
    if w not in seen: seen.add(w); words.append(w)
trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=eng)
g = trie.plan_device_arrays()
print(f"gather plan: {g['n_parts']} parts of <= {g['max_local']}, lds {g['lds_bytes']}", flush=True)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))

B = 1024
x = torch.randn((B, V), device=dev) * 3
_, lse, _ = eng.step(x, rng_mode=0)
for cap in caps:
    s = trie.plan_device_arrays(cap or None, sweep=True)
    print(f"sweep plan cap {s['cap']}: {s['n_parts']} parts of <= {s['max_local']}, lds {s['lds_bytes']}, top {s['n_top']} nodes / lds {s['lds_top_bytes']}", flush=True)
    for Bb in (1, 7, 64, 1024):
        for dt in (torch.float32, torch.bfloat16):
            xb = x[:Bb].to(dt).contiguous(); lb = lse[:Bb].contiguous()
            for op in (0, 1):
                a = eng.trie_rows(xb, g, op, True, lse=lb, layout="rows")
                b = eng.trie_rows(xb, s, op, True, lse=lb, layout="rows")
                torch.cuda.synchronize()
                assert torch.equal(a, b), (Bb, dt, op, (a != b).sum().item())
            a = eng.trie_rows(xb, g, 0, True, lse=lb, layout="slots")
            b = eng.trie_rows(xb, s, 0, True, lse=lb, layout="slots")
            assert torch.equal(a[:, g["slot_of"].long()], b[:, s["slot_of"].long()])
            if Bb in (1, 1024):
                for layout in ("rows", "slots"):
                    tg = timed(lambda: eng.trie_rows(xb, g, 0, True, lse=lb, layout=layout))
                    ts = timed(lambda: eng.trie_rows(xb, s, 0, True, lse=lb, layout=layout))
                    print(f"B={Bb} {str(dt)[6:]} {layout}: gather {tg:8.1f} us   sweep {ts:8.1f} us", flush=True)
    # weights (not log-probs), odd pitch
    w = torch.rand((33, V + 3), device=dev)[:, :V]
    a = eng.trie_rows(w, g, 0, False, layout="rows"); b = eng.trie_rows(w, s, 0, False, layout="rows")
    assert torch.equal(a, b)
print("ok")
