"""Where a (row, part) workgroup of glb_trie_rows spends its time (diagnostic build: `make -C genlm-backend_amd/csrc dbg`):
every workgroup leaves [start, leaves in flight, leaves in LDS, reduced, written] (s_memrealtime, 100 MHz) in the part
of the workspace the selection's slots would take.  Usage: stamps_trie.py [rows|slot_rows] [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import genlm_backend_amd  # noqa: E402,F401
from genlm_backend_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("GLB_DBG_LIB", "libglb_hip_dbg.so"))
from genlm_backend_amd.engine import HipEngine  # noqa: E402
from genlm_backend_amd.tokenization import Token  # noqa: E402
from genlm_backend_amd.trie import TokenByteTrie  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "slot_rows"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = HipEngine("cuda:0")
dev = eng.device
rs = np.random.default_rng(0)
words, seen = [], set()
while len(words) < 50257:
    w = bytes(rs.integers(97, 123, int(rs.integers(1, 9))).astype(np.uint8))
    if w not in seen:
        seen.add(w)
        words.append(w)
trie = TokenByteTrie([Token(i, w) for i, w in enumerate(words)], engine=eng)
if os.environ.get("GLB_TRIE_CAP"):
    trie.PLAN_CAP = int(os.environ["GLB_TRIE_CAP"])
trie.sweep = os.environ.get("GLB_TRIE_SWEEP", "0") == "1"
if trie.sweep and os.environ.get("GLB_TRIE_CAP"):
    trie.sweep_cap = lambda: int(os.environ["GLB_TRIE_CAP"])
pl = trie.plan(sweep=trie.sweep)
x = torch.randn((B, len(words)), device=dev) * 3
_, lse, _ = eng.step(x, rng_mode=0)
if os.environ.get("GLB_TRIE_DT") == "bf16":
    x = x.to(torch.bfloat16)
for i in range(6):
    if i == 5:
        eng._trie_ws.zero_()  # (only the workgroups of the last call leave stamps)
    trie.masses_from_logits(x, lse, layout=layout)
torch.cuda.synchronize()
n_blocks = min(65536, (B + 7) // 8 * 8 * pl["n_parts"]) if not trie.sweep else min(8192, B) * pl["n_parts"]  # (sweep: a record per (row, part))
cut = 0 if pl["n_top"] == 0 else (B * pl["n_cut"] * 4 + 255) // 256 * 256
st = eng._trie_ws[cut: cut + n_blocks * 64].view(torch.int64).view(n_blocks, 8).cpu().numpy()
st = st[st[:, 0] > 0]
t = (st[:, :5] - st[:, 0].min()) / 100.0
seg = np.diff(t, axis=1)
print(f"{layout} B={B}: {pl['n_parts']} parts of <= {pl['max_local']} slots; {len(st)} workgroups, last one done at {t[:, 4].max():.1f} us")
for name, col in zip(["start -> leaves in flight", "-> leaves in LDS (barrier)", "-> reduced", "-> written"], seg.T):
    print(f"  {name:32s} mean {col.mean():6.2f}  p10 {np.percentile(col, 10):6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f} us")
if trie.sweep:
    d = (st[:, 5:8] - st[:, 2:3]) / 100.0
    print("  after the barrier, the three deepest internal depths done at", " / ".join(f"{x:.2f}" for x in d.mean(0)), "us")
    print("  internal nodes per depth of part 0:", np.diff(pl["idepth"][:pl["desc"][0, 3] + 1]).tolist())
life = t[:, 4] - t[:, 0]
print(f"  workgroup lifetime mean {life.mean():.2f} us; resident on average {life.sum() / t[:, 4].max():.0f} workgroups")
