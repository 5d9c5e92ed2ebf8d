"""log-softmax materialisation (glb_log_softmax_rows, cache.py:93-98 replacement): read V*s + write V*4 per row."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd import _lib
if os.environ.get("GLB_DBG_LIB"):  # diagnostic build of the library (make -C genlm-backend_amd/csrc dbg)
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", os.environ["GLB_DBG_LIB"])
from genlm_backend_amd.engine import HipEngine
eng = HipEngine("cuda:0"); dev = eng.device
# (rows, vocabulary, logits dtype, output in the logits' own dtype - what the reference returns, cache.py:96 - instead of float32)
for B, V, dt, same in [(1024, 50257, torch.float32, False), (256, 50257, torch.float32, False), (1024, 50257, torch.bfloat16, False),
                       (512, 128256, torch.bfloat16, False), (64, 50257, torch.float32, False), (1024, 50257, torch.bfloat16, True),
                       (512, 128256, torch.bfloat16, True)]:
    bufs = [(torch.randn((B, V), device=dev) * 3).to(dt) for _ in range(3)]
    outs = [torch.empty((B, V), device=dev, dtype=dt if same else torch.float32) for _ in range(3)]
    for i in range(3): eng.log_softmax_rows(bufs[i], out=outs[i])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for i, (a, b) in enumerate(ev):
        a.record(); eng.log_softmax_rows(bufs[i % 3], out=outs[i % 3]); b.record()
    torch.cuda.synchronize()
    t = np.median([a.elapsed_time(b) * 1e3 for a, b in ev])
    byt = B * V * (bufs[0].element_size() + outs[0].element_size())
    ref = torch.log_softmax(bufs[0].float(), -1)
    err = (outs[0].float() - ref).abs().max().item()
    # torch for comparison
    for _ in range(3): torch.log_softmax(bufs[0].float(), -1)
    torch.cuda.synchronize()
    ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for i, (a, b) in enumerate(ev2):
        a.record(); torch.log_softmax(bufs[i % 3].float(), -1); b.record()
    torch.cuda.synchronize()
    t2 = np.median([a.elapsed_time(b) * 1e3 for a, b in ev2])
    print(f"log_softmax_rows B={B} V={V} {dt} -> {outs[0].dtype}: {t:8.1f} us  {byt / t / 1e6:6.3f} TB/s ({byt / t / 1e6 / 8 * 100:4.1f}% of 8 TB/s)  max|err| vs torch {err:.2e}   torch.log_softmax(float): {t2:8.1f} us", flush=True)
