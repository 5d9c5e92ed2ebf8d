"""cProfile of the asyncio API path (bench.py --workload api) on the GPU: where the host time of a step goes."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
import bench
eng = HipEngine("cuda:0")
w = bench.ApiWorkload(eng, torch.device("cuda:0"), 0, 1, None, logprobs=len(sys.argv) > 1 and sys.argv[1] == "logprobs",
                      coro=len(sys.argv) > 1 and sys.argv[1] == "coro", auto_kv="autokv" in sys.argv[1:])
for i in range(4):
    w.step(i, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    w.step(i, True)
torch.cuda.synchronize()
pr.disable()
dt = time.perf_counter() - t0
print(f"{dt / 10 * 1e3:.1f} ms per step")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
