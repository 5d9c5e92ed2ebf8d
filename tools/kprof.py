"""Single-configuration launch loop for rocprofv3 passes: kprof.py SHAPE MASK RNG [iters]   (SHAPE: gpt2 | llama | B,V,dtype)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import run
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
shape, mask, rng = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
eng = HipEngine("cuda:0")
if shape == "gpt2":
    B, V, dt = 1024, 50257, torch.float32
elif shape == "llama":
    B, V, dt = 512, 128256, torch.bfloat16
else:
    b, v, d = shape.split(",")
    B, V, dt = int(b), int(v), {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[d]
run(eng, B, V, dt, mask, rng, 4 if B * V < 80_000_000 else 2, iters)
