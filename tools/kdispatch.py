"""Dispatch tuning: persistent kernel (variant 21/22/23) vs one-workgroup-per-particle (variant -1) over shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import run, HipEngine
eng = HipEngine("cuda:0")
f32, bf16 = torch.float32, torch.bfloat16
cases = [(320, 50257, f32, 21), (384, 50257, f32, 21),
         (1024, 32000, bf16, 24), (4096, 32000, bf16, 24), (4096, 32000, f32, 22), (1024, 151936, bf16, 25), (512, 151936, bf16, 25),
         (1024, 65536, f32, 25)]
for B, V, dt, var in cases:
    for v in (-1, var):
        if v == 0: continue
        try:
            run(eng, B, V, dt, 1, 1, 4 if B * V * (4 if dt == f32 else 2) < 3e8 else 2, 30, variant=v)
        except Exception as e:
            print("B", B, "V", V, dt, "variant", v, "failed:", str(e)[:80])
