"""Single-configuration launch loop for rocprofv3 counter passes: kprof.py MASK RNG VARIANT [iters]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import run
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
mask, rng, var = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
eng = HipEngine("cuda:0")
run(eng, 1024, 50257, torch.float32, mask, rng, 4, iters, variant=var)
