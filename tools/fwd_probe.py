"""Where the SIS step's forward time goes and what the cheap PyTorch-side levers are (not the product: plumbing)."""
import os, sys, time
import torch
from transformers import GPT2Config, GPT2LMHeadModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(tag, model, ids, n=8):
    with torch.no_grad():
        for _ in range(3): model.transformer(ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): model.transformer(ids)
        torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)
ids = torch.randint(0, 50257, (919, 13), device=dev)
m = GPT2LMHeadModel(GPT2Config()).to(dev).eval()
run("baseline (gelu_new, sdpa default)", m, ids)
for blk in m.transformer.h:
    blk.mlp.act = torch.nn.GELU(approximate="tanh")
run("fused tanh-gelu", m, ids)
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.MATH, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION):
    try:
        with sdpa_kernel(be):
            run(f"fused gelu + sdpa {be}", m, ids)
    except Exception as e:
        print(be, "failed", str(e)[:80])
m2 = GPT2LMHeadModel(GPT2Config(attn_implementation="eager")).to(dev).eval()
for blk in m2.transformer.h:
    blk.mlp.act = torch.nn.GELU(approximate="tanh")
run("fused gelu + eager attention", m2, ids)
