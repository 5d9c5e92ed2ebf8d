"""Can the one-token forward on the KV slab (in place: constant shapes) be captured into a hipGraph?  Times eager vs
replay for the GPT-2-small and Llama-3.2-1B shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import genlm_backend_amd
from genlm_backend_amd.engine import HipEngine
from genlm_backend_amd.llm import AsyncAmdLM
from genlm_backend_amd.kv import SharedSlabKV

dev = torch.device("cuda:0")
eng = HipEngine(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
if which == "gpt2":
    from transformers import GPT2Config
    cfg, dtype, R = GPT2Config(), torch.float32, 1024
else:
    from transformers import LlamaConfig
    cfg = LlamaConfig(vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                      num_attention_heads=32, num_key_value_heads=8, head_dim=64, max_position_embeddings=4096,
                      rope_theta=500000.0, rms_norm_eps=1e-5, tie_word_embeddings=True, bos_token_id=128000, eos_token_id=128001)
    dtype, R = torch.bfloat16, 512
llm = AsyncAmdLM.from_config(cfg, None, device=dev, dtype=dtype, seed=1, engine=eng, batch_size=R)
cap = 24
# prefill 8 tokens into the rows
ids0 = torch.randint(1, 1000, (R, 8), device=dev)
out = llm._body(input_ids=ids0, use_cache=True)
srcs = [(ly.keys.contiguous(), ly.values.contiguous()) for ly in out.past_key_values.layers]
pkv = SharedSlabKV(eng, R, cap, len(srcs))
pkv.fill_rows(srcs, torch.arange(R, dtype=torch.int32, device=dev), torch.full((R,), 8, dtype=torch.int32, device=dev))
s_ids = torch.randint(1, 1000, (R, 1), device=dev)
s_pos = torch.full((R,), 8, dtype=torch.int32, device=dev)

def fwd():
    pkv.set_forward_in_place(s_pos)
    o = llm._body(input_ids=s_ids, position_ids=s_pos.view(-1, 1).long(), attention_mask=pkv.attention_mask(s_pos),
                  past_key_values=pkv, use_cache=True)
    return o.last_hidden_state[:, 0]

for _ in range(3): ref = fwd().clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): fwd()
torch.cuda.synchronize()
print(f"{which}: eager {1e3 * (time.perf_counter() - t0) / 20:.2f} ms per one-token forward over {R} rows", flush=True)
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): fwd()
torch.cuda.current_stream().wait_stream(side)
try:
    with torch.cuda.graph(g):
        static_h = fwd()
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:400])
    sys.exit(0)
g.replay(); torch.cuda.synchronize()
print("replay == eager:", torch.equal(static_h, ref), float((static_h.float() - ref.float()).abs().max()))
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print(f"{which}: graph replay {1e3 * (time.perf_counter() - t0) / 20:.2f} ms", flush=True)
# a different position / token through the static buffers
s_pos.fill_(9); s_ids.copy_(torch.randint(1, 1000, (R, 1), device=dev))
g.replay(); a = static_h.clone(); b = fwd(); torch.cuda.synchronize()
print("other position: replay == eager:", torch.equal(a, b), float((a.float() - b.float()).abs().max()))
