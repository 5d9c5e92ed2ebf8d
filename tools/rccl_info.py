"""What RCCL itself says about the collectives of the path (run with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL): a one-rank
"nccl" group on this GPU, the per-step all-gather of log-weights (1024 + 1 floats: 4 KiB), an all_to_all_single with uneven
splits (the resampling exchange) and the float64 all_reduce(MAX) of bench.py's clock.  One GPU is all a gpurun box has; the
N > 1 topology (xGMI rings) can only be printed by the driver's 8-GPU run."""
import socket

import torch
import torch.distributed as dist

s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
lw = torch.randn(1025, device=dev)
out = torch.empty(1025, device=dev)
for _ in range(3):
    dist.all_gather_into_tensor(out, lw)
rows = torch.randint(0, 1000, (37, 22), dtype=torch.int32, device=dev)
recv = torch.empty_like(rows)
dist.all_to_all_single(recv, rows, output_split_sizes=[37], input_split_sizes=[37])
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(out, lw) and torch.equal(recv, rows)
print("collectives ok; backend", dist.get_backend(), "world", dist.get_world_size())
dist.destroy_process_group()
