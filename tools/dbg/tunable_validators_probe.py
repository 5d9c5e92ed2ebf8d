import sys; sys.path.insert(0, "/root/repo")
import torch, genlm_backend_amd
from genlm_backend_amd import gemm_tuning
import tempfile, os
src = open(os.path.join(os.path.dirname(gemm_tuning.__file__), "tuned", "gfx950.csv")).read().replace("ROCBLAS_VERSION,5.0.2", "ROCBLAS_VERSION,9.9.9")
f = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False); f.write(src); f.close()
print("mismatched validators ->", gemm_tuning.use_recorded(f.name))
a = torch.randn(64, 64, device="cuda"); print((a @ a).sum().item() != 0)
print("good file ->", gemm_tuning.use_recorded())
