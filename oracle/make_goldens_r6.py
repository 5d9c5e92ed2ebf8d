"""Round-6 golden vectors.  TEST INFRASTRUCTURE ONLY.  Run here (CPU container):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo python -B /root/repo/oracle/make_goldens_r6.py

Writes tests/golden/ref_round6.npz:
  parity512_llama::*  torch-CPU results of the reference's per-particle op sequence (cache.py:96, README.md:84-87,
             base.py:136-141) at BASELINE config 5's FULL size, 512 x 128256 bf16 logits (tests/synth.py logits rounded to
             bf16, two masks): sampled ids under torch.manual_seed-style generator seeding, logZ, lse and the exponential
             race's margins - the recipe of parity1024::* (make_goldens_r2.py) on the 16-bit shape.  The 16-bit logits are
             upcast exactly and the ops run in float32, as every 16-bit golden of this repository: the
             reference's own bf16 log_softmax output differs from it by bf16 rounding only (bar 3e-2, tested elsewhere).
Nothing of the reference package is imported: these are torch ops on the build's synthetic logits.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def golden_parity_llama(out):
    from tests import synth

    B, V = 512, 128256
    x = torch.from_numpy(synth.logits(23, B, V)).to(torch.bfloat16).to(torch.float32)
    masks = torch.from_numpy(synth.binary_masks(23, 2, V))
    mid = torch.arange(B) % 2
    logps = torch.log_softmax(x, -1)
    masked = logps + masks[mid]
    logZ = masked.logsumexp(-1)
    g = torch.Generator()
    g.manual_seed(2025)
    p = (masked - logZ[:, None]).exp()
    tok = torch.multinomial(p, 1, generator=g).flatten()
    g.manual_seed(2025)
    q = torch.empty(B, V).exponential_(1, generator=g)
    top2 = (p / q).topk(2, -1).values
    out["parity512_llama::logZ"] = logZ.numpy()
    out["parity512_llama::lse"] = x.logsumexp(-1).numpy()
    out["parity512_llama::token"] = tok.numpy().astype(np.int32)
    out["parity512_llama::margin"] = ((top2[:, 0] - top2[:, 1]) / top2[:, 0]).numpy()


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    out = {}
    golden_parity_llama(out)
    np.savez_compressed(os.path.join(OUT, "ref_round6.npz"), **out)
    print("ref_round6.npz:", {k: v.shape for k, v in out.items()}, "smallest margin", float(out["parity512_llama::margin"].min()))
