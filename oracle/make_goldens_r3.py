"""Round-3 goldens: tests/golden/ref_round3.npz, produced by running the REFERENCE (genlm/genlm-backend, imported
read-only from /root/reference) - `AsyncLM.batch_sample` (base.py:148-179) on the tiny GPT-2 of ref_hotpath_tiny.npz
with ragged prompts, a temperature, TWO stopping tokens and sequences that end at different steps.  Data only; this
script is the committed recipe.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python -B /root/repo/oracle/make_goldens_r3.py
"""
import asyncio
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (shims, tiny model, reference_llm)

PROMPTS = [[3, 1, 4, 1, 5], [9, 9], [2, 7, 1, 8, 2, 8], [100], [6, 6, 6], [31, 41, 59, 26], [5, 3, 5, 8, 9, 7, 9],
           [11, 12], [3, 1, 4, 1, 5], [77, 78, 79]]
MAX_TOKENS, TEMP, SEED = 10, 0.2, 2024  # (a low temperature: at T ~ 1 a random-init model is so flat that the shared noise row decides alone)


def main():
    mg.install_shims()
    model = mg.tiny_model(0)  # the weights stored in ref_hotpath_tiny.npz
    out = {}
    # pass 1: no stopping tokens - see what the seeded chains generate, then pick two stopping tokens that cut
    # different sequences at different steps (an id from an early step of one chain, one from a late step of another)
    llm = mg.reference_llm(model)
    free = asyncio.run(llm.batch_sample(PROMPTS, max_tokens=MAX_TOKENS, eos_token_ids=[], temperature=TEMP, seed=SEED))
    def cut(row, eos):
        for n, t in enumerate(row):
            if t in eos:
                return n
        return len(row)

    toks = sorted({t for r in free for t in r})
    best = None
    for a in toks:  # the first pair of stopping tokens (ascending ids) that gives the most distinct lengths
        for b in toks:
            if b <= a:
                continue
            k = len({cut(r, (a, b)) for r in free})
            if best is None or k > best[0]:
                best = (k, a, b)
    eos = [int(best[1]), int(best[2])]
    llm = mg.reference_llm(model)
    ids = asyncio.run(llm.batch_sample(PROMPTS, max_tokens=MAX_TOKENS, eos_token_ids=eos, temperature=TEMP, seed=SEED))
    lens = [len(r) for r in ids]
    assert len(set(lens)) >= 3 and min(lens) < MAX_TOKENS, lens  # sequences end at different steps
    out["bs_prompts"] = np.array([p + [-1] * (8 - len(p)) for p in PROMPTS], np.int32)
    out["bs_eos"] = np.array(eos, np.int32)
    out["bs_params"] = np.array([MAX_TOKENS, SEED], np.int64)
    out["bs_temperature"] = np.array([TEMP], np.float64)
    out["bs_ids"] = np.array([r + [-1] * (MAX_TOKENS - len(r)) for r in ids], np.int32)
    out["bs_ids_free"] = np.array(free, np.int32)
    np.savez_compressed(os.path.join(mg.OUT, "ref_round3.npz"), **out)
    print("ref_round3.npz:", {k: v.shape for k, v in out.items()}, "lengths", lens, "eos", eos)


if __name__ == "__main__":
    main()
