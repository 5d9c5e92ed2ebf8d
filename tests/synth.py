"""Seeded synthetic inputs shared by the CPU and GPU tests (SURVEY.md §8d config 2/5 shapes)."""
import numpy as np


def logits(seed, B, V, scale=3.0, outliers=3):
    """~N(0, scale^2) with a few +-30 outliers per row (exercises max subtraction)."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, V), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    for b in range(B):
        idx = rng.integers(0, V, size=outliers)
        x[b, idx] += rng.choice(np.array([-30.0, 30.0], dtype=np.float32), size=outliers)
    return x


def binary_masks(seed, K, V, p_forbid=1.0 / 3.0):
    rng = np.random.default_rng(seed + 1000)
    m = np.where(rng.random((K, V)) < p_forbid, -np.inf, 0.0).astype(np.float32)
    m[:, 0] = 0.0  # never empty
    return m


def contexts(seed, n, n_distinct, lo=1, hi=12, vocab=1000):
    rng = np.random.default_rng(seed)
    pool = [list(rng.integers(0, vocab, size=rng.integers(lo, hi + 1)).astype(int)) for _ in range(n_distinct)]
    return [list(pool[i]) for i in rng.integers(0, n_distinct, size=n)]
