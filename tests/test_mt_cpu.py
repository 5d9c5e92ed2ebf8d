"""The MT19937 machinery of the parity draw on the host (no GPU): the library's serial stream equals the oracle's (libm's
log1p, pinned against torch in test_oracle.py) - i.e. glibc's log1p restated in csrc/glb_log1p.hpp is glibc's log1p -, and
the jump polynomials (Berlekamp-Massey, t^J mod phi) move a window exactly as far as stepping the generator does."""
import ctypes as C

import numpy as np

import genlm_backend_amd  # noqa: F401
from genlm_backend_amd import _lib


def _serial_words(seed, n):
    """Untempered words x[0 .. n) of the generator seeded with `seed` (x[0..623] = the seeded array)."""
    x = np.zeros(n + 624 + 227, np.uint32)
    x[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        x[i] = (1812433253 * (int(x[i - 1]) ^ (int(x[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    # the recurrence reaches back 227 words: blocks of 227 are independent
    for base in range(0, n, 227):
        a, b, c = x[base:base + 227], x[base + 1:base + 228], x[base + 397:base + 624]
        y = (a & np.uint32(0x80000000)) | (b & np.uint32(0x7FFFFFFF))
        x[base + 624:base + 851] = c ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))
    return x[:n]


def test_library_stream_equals_the_oracle_stream(oracle):
    """20 million variates through glibc's log1p restated (csrc/glb_log1p.hpp) against the C library's own (the oracle)."""
    lib = _lib.load()
    n = 20_000_000
    st = _lib.MT19937()
    lib.glb_mt19937_seed(C.byref(st), 1234)
    got = np.empty(n, np.float32)
    assert lib.glb_mt19937_exponential_f32(C.byref(st), got.ctypes.data_as(C.c_void_p), n) == 0
    want, _ = oracle.mt_exponential(1234, n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_short_form_of_the_exponential_gives_the_c_librarys_float(tmp_path):
    """csrc/glb_log1p.hpp compiled with g++ against the C library's log1p (tests/native/log1p_check.cpp), 20 million
    arguments and the edges: the exact form - glibc's algorithm without the division that is 0 / u for the stream's
    arguments - gives the library's float every time; the short form (table + series: what the device runs for all but one
    value in 10^5) gives the same float whenever it does not ask for the exact one, and asks rarely."""
    import os
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        import pytest

        pytest.skip("no g++")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "log1p_check.cpp")
    exe = str(tmp_path / "log1p_check")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    n, redo, undetected, exact_bad = (int(x) for x in subprocess.run([exe, "20000000"], check=True, capture_output=True,
                                                                     text=True).stdout.split())
    assert n > 20_000_000 and undetected == 0 and exact_bad == 0
    assert redo < n // 20_000  # (one in 10^5 expected)


def test_window_is_the_seeded_state_and_jumps_equal_stepping():
    lib = _lib.load()
    V = 50257
    stride = 2 * V
    n_small, n_big = 4, 3
    polys = np.zeros((n_small + n_big, _lib.MT_POLY_WORDS), np.uint64)
    assert lib.glb_mt19937_jump_polys(stride, n_small, n_big, polys.ctypes.data_as(C.c_void_p)) == 0
    assert polys[0, 0] == 1 and not polys[0, 1:].any() and polys[n_small, 0] == 1
    assert not (polys[:, 311] >> np.uint64(33)).any()  # degree < 19937
    seed = 99
    win = np.zeros(624, np.uint32)
    assert lib.glb_mt19937_window(seed, win.ctypes.data_as(C.c_void_p)) == 0
    far = n_small * (n_big - 1) * stride + 3 * stride
    x = _serial_words(seed, far + 624 + 8)
    assert np.array_equal(win, x[:624])
    out = np.zeros(624, np.uint32)

    def jump(src, p):
        assert lib.glb_mt19937_jump_host(src.ctypes.data_as(C.c_void_p), polys[p].ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p)) == 0
        return out.copy()

    def same(w, o):  # only the top bit of a window's word 0 is state
        return np.array_equal(w[1:], x[o + 1:o + 624]) and (int(w[0]) >> 31) == (int(x[o]) >> 31)

    for r in range(n_small):
        assert same(jump(win, r), r * stride), r
    for m in range(n_big):
        big = jump(win, n_small + m)
        assert same(big, m * n_small * stride), m
        for r in (1, n_small - 1):  # two levels: what the device does
            assert same(jump(big, r), (m * n_small + r) * stride), (m, r)


def test_polys_of_another_stride_and_argument_errors():
    lib = _lib.load()
    polys = np.zeros((2 + 1, _lib.MT_POLY_WORDS), np.uint64)
    assert lib.glb_mt19937_jump_polys(1, 2, 1, polys.ctypes.data_as(C.c_void_p)) == 0
    assert polys[1, 0] == 2 and not polys[1, 1:].any()  # t^1
    assert lib.glb_mt19937_jump_polys(0, 2, 1, polys.ctypes.data_as(C.c_void_p)) == _lib.GLB_EINVAL
    assert lib.glb_mt19937_jump_polys(7, 1, 1, polys.ctypes.data_as(C.c_void_p)) == _lib.GLB_EINVAL
    assert lib.glb_mt19937_jump_polys(7, 2, 1, None) == _lib.GLB_EINVAL
    # (windows as partial planes: 16 / 4; behind them one word per row window: which windows a call's rows read - round 6)
    assert lib.glb_mt19937_rows_workspace(1024, 32) == (33 * 16 + 33 * 32 * 4) * 624 * 4 + 33 * 32 * 4
