"""The C-ABI shared library loads on a CPU-only host and exports every symbol include/glb.h declares
(no compute calls here); the product has no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import genlm_backend_amd
    from genlm_backend_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "glb.h")).read()
    declared = set(re.findall(r"\b(glb_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"glb_step_args", "glb_mt19937"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.glb_abi_version() == 9
    assert b"gfx950" in lib.glb_version()
    assert C.sizeof(_lib.StepArgs) == 224  # layout guard of glb_step_args
    # ... and of the other argument blocks (sizeof in C, include/glb.h compiled with gcc: 264 / 200 / 144 / 160 / 136)
    assert (C.sizeof(_lib.KvPlanArgs), C.sizeof(_lib.TrieArgs), C.sizeof(_lib.TriePlan), C.sizeof(_lib.TrieRowsArgs),
            C.sizeof(_lib.MtRowsArgs)) == (264, 200, 144, 160, 136)


def test_argument_errors_do_not_touch_the_gpu():
    from genlm_backend_amd import _lib

    lib = _lib.load()
    a = _lib.StepArgs()
    a.struct_size = 1
    assert lib.glb_logprob_mask_sample(C.byref(a), None) == _lib.GLB_EINVAL
    assert "struct_size" in _lib.last_error()
    a.struct_size = C.sizeof(_lib.StepArgs)
    assert lib.glb_logprob_mask_sample(C.byref(a), None) == _lib.GLB_EINVAL
    assert lib.glb_log_softmax_rows(None, 0, 1, 1, 1, 1.0, None, 0, 0, None, None, 0, None) == _lib.GLB_EINVAL
    assert lib.glb_workspace_check(None, None) == _lib.GLB_EINVAL
    assert lib.glb_workspace_error_word(None) is None  # not registered
    assert lib.glb_set_spin_limit(0) == _lib.GLB_OK
    assert lib.glb_mask_prepare(None, 1, 64, 2, 0, None, 0, None) == _lib.GLB_EINVAL
    assert lib.glb_step_workspace_bytes(1024, 1024, 50257, 2) >= 1024 * 13 * 32
    assert lib.glb_mask_prepared_bytes(2, 50257) >= 2 * 13 * 512
    assert lib.glb_log_softmax_workspace_bytes(8, 50257) >= 8 * 13 * 32
    ta, tp = _lib.TrieRowsArgs(), _lib.TriePlan()
    assert lib.glb_trie_rows(None, None, None) == _lib.GLB_EINVAL
    assert lib.glb_trie_rows(C.byref(ta), C.byref(tp), None) == _lib.GLB_EINVAL and "struct_size" in _lib.last_error()
    ta.struct_size, tp.struct_size = C.sizeof(_lib.TrieRowsArgs), C.sizeof(_lib.TriePlan)
    ta.n_rows = 1
    assert lib.glb_trie_rows(C.byref(ta), C.byref(tp), None) == _lib.GLB_EINVAL  # no weights
    assert lib.glb_trie_rows_workspace(0, C.byref(tp)) == 0 and lib.glb_trie_rows_workspace(4, C.byref(tp)) >= 4 << 20
    assert lib.glb_group_contexts(None, None, None, 4, None, None, None, None, None, 0, None) == _lib.GLB_EINVAL
    assert lib.glb_group_contexts_workspace(1024) >= 1024 * 4 * 4


def test_host_rng_helpers_match_oracle(oracle):
    from genlm_backend_amd import _lib

    lib = _lib.load()
    st = _lib.MT19937()
    lib.glb_mt19937_seed(C.byref(st), 1234)
    out = np.empty(5000, np.float32)
    assert lib.glb_mt19937_exponential_f32(C.byref(st), out.ctypes.data_as(C.c_void_p), 5000) == 0
    want, _ = oracle.mt_exponential(1234, 5000)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    c = (C.c_uint32 * 4)(1, 2, 3, 4)
    k = (C.c_uint32 * 2)(5, 6)
    o = (C.c_uint32 * 4)()
    lib.glb_philox4x32_10(c, k, o)
    assert list(o) == oracle.philox([1, 2, 3, 4], [5, 6])


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from genlm_backend_amd.engine import HipEngine

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipEngine("cuda:0")


def test_product_never_imports_the_oracle():
    """The product may mention the oracle in comments, but must not import, include or link it."""
    import re

    pkg = os.path.join(ROOT, "genlm-backend_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|(libglb_oracle)|(#include\s*[<\"].*oracle)|(orc_[a-z_0-9]+\s*\()", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), (dirpath, f)
