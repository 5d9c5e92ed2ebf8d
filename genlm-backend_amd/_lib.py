"""ctypes binding of libglb_hip.so (C ABI: include/glb.h).

There is no fallback: if the shared object is missing or cannot be loaded this module raises, and
every compute entry point raises when the HIP runtime sees no device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglb_hip.so")

ABI_VERSION = 9
GLB_OK, GLB_EINVAL, GLB_EUNSUPPORTED, GLB_EHIP, GLB_ENOSPC = 0, 1, 2, 3, 4
F32, BF16, F16 = 0, 1, 2
MASK_NONE, MASK_BITS, MASK_F32, MASK_PREPARED = 0, 1, 2, 3
RNG_NONE, RNG_PHILOX, RNG_NOISE = 0, 1, 2
STEP_HW_EXP = 1  # glb_step_args.flags: the hardware-exponential contract of 16-bit rows


class GlbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"glb error {code}: {msg}")
        self.code = code


class StepArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("logits", C.c_void_p),
        ("dtype", C.c_int32),
        ("n_rows", C.c_int64),
        ("vocab", C.c_int64),
        ("ld", C.c_int64),
        ("logit_scale", C.c_float),
        ("n_particles", C.c_int64),
        ("row_of", C.c_void_p),
        ("mask_kind", C.c_int32),
        ("mask", C.c_void_p),
        ("mask_ld", C.c_int64),
        ("n_masks", C.c_int64),
        ("mask_id", C.c_void_p),
        ("row_mask_id", C.c_void_p),
        ("rng_mode", C.c_int32),
        ("noise", C.c_void_p),
        ("noise_ld", C.c_int64),
        ("seed", C.c_uint64),
        ("offset", C.c_uint64),
        ("particle_base", C.c_int64),
        ("out_logZ", C.c_void_p),
        ("out_lse", C.c_void_p),
        ("out_token", C.c_void_p),
        ("out_margin", C.c_void_p),
        ("flags", C.c_int32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


class TriePlan(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + [(k, C.c_int32) for k in ("n_parts", "n_top", "n_cut", "n_slots", "max_local", "top_base",
                                                                          "lds_bytes")] + [
        ("n_nodes", C.c_int64)] + [(k, C.c_void_p) for k in ("desc", "idepth", "leaf_src", "leaf_local", "run_tab", "top_local", "slot_of",
                                                             "cptr16", "inode16", "pn_local16", "tok_local16", "inode64")] + [
        ("lds_top_bytes", C.c_int32), ("vocab", C.c_int32)]


class TrieRowsArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("weights", C.c_void_p),
        ("dtype", C.c_int32),
        ("ld", C.c_int64),
        ("n_rows", C.c_int64),
        ("vocab", C.c_int64),
        ("lse", C.c_void_p),
        ("logit_scale", C.c_float),
        ("from_logprobs", C.c_int32),
        ("op", C.c_int32),
        ("out_slots", C.c_void_p),
        ("out_slots_ld", C.c_int64),
        ("out_nodes", C.c_void_p),
        ("out_nodes_ld", C.c_int64),
        ("sel_nodes", C.c_void_p),
        ("n_sel", C.c_int64),
        ("out_sel", C.c_void_p),
        ("out_sel_ld", C.c_int64),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
        ("sel_row_stride", C.c_int64),
    ]


class TrieArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("weights", C.c_void_p),
        ("dtype", C.c_int32),
        ("ld", C.c_int64),
        ("n_rows", C.c_int64),
        ("vocab", C.c_int64),
        ("lse", C.c_void_p),
        ("logit_scale", C.c_float),
        ("from_logprobs", C.c_int32),
        ("op", C.c_int32),
        ("n_nodes", C.c_int64),
        ("n_levels", C.c_int64),
        ("leaf_node", C.c_void_p),
        ("level_start_host", C.c_void_p),
        ("level_nodes", C.c_void_p),
        ("child_ptr", C.c_void_p),
        ("child_idx", C.c_void_p),
        ("out", C.c_void_p),
        ("out_ld", C.c_int64),
        ("sel_nodes", C.c_void_p),
        ("n_sel", C.c_int64),
        ("out_sel", C.c_void_p),
        ("out_sel_ld", C.c_int64),
        ("keep_node_major", C.c_int32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


class KvPlanArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n", C.c_int64), ("n_rows", C.c_int64), ("cap", C.c_int64),
        ("group_of", C.c_void_p), ("rep", C.c_void_p), ("n_groups", C.c_void_p),
        ("old_row", C.c_void_p),
        ("old_row_by_context", C.c_int32),
        ("lengths", C.c_void_p),
        ("row_stamps", C.c_void_p),
        ("call_no", C.c_int64),
        ("row_tok", C.c_void_p), ("row_len", C.c_void_p),
        ("row_hash", C.c_void_p),
        ("group_hash", C.c_void_p),
        ("tokens", C.c_void_p),
        ("starts", C.c_void_p),
        ("out_group_row", C.c_void_p), ("out_logits_row", C.c_void_p), ("out_rows_a", C.c_void_p), ("out_ctx_a", C.c_void_p),
        ("out_pos_a", C.c_void_p), ("out_ctx_b", C.c_void_p), ("out_rows_b", C.c_void_p), ("out_copy_src", C.c_void_p),
        ("out_copy_len", C.c_void_p), ("out_ctx_of_row", C.c_void_p), ("out_pos_of_row", C.c_void_p),
        ("out_row_of_context", C.c_void_p), ("out_head", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


class MT19937(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int32)]


MT_POLY_WORDS = 312


class MtRowsArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("window", C.c_void_p), ("window_out", C.c_void_p), ("polys", C.c_void_p),
        ("n_small", C.c_int32), ("n_big", C.c_int32),
        ("vocab", C.c_int64), ("max_draw_rows", C.c_int64),
        ("n_draw", C.c_void_p),
        ("n_out_rows", C.c_int64),
        ("row_slot", C.c_void_p),
        ("out", C.c_void_p), ("out_ld", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("reuse_windows", C.c_int32), ("rows_from", C.c_void_p), ("rows_from_ld", C.c_int64),
    ]


# every symbol include/glb.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
SYMBOLS = {
    "glb_version": (C.c_char_p, []),
    "glb_abi_version": (C.c_int, []),
    "glb_last_error": (C.c_int, [C.c_char_p, _sz]),
    "glb_device_count": (C.c_int, []),
    "glb_step_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "glb_logprob_mask_sample": (C.c_int, [C.POINTER(StepArgs), _vp]),
    "glb_logprob_mask_sample_timed": (C.c_int, [C.POINTER(StepArgs), _vp, _vp, _vp]),
    "glb_workspace_init": (C.c_int, [_vp, _sz, _vp]),
    "glb_workspace_release": (C.c_int, [_vp]),
    "glb_workspace_check": (C.c_int, [_vp, _vp]),
    "glb_workspace_error_word": (_vp, [_vp]),
    "glb_set_spin_limit": (C.c_int, [C.c_uint64]),
    "glb_mask_prepared_bytes": (_sz, [_i64, _i64]),
    "glb_mask_prepare": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _sz, _vp]),
    "glb_mask_prepare_rows": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp, _sz, _vp]),
    "glb_log_softmax_workspace_bytes": (_sz, [_i64, _i64]),
    "glb_log_softmax_rows": (C.c_int, [_vp, _i32, _i64, _i64, _i64, _f32, _vp, _i32, _i64, _vp, _vp, _sz, _vp]),
    "glb_mask_f32_to_bits": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "glb_group_contexts_workspace": (_sz, [_i64]),
    "glb_group_contexts": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "glb_hash_contexts": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "glb_match_prefixes": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "glb_gather_padded": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "glb_gather_kv_padded": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    "glb_particles_advance": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "glb_normalize_weights": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "glb_kv_append": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "glb_kv_gather_rows": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _i32, _vp]),
    "glb_gather_rows_i32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "glb_match_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "glb_slab_attention": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                     _f32, _i32, _vp, _vp]),
    "glb_short_attention": (C.c_int, [_vp, C.POINTER(C.c_int64), _vp, C.POINTER(C.c_int64), _vp, C.POINTER(C.c_int64), _vp, _i64, _i64,
                                      _i64, _i64, _i64, _i64, _i64, _i64, _f32, _i32, _vp, _vp]),
    "glb_kv_plan_workspace": (_sz, [_i64, _i64]),
    "glb_kv_plan": (C.c_int, [C.POINTER(KvPlanArgs), _vp]),
    "glb_trie_workspace": (_sz, [_i64, _i64]),
    "glb_trie_reduce": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _sz,
                                  _vp]),
    "glb_trie_workspace_ex": (_sz, [_i64, _i64]),
    "glb_trie_masses": (C.c_int, [C.POINTER(TrieArgs), _vp]),
    "glb_trie_rows_workspace": (_sz, [_i64, C.POINTER(TriePlan)]),
    "glb_trie_rows": (C.c_int, [C.POINTER(TrieRowsArgs), C.POINTER(TriePlan), _vp]),
    "glb_resample_workspace": (_sz, [_i64]),
    "glb_resample_systematic": (C.c_int, [_vp, _i64, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _sz, _vp]),
    "glb_mt19937_seed": (None, [C.POINTER(MT19937), C.c_uint64]),
    "glb_mt19937_exponential_f32": (C.c_int, [C.POINTER(MT19937), _vp, _i64]),
    "glb_mt19937_window": (C.c_int, [C.c_uint64, _vp]),
    "glb_mt19937_jump_polys": (C.c_int, [_i64, _i32, _i32, _vp]),
    "glb_mt19937_jump_host": (C.c_int, [_vp, _vp, _vp]),
    "glb_mt19937_rows_workspace": (_sz, [_i64, _i32]),
    "glb_mt19937_exponential_rows": (C.c_int, [C.POINTER(MtRowsArgs), _vp]),
    "glb_comm_unique_id": (C.c_int, [_vp]),
    "glb_comm_init": (C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_void_p)]),
    "glb_allgather_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "glb_comm_destroy": (C.c_int, [_vp]),
    "glb_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}

_lib = None


def load():
    """Load libglb_hip.so and bind every symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C genlm-backend_amd/csrc -j8` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    # torch first: it brings its own HIP runtime, and the runtime that is loaded first serves the process - loading
    # this library (linked against /opt/rocm's) ahead of torch leaves torch.cuda unable to find the device
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.glb_abi_version() != ABI_VERSION:
        raise ImportError(f"libglb_hip.so ABI {lib.glb_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def last_error():
    buf = C.create_string_buffer(512)
    load().glb_last_error(buf, 512)
    return buf.value.decode()


def check(rc):
    if rc != GLB_OK:
        raise GlbError(rc, last_error())
