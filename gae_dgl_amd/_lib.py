"""ctypes binding of libgae_hip.so (the C ABI declared in include/gae_hip.h).

There is NO fallback: if the shared library is missing or cannot be loaded
every op raises -- the product path never routes through PyTorch/CPU code."""
import collections
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GAE_HIP_LIB: load another build of the same library (kernel experiments, tools/r02/bce_whatif.sh); never a fallback
LIB_PATH = os.environ.get("GAE_HIP_LIB") or os.path.join(HERE, "lib", "libgae_hip.so")

F32, BF16, U8 = 0, 1, 2
SPMM_STORE_PAD = 1
SPMM_TILE = 2
SPMM_ACCUMULATE = 4
SPMM_SKIP_ROWS = 8
SPMM_ELL_WIDTH = 16
ACT_IDENTITY, ACT_RELU = 0, 1

_i32, _i64, _u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
_p, _f, _int = ctypes.c_void_p, ctypes.c_float, ctypes.c_int


class DeviceInfo(ctypes.Structure):
    _fields_ = [("compute_units", _i32), ("wavefront_size", _i32), ("lds_bytes_per_cu", _i32),
                ("l2_bytes", _i32), ("hbm_bytes", _i64), ("clock_khz", _i32),
                ("gfx_major_minor", _i32), ("name", ctypes.c_char * 64)]


class SpmmPlan(ctypes.Structure):
    _fields_ = [("threshold", _i32), ("segment_edges", _i32), ("n_heavy", _i64), ("n_segments", _i64),
                ("heavy_rows", _p), ("heavy_seg_base", _p), ("seg_heavy", _p), ("ell", _p), ("ell_width", _i32),
                ("reserved", _i32), ("hot_indices", _p), ("vh_n_rows", _i64), ("vh_n_virtual", _i64), ("vh_rows", _p),
                ("vh_indptr", _p), ("vh_indices", _p), ("vh_hot_indices", _p), ("vh_identity", _p),
                ("vh_part_ptr", _p), ("vh_part_pos", _p), ("seg_desc", _p), ("light_desc", _p), ("n_light", _i64),
                ("mid_indices", _p), ("mid_tagged", _i32), ("reserved2", _i32), ("vh_desc", _p), ("skip_rows", _p)]


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("n", _i64),
                ("partials", _p), ("n_partials", _i64), ("partial_stride", _i64), ("row_len", _i64), ("row_pitch", _i64)]


class BceTail(ctypes.Structure):
    """gae_bce_tail: the deferred final reduction of a fused loss call (gae_x_decoder_bce_defer_finalize)"""
    _fields_ = [("dense_partial", _p), ("n_dense", _i64), ("edge_partial", _p), ("n_edge", _i64), ("S", _p),
                ("DP", _i32), ("reserved", _i32), ("pad_terms", ctypes.c_double), ("inv_n2", ctypes.c_double),
                ("loss_out", _p), ("bump_draw", _p), ("scal", _p),
                ("kl_partial", _p), ("n_kl", _i64), ("kl_scale", ctypes.c_double), ("kl_out", _p), ("rec_out", _p)]


class BcePrep(ctypes.Structure):
    """gae_bce_prep: where a producer kernel puts the prepare step's outputs (gae_x_decoder_bce_prep_layout)"""
    _fields_ = [("Zt", _p), ("Zhi", _p), ("Zlo", _p), ("colsum_partial", _p), ("scal", _p),
                ("all_pairs", ctypes.c_double), ("max_blocks", _i64), ("DP", _i32), ("reserved", _i32)]


ADAM_MAX_TENSORS = 16
ADAM_STATE_WORDS = 6      # uint64 words of gae_adam_step's device state

# name -> (restype, argtypes); mirrors include/gae_hip.h one to one
SIGNATURES = {
    "gae_version": (_int, []),
    "gae_last_error": (ctypes.c_char_p, []),
    "gae_device_info_get": (_int, [_int, ctypes.POINTER(DeviceInfo)]),
    "gae_spmm_csr_ep": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _i64, _int, _p, _int, _p]),
    "gae_linear2_fwd": (_int, [_p, _i64, _i64, _i64, _p, _i64, _p, _i64, _int, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _p,
                               _i64, _p]),
    "gae_linear2_fill_dead": (_int, [_p, _i64, _int, _p, _i64, _i64, _p, _i64, _p, _i64, _p]),
    "gae_gcn2_bwd_dense_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "gae_gcn2_bwd_dense": (_int, [_p, _i64, _p, _i64, _p, _i64, _int, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p, _p,
                                  _p, _p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _p, _p]),
    "gae_spmm_plan_sizes": (_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _i64, _p]),
    "gae_spmm_plan_scratch_bytes": (_i64, [_i64, _i64, _i64, _i64, _i32]),
    "gae_spmm_plan_build_rows": (_int, [_p, _p, _i64, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                        _i64, _p, _p]),
    "gae_spmm_plan_build_pinned": (_int, [_p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "gae_tuning_set": (_int, [ctypes.c_char_p, _i64]),
    "gae_tuning_get": (_int, [ctypes.c_char_p, _p]),
    "gae_csr_from_coo_workspace_bytes": (_i64, [_i64, _i64]),
    "gae_csr_from_coo": (_int, [_p, _p, _i64, _i64, _i64, _p, _p, _p, _i64, _p, _p]),
    "gae_degree_norm": (_int, [_p, _i64, _p, _p, _p]),
    "gae_rows_pack": (_int, [_p, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "gae_csr_to_dense": (_int, [_p, _p, _i64, _i64, _p, _i64, _p]),
    "gae_batch_plan": (_int, [_p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "gae_batch_select": (_int, [_p, _i64, _p, _i64, _p, _p]),
    "gae_x_batch_plan_next": (_int, [_p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p]),
    "gae_batch_gather": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _i64, _p, _p, _i64, _i64,
                                _p, _p, _p, _i64, _p, _i32, _i64, _p, _p]),
    "gae_x_batch_gather_next": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64,
                                     _p, _p, _p, _i64, _p, _i32, _p, _p]),
    "gae_decoder_bce_padded": (_int, [_p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _f, _u64, _u64, _p, _p, _p, _i64,
                                      _p, _i64, _p]),
    "gae_bce_logits_workspace_bytes": (_i64, []),
    "gae_bce_logits": (_int, [_p, _i64, _p, _i64, _i64, _i64, _f, _p, _p, _i64, _p, _i64, _p]),
    "gae_segment_readout": (_int, [_p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p]),
    "gae_spmm_ell_build": (_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    "gae_spmm_workspace_bytes": (_i64, [ctypes.POINTER(SpmmPlan), _i64]),
    "gae_spmm_csr": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _int, _p, _p,
                            ctypes.POINTER(SpmmPlan), _p, _i64, _int, _p]),
    "gae_spmm_blockdiag_lds_bytes": (_i64, [_i64, _i64, _i64]),
    "gae_gcn_layer_fused": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _p, _i64, _int,
                                   _p, _i64, _p]),
    "gae_spmm_csr_blockdiag": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _int,
                                      _p]),
    "gae_spmm_csr_epilogue": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _p, _i64, _i64, _p, _p, ctypes.POINTER(SpmmPlan),
                                     _p, _int, _i64, _i64, _p]),
    "gae_xw_fwd_splits": (_i64, [_i64, _i64, _i64, _int]),
    "gae_xw_usable": (_int, [_p, _i64, _int, _i64, _i64, _i64]),
    "gae_xw_fwd_workspace_bytes": (_i64, [_i64, _i64, _i64, _int]),
    "gae_xw_fwd": (_int, [_p, _i64, _int, _i64, _i64, _p, _i64, _p, _i64, _int, _p, _i64, _p, _i64, _int, _p]),
    "gae_xw_wgrad_workspace_bytes": (_i64, [_i64, _i64, _int]),
    "gae_xw_wgrad": (_int, [_p, _i64, _int, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _p,
                            _p, _i64, _p]),
    "gae_dense_to_csr_count": (_int, [_p, _i64, _i64, _i64, _p, _p]),
    "gae_dense_to_csr_fill": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p]),
    "gae_spx_fwd_workspace_bytes": (_i64, [_i64]),
    "gae_spx_fwd": (_int, [_p, _p, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _p]),
    "gae_spx_wgrad_layout": (_int, [_i64, _i64, _i64, _p]),
    "gae_spx_wgrad": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _i64,
                             _p, _int, _p, _i64, _p]),
    "gae_linear_fwd_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gae_linear_fwd": (_int, [_p, _i64, _i64, _i64, _p, _p, _i64, _int, _p, _i64, _p, _i64, _p]),
    "gae_linear_bwd_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gae_x_linear_bwd_partials": (_int, [_p, _i64, _p, _i64, _int, _p, _i64, _i64, _i64, _i64, _int, _int, _p, _i64, _p, _p]),
    "gae_x_xw_wgrad_partials": (_int, [_p, _i64, _int, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _int, _int,
                                     _p, _i64, _p, _p]),
    "gae_linear_bwd": (_int, [_p, _i64, _p, _i64, _int, _p, _i64, _p, _i64, _i64, _i64,
                              _p, _p, _p, _i64, _p, _i64, _p]),
    "gae_dropout_mask": (_int, [_p, _i64, _f, _u64, _u64, _p, _p]),
    "gae_normal_noise": (_int, [_p, _i64, _u64, _u64, _p, _p]),
    "gae_vgae_head_workspace_bytes": (_i64, [_i64]),
    "gae_vgae_head_fwd": (_int, [_p, _p, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _p]),
    "gae_vgae_head_bwd": (_int, [_p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _p, _p]),
    "gae_x_gcn_layer_fused2": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _i64, _int, _i64,
                                    _i64, _p, _p, _i64, _int, _p, _i64, _p]),
    "gae_decoder_dense": (_int, [_p, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "gae_decoder_dense_bwd_workspace_bytes": (_i64, [_i64, _i64]),
    "gae_decoder_dense_bwd": (_int, [_p, _i64, _p, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p]),
    "gae_adam_step": (_int, [ctypes.POINTER(AdamTensor), _i32, _f, _f, _f, _f, _f, _p, _p]),
    "gae_x_gcn_layer_fused_wgrad_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gae_x_gcn_layer_fused_wgrad": (_int, [_p, _p, _i64, _p, _i64, _i64, _p, _p, ctypes.POINTER(SpmmPlan), _p, _i64, _i64, _p,
                                         _i64, _p, _i64, _p, _p, _p, _i64, _p, _p]),
    "gae_x_decoder_bce_prep_layout": (_int, [_i64, _i64, _p, _i64, ctypes.POINTER(BcePrep)]),
    "gae_x_gcn_layer_fused_prep": (_int, [_p, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _p, ctypes.POINTER(SpmmPlan), _p, _i64,
                                        _i64, _p, _i64, _p, _i64, ctypes.POINTER(BcePrep), _p, _i64, _f, _u64, _u64, _p, _p,
                                        _p, _p]),
    "gae_x_decoder_bce_prepared": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _f, _p, _f, _p, _i64, _p, _p, _i64, _p, _i64,
                                        _p]),
    "gae_x_vgae_head_prep": (_int, [_p, _p, _i64, _p, _int, _u64, _u64, _p, _i64, _i64, _p, ctypes.POINTER(BcePrep), _p, _i64,
                                  _p, _p]),
    "gae_x_gcn_layer_fused2_wgrad": (_int, [_p, _p, _i64, _p, _i64, _i64, _p, _p, ctypes.POINTER(SpmmPlan), _p, _p, _i64, _i64,
                                          _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _p]),
    "gae_x_adam_step_tail": (_int, [ctypes.POINTER(AdamTensor), _i32, _f, _f, _f, _f, _f, _p, ctypes.POINTER(BceTail), _p]),
    "gae_x_decoder_bce_defer_finalize": (_int, [ctypes.POINTER(BceTail)]),
    "gae_x_decoder_bce_finalize": (_int, [ctypes.POINTER(BceTail), _p]),
    "gae_decoder_bce_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gae_decoder_bce_rows": (_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _f, _f, _u64, _u64, _p,
                                    _p, _p, _i64, _p, _i64, _p]),
    "gae_decoder_bce": (_int, [_p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _f, _f, _u64, _u64, _p, _p, _p, _i64, _p,
                               _i64, _p]),
}

_lib = None


class GaeHipError(RuntimeError):
    pass


def load():
    """Load libgae_hip.so (once).  Raises GaeHipError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GaeHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(gae_dgl_amd has no CPU / PyTorch fallback)")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise GaeHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gae_last_error().decode(errors="replace")
        kind = "argument error" if rc < 0 else "hipError_t"
        raise GaeHipError(f"{what} failed ({kind} {rc}): {msg}")


CALLS = collections.Counter()      # entry point -> calls made through call() by this process (tests assert on it: which
                                   # kernels a step really went through; a HIP-graph replay makes none)


def call(name, *args):
    CALLS[name] += 1
    check(getattr(load(), name)(*args), name)


def tuning_get(name):
    """current value of a gae_tuning_set knob (or of a telemetry value such as "bce_last_kind")"""
    out = ctypes.c_int64(0)
    check(load().gae_tuning_get(name.encode(), ctypes.byref(out)), "gae_tuning_get")
    return int(out.value)
