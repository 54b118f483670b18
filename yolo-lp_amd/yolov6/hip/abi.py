"""ctypes binding of libyololp_hip.so (C ABI declared in include/lp_hip.h).

The library is the only implementation of the GPU path: if it cannot be loaded
this module raises -- nothing falls back to eager torch ops on a GPU.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_void_p  # noqa: F401

LP_F16, LP_BF16, LP_F32 = 0, 1, 2
LP_ACT_NONE, LP_ACT_RELU, LP_ACT_SILU = 0, 1, 2
LP_PRED_COLS, LP_DET_COLS, LP_MAX_SRC = 290, 28, 4
LP_VARIANT_STREAM64, LP_VARIANT_STREAM128, LP_VARIANT_ROWS = 16, 17, 18   # lp_engine_set_op_variant codes beyond the tiles
LP_VARIANT_PIPE_D, LP_VARIANT_PIPE_B, LP_VARIANT_PIPE_F, LP_VARIANT_PIPE_C = 32, 33, 34, 35        # pipelined 3x3 stride-1 kernel (nbuf 3)
LP_VARIANT_PIPE16_D, LP_VARIANT_PIPE16_F = 39, 41                              # the same on v_mfma_f32_16x16x32 (another fp32 summation order)
LP_VARIANT_PIPE16_V0, LP_VARIANT_PIPE16_V1 = 42, 43                           # ... with tiles of any number of 16-pixel blocks
LP_VARIANT_PIPE16_S2A, LP_VARIANT_PIPE16_S2B = 48, 49                         # 3x3 stride 2 on v_mfma_f32_16x16x32 (two-slot ring, 16-pixel blocks)
LP_VARIANT_FUSED_BIFUSION = 45                                                                   # BiFusion's transposed conv + cv1 + cv3 as one kernel
LP_VARIANT_BOX_SPARSE, LP_VARIANT_BOX_DENSE = 46, 47                                                    # head_box in the detections-only forward: candidates only / every anchor
LP_VARIANT_FUSED_PW_S2 = 38                                                                      # a 1x1 layer + the 3x3 stride-2 layer behind it as one kernel
LP_VARIANT_FUSED_STEM2 = 37                                                                      # input op + stem + the layer behind it as one kernel
LP_VARIANT_PIPE_P = 36                                                                           # the stem reading the NCHW frame itself
LP_EVAL_NCOUNTS = 43   # lp_eval_counts: length of the counts vector (include/lp_hip.h)

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # .../yolo-lp_amd
LIB_PATH = os.environ.get('LP_HIP_LIB') or os.path.join(_PKG_ROOT, 'libyololp_hip.so')   # LP_HIP_LIB: debug builds
CSRC_DIR = os.path.join(_PKG_ROOT, 'csrc')


class ConvDesc(ctypes.Structure):
    """lp_conv_desc"""
    _fields_ = [('n_src', c_int), ('src', c_int * LP_MAX_SRC), ('dst', c_int), ('ksize', c_int), ('stride', c_int),
                ('act', c_int), ('res', c_int), ('res_alpha', c_float), ('weight', c_void_p), ('bias', c_void_p), ('dst2', c_int)]


#: name -> (restype, argtypes): every symbol include/lp_hip.h declares
SYMBOLS = {
    'lp_version': (c_char_p, []),
    'lp_last_error': (c_char_p, []),
    'lp_engine_create': (c_int, [POINTER(c_void_p), c_int]),
    'lp_engine_destroy': (None, [c_void_p]),
    'lp_engine_tensor': (c_int, [c_void_p, c_int, c_int]),
    'lp_engine_set_lane': (c_int, [c_void_p, c_int]),
    'lp_engine_add_input': (c_int, [c_void_p, c_int]),
    'lp_engine_add_conv': (c_int, [c_void_p, POINTER(ConvDesc)]),
    'lp_engine_add_deconv2x2': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'lp_engine_add_pool5_chain': (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    'lp_engine_add_head_cls': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'lp_engine_add_head_box': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'lp_engine_finalize': (c_int, [c_void_p, c_int]),
    'lp_engine_weight_bytes': (c_size_t, [c_void_p]),
    'lp_engine_upload': (c_int, [c_void_p, c_void_p, c_void_p]),
    'lp_engine_arena_bytes': (c_size_t, [c_void_p, c_int, c_int, c_int]),
    'lp_engine_bind': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int]),
    'lp_engine_tensor_info': (c_int, [c_void_p, c_int, POINTER(c_size_t), POINTER(c_int), POINTER(c_int),
                                      POINTER(c_int), POINTER(c_int)]),
    'lp_engine_num_anchors': (c_int, [c_void_p]),
    'lp_engine_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'lp_engine_set_graph': (c_int, [c_void_p, c_int]),
    'lp_engine_set_single_lane': (c_int, [c_void_p, c_int]),
    'lp_engine_set_mfma16': (c_int, [c_void_p, c_int]),
    'lp_engine_op_carrier': (c_int, [c_void_p, c_int, c_int]),
    'lp_engine_num_ops': (c_int, [c_void_p]),
    'lp_engine_op_info': (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                  POINTER(c_double), POINTER(c_double)]),
    'lp_engine_forward_det': (c_int, [c_void_p, c_void_p, c_int, ctypes.c_double, c_void_p, c_size_t, c_void_p]),
    'lp_nms_candidates': (c_int, [c_int, c_int, ctypes.c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'lp_engine_profile': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, POINTER(c_float), c_int]),
    'lp_engine_profile_ops': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, POINTER(c_float), c_int, c_int]),
    'lp_engine_autotune': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]),
    'lp_engine_op_variant': (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]),
    'lp_engine_set_op_variant': (c_int, [c_void_p, c_int, c_int, c_int]),
    'lp_engine_copy_tuning': (c_int, [c_void_p, c_void_p]),
    'lp_nms_workspace_bytes': (c_size_t, [c_int, c_int]),
    'lp_nms_candidate_counts': (c_void_p, [c_void_p, c_int, c_int]),
    'lp_preprocess_letterbox': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p]),
    'lp_rescale_round': (c_int, [c_void_p, c_int, c_double, c_double, c_double, c_int, c_int, c_void_p]),
    'lp_eval_counts': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'lp_check_sigmoid_monotone': (c_int, [c_void_p, c_void_p]),
    'lp_debug_poison_lds': (c_int, [c_void_p]),
    'lp_check_iou_predicate': (c_int, [c_void_p, ctypes.c_longlong, c_double, c_void_p, c_void_p]),
    'lp_plan_stem_tile': (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'lp_plan_block_tile': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'lp_nms': (c_int, [c_void_p, c_int, c_int, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_size_t, c_void_p]),
}

_lib = None


def load():
    """Load the library once and attach prototypes; raises RuntimeError when it is missing."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64; importing it first makes this library bind to that same HIP
        # runtime (one runtime per process: a second copy loaded from /opt/rocm finds no device)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('HIP extension %s is missing: build it with `make -C %s` (or '
                               '`python -c "import __graft_entry__ as g; g.build()"`). The GPU path has no '
                               'eager fallback.' % (LIB_PATH, CSRC_DIR))
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError here = header and library are out of sync
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc, what=''):
    """Raise RuntimeError with the library's message for a negative lp_status."""
    if rc < 0:
        msg = load().lp_last_error()
        raise RuntimeError('%s failed (%d): %s' % (what or 'libyololp_hip', rc, msg.decode() if msg else '?'))
    return rc
