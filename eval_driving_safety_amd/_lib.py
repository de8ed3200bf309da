"""ctypes binding of libadvengine.so (C ABI: include/advengine.h).

torch is imported BEFORE the library is opened on purpose: PyTorch-ROCm ships its own
libamdhip64.so.7, and the dynamic loader then resolves this library's NEEDED entry of the
same soname to that already-loaded runtime, so device pointers and streams handed over
from torch tensors belong to the runtime the kernels are launched with.
"""
import ctypes
import os

import torch  # noqa: F401  (see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADVENGINE_LIB: tuning builds of the same library (tools/); the default is the in-tree build next to this file
LIB_PATH = os.environ.get("ADVENGINE_LIB") or os.path.join(_HERE, "libadvengine.so")
# the -DADV_TEST_HOOKS build of the same sources (route switches read from the environment): tests/ and tools/ only, via using()
HOOKS_LIB_PATH = os.path.join(_HERE, "libadvengine_hooks.so")

ADV_OK = 0
ADV_EINVAL = -22
ADV_EALIGN = -14
ADV_ELAUNCH = -5
ADV_SPACE_AFFINE = 0
ADV_SPACE_IDENTITY = 1
ADV_SPACE_AFFINE_RCP = 2
ABI_VERSION = 11


class AdvSpace(ctypes.Structure):
    """adv_space_t"""
    _fields_ = [("kind", ctypes.c_int32),
                ("scale", ctypes.c_float * 3),
                ("shift", ctypes.c_float * 3),
                ("lo", ctypes.c_float * 3),
                ("hi", ctypes.c_float * 3),
                ("export_add", ctypes.c_double * 3)]


class AdvCleanIndex(ctypes.Structure):
    """adv_clean_index_t (all pointers are device memory)"""
    _fields_ = [("index", ctypes.c_void_p),
                ("ok", ctypes.c_void_p),
                ("lut", ctypes.c_void_p),
                ("valid_hw", ctypes.c_void_p),
                ("valid_h", ctypes.c_int32),
                ("valid_w", ctypes.c_int32)]


class AdvEngineError(RuntimeError):
    def __init__(self, fn, code, detail):
        super().__init__("%s failed: %s (code %d)" % (fn, detail, code))
        self.code = code


_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float
_SP = ctypes.POINTER(AdvSpace)
_F3 = ctypes.POINTER(ctypes.c_float)
_CI = ctypes.POINTER(AdvCleanIndex)
_I3 = ctypes.POINTER(ctypes.c_int32)

# name -> argtypes; every function returns int unless listed in _OTHER_RESTYPE
SIGNATURES = {
    "adv_denormalize_f32": [_P, _P, _L, _I, _I, _SP, _P],
    "adv_normalize_f32": [_P, _P, _L, _I, _I, _SP, _P],
    "adv_pgd_step_f32": [_P, _P, _P, _P, _P, _L, _I, _I, _SP, _F, _F, _I, _I, _L, _L, _P],
    "adv_export_u8_f32": [_P, _P, _L, _I, _I, _SP, _I, _I, _L, _L, _P],
    "adv_clean_index_build_f32": [_P, _P, _CI, _P, _L, _I, _I, _SP, _I, _I, _L, _L, _P],
    "adv_import_u8_f32": [_P, _L, _L, _P, _P, _CI, _I, _I, _L, _I, _I, _SP, _P],
    "adv_pgd_step_indexed_f32": [_P, _P, _P, _CI, _P, _P, _L, _I, _I, _SP, _F, _F, _I, _I, _L, _L, _P],
    "adv_disc_mask_f32": [_P, _I, _I, _I, _I, _I, _P],
    "adv_patch_paste_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "adv_patch_paste_batch_f32": [_P, _P, _L, _I, _I, _I, _P, _I, _P],
    "adv_patch_update_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _F3, _F3, _P, _P],
    "adv_patch_delta_batch_f32": [_P, _P, _L, _I, _I, _I, _P, _I, _F, _F, _P, _P],
    "adv_patch_apply_f32": [_P, _P, _I, _F3, _F3, _P],
    "adv_psv_build_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "adv_psv_build_bwd_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "adv_psv_build_lerp_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "adv_psv_build_lerp_bwd_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "adv_roi_align_fwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "adv_roi_align_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P],
    "adv_roi_align_bwd_cl_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P],
    "adv_roi_gout_channel_last_f32": [_P, _P, _I, _I, _I, _I, _P],
    "adv_nms_f32": [_P, _I, _F, _P, _P, _P, _P],
    "adv_dense_align_cost_f32": [_P, _P, _I, _I, _I, _P, _P, _I, _P, _F, _F, _I, _P, _P],
    "adv_dense_align_argmin_f32": [_P, _I, _I, _P, _F, _P, _P, _P],
    "adv_conv3d_k3_prep_weights_f32": [_P, _P, _I, _I, _I, _P],
    "adv_conv3d_k3_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv3d_k3_masked_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv3d_k3_s2_stage_channels": [_P, _I, _I],
    "adv_conv3d_k3_ex_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), _I, _I3, _I3, _I3, _P],
    "adv_conv_transpose3d_k3_s2_f32": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv_transpose3d_k3_s2_dgrad_f32": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "adv_space_to_depth2_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "adv_depth_regress_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_depth_regress_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_grid_sample3d_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_grid_sample3d_plan_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_grid_sample3d_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_grid_sample3d_bwd_ws_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_sigmoid_focal_loss_f32": [_P, _P, _P, _P, _L, _I, _F, _F, _P],
    "adv_relu_backward_f32": [_P, _P, _P, _L, _P],
    "adv_box_iou_rows_f32": [_P, _P, _P, _P, _P, _L, _I, _P],
    "adv_box_encode6_f32": [_P, _P, _P, _P, _P, _P, _L, _I, _P],
    "adv_box_decode_stereo_f32": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _P],
    "adv_masked_loss_f32": [_P, _P, _P, _P, _P, _L, _I, _F, _I, _P],
    "adv_masked_loss_bwd_f32": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "adv_objective_chain_f32": [_P, _P, _P, _P, _I, _P],
    "adv_rpn_pack_fwd_f32": [_P, _P, _P, _I, _I, _L, _I, _P],
    "adv_rpn_pack_bwd_f32": [_P, _P, _P, _P, _I, _I, _L, _I, _P],
    "adv_box_partition_stereo_f32": [_P, _P, _P, _P, _P, _P, _I, _P],
    "adv_box_sample_rois_f32": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "adv_stem_pool_fwd_f32": [_P, _P, _P, _P, _L, _I, _I, _I, _P],
    "adv_stem_pool_bwd_f32": [_P, _P, _P, _L, _I, _I, _P],
    "adv_bev_fold_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "adv_bev_fold_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "adv_bilinear_up_f32": [_P, _P, _L, _I, _I, _I, _I, _P],
    "adv_bilinear_up_bwd_f32": [_P, _P, _L, _I, _I, _I, _I, _P],
    "adv_conv2d_1x1_prep_weights_f32": [_P, _P, _I, _I, _I, _P],
    "adv_conv2d_1x1_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P],
    "adv_bias_act_f32": [_P, _P, _P, _L, _I, _L, _I, _P],
    "adv_conv2d_3x3_prep_weights_f32": [_P, _P, _I, _I, _I, _P],
    "adv_conv2d_3x3_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv2d_wino_prep_weights_f32": [_P, _P, _I, _I, _I, _P],
    "adv_conv2d_wino_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv3d_wino_prep_weights_f32": [_P, _P, _I, _I, _I, _P],
    "adv_conv3d_wino_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv2d_wino4_prep_weights_f32": [_P, _P, _I, _I, _I, _P],
    "adv_conv2d_wino4_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv2d_wino4_ksplit_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adv_conv3d_wino4_prep_weights_f32": [_P, _P, _I, _I, _I, _P],
    "adv_conv3d_wino4_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
}
_OTHER = {
    "adv_abi_version": ([], _I),
    "adv_build_has_test_hooks": ([], _I),
    "adv_roi_align_bwd_workspace_ints": ([_I, _I, _I, _I, _I, _I, _I], ctypes.c_int64),
    "adv_roi_align_bwd_segments": ([_I], _I),
    "adv_roi_gout_channel_last_floats": ([_I, _I, _I, _I], ctypes.c_int64),
    "adv_masked_loss_workspace_floats": ([], ctypes.c_int64),
    "adv_grid_sample3d_plan_bytes": ([_I, _I, _I, _I, _I, _I, _I], ctypes.c_int64),
    "adv_grid_sample3d_bwd_workspace_floats": ([_I, _I, _I, _I, _I], ctypes.c_int64),
    "adv_conv2d_1x1_prep_floats": ([_I, _I, _I], ctypes.c_int64),
    "adv_conv2d_3x3_prep_floats": ([_I, _I, _I], ctypes.c_int64),
    "adv_conv2d_wino_prep_floats": ([_I, _I, _I], ctypes.c_int64),
    "adv_conv3d_wino_prep_floats": ([_I, _I, _I], ctypes.c_int64),
    "adv_conv2d_wino4_prep_floats": ([_I, _I, _I], ctypes.c_int64),
    "adv_conv3d_wino4_prep_floats": ([_I, _I, _I], ctypes.c_int64),
    "adv_conv2d_wino4_ksplit_pick": ([_I, _I, _I, _I, _I], _I),
    "adv_conv2d_wino4_ksplit_chunk": ([_I, _I, _I], _I),
    "adv_last_hip_error": ([], _I),
    "adv_strerror": ([_I], ctypes.c_char_p),
    "adv_space_dsgn": ([_SP], None),
    "adv_space_srcnn": ([_SP], None),
    "adv_space_dsgn_gpu_reference": ([_SP], None),
}
EXPORTED = sorted(list(SIGNATURES) + list(_OTHER))

_lib = None


def _open(path):
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing at %s - build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C eval_driving_safety_amd/csrc` (needs hipcc, targets gfx950). "
            "This package has no CPU / PyTorch fallback." % (os.path.basename(path), path))
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, (argtypes, restype) in _OTHER.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.adv_abi_version()
    if got != ABI_VERSION:
        raise ImportError("%s ABI %d, binding expects %d - rebuild" % (os.path.basename(path), got, ABI_VERSION))
    return lib


def load():
    """Open libadvengine.so (once).  Raises if it has not been built - there is no fallback."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


class using:
    """``with _lib.using(_lib.HOOKS_LIB_PATH): ...`` - route every call inside the block to another build of the library (the
    -DADV_TEST_HOOKS one, whose ADV_* switches the A/B tests and tools/ flip).  Test / tuning infrastructure: nothing in the
    package enters it, and the shipped library reads no environment variable (adv_build_has_test_hooks() == 0)."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        load()
        self.prev, _lib = _lib, _open(self.path)
        return _lib

    def __exit__(self, *a):
        global _lib
        _lib = self.prev


def check(name, code):
    if code != ADV_OK:
        lib = load()
        detail = lib.adv_strerror(code).decode()
        if code == ADV_ELAUNCH:
            detail += " [hipError_t %d]" % lib.adv_last_hip_error()
        raise AdvEngineError(name, code, detail)


def call(name, *args):
    check(name, getattr(load(), name)(*args))
