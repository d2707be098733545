"""ctypes binding of libssdk.so (the HIP/gfx950 implementation behind ``include/ssdk.h``).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.  Tensors are
handed over as raw device pointers (``Tensor.data_ptr()``) on torch's current HIP stream; torch only owns the
memory and the stream.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SSDK_LIB') or os.path.join(_HERE, 'libssdk.so')   # SSDK_LIB: A/B builds of the same ABI
CSRC = os.path.join(_HERE, 'csrc')

_lib = None


class SsdkError(RuntimeError):
    pass


def build(jobs=8, force=False):
    """Compile every HIP source for gfx950 into ``libssdk.so`` (in-tree; hipcc cross-compiles without a GPU)."""
    args = ['make', '-C', CSRC, f'-j{jobs}']
    if force:
        args.append('-B')
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_SIGNATURES = {
    # name: (restype, argtypes)
    'ssdk_version': (C.c_int, []),
    'ssdk_last_error_string': (C.c_char_p, []),
    'ssdk_set_deterministic': (C.c_int, [C.c_int]),
    'ssdk_get_deterministic': (C.c_int, []),
    'ssdk_anchor_sizes_ssd': (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    'ssdk_anchor_sizes_retina': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int]),
    'ssdk_linspace_f32': (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_void_p]),
    'ssdk_anchors_level': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'ssdk_anchors_level_ex': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    'ssdk_encode_ground_truth_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'ssdk_encode_ground_truth': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p]),
    'ssdk_match_per_prediction_workspace_bytes': (C.c_size_t, [C.c_int]),
    'ssdk_match_per_prediction': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    'ssdk_multibox_loss_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'ssdk_hard_negative_mining': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                            C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_naive_sampler': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_multibox_loss_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_multibox_loss_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_multibox_loss_bwd_ex': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_encode_box': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                  C.c_int, C.c_void_p]),
    'ssdk_decode_box': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                  C.c_void_p]),
    'ssdk_heads_fwd_workspace_bytes': (C.c_size_t, []),
    'ssdk_heads_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_heads_fwd_ex': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    'ssdk_heads_fwd_fast_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'ssdk_heads_fwd_fast': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    'ssdk_conv2d_fwd_fast_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'ssdk_conv2d_fwd_fast': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_heads_fwd_timeouts': (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    'ssdk_streamk_poisoned': (C.c_int, []),
    'ssdk_debug_streamk_fault': (C.c_int, [C.c_int, C.c_uint]),
    'ssdk_streamk_reset': (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_box_to_corners': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    'ssdk_box_to_centroids': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    'ssdk_box_area': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    'ssdk_box_intersection': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_box_iou': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_nms_workspace_bytes': (C.c_size_t, [C.c_int]),
    'ssdk_nms': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_debug_heads_bwd_layout': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]),
    'ssdk_heads_bwd_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'ssdk_heads_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_heads_bwd_ex': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_heads_bwd_fast_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'ssdk_heads_bwd_fast': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_conv2d_bwd_fast_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'ssdk_conv2d_bwd_fast': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_conv2d_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'ssdk_conv2d_fwd_ws': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_conv2d_bwd_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'ssdk_conv2d_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_conv2d_bwd_sk': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_conv2d_transpose_weights': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_relu_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    'ssdk_batchnorm_workspace_bytes': (C.c_size_t, [C.c_int]),
    'ssdk_batchnorm_fwd': (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_batchnorm_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    'ssdk_batchnorm_stats': (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_batchnorm_apply': (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                       C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'ssdk_batchnorm_bwd_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p]),
    'ssdk_batchnorm_bwd_apply': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ssdk_batchnorm_fwd_chained': (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                             C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ssdk_batchnorm_bwd_chained': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ssdk_batchnorm_stats_accumulate': (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_batchnorm_apply_chained': (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                               C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ssdk_upsample_nearest_add_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p]),
    'ssdk_upsample_nearest_add_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p]),
    'ssdk_global_avgpool_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_global_avgpool_bwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_sigmoid_gate_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_sigmoid_gate_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'ssdk_sfam_pool_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ssdk_sfam_gate_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ssdk_sfam_gate_bwd_reduce': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ssdk_sfam_gate_bwd_apply': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ssdk_postprocess_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'ssdk_postprocess_workspace_bytes_ex': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'ssdk_postprocess': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ssdk_depthwise_conv2d_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p]),
    'ssdk_depthwise_conv2d_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'ssdk_mixup_images': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    'ssdk_mixup_ground_truth': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    'ssdk_mean_average_precision_workspace_bytes': (C.c_size_t, [C.c_longlong, C.c_longlong, C.c_int]),
    'ssdk_mean_average_precision': (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_int,
                                              C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}


class HeadLevel(C.Structure):
    """ssdk_head_level (include/ssdk.h)."""
    _fields_ = [('x', C.c_void_p), ('h', C.c_int), ('w', C.c_int), ('cin', C.c_int),
                ('w_score', C.c_void_p), ('b_score', C.c_void_p), ('n_score', C.c_int),
                ('w_loc', C.c_void_p), ('b_loc', C.c_void_p), ('n_loc', C.c_int),
                ('scores_offset', C.c_longlong), ('locs_offset', C.c_longlong),
                ('dx', C.c_void_p), ('dw_score', C.c_void_p), ('db_score', C.c_void_p), ('dw_loc', C.c_void_p),
                ('db_loc', C.c_void_p)]


class LossParams(C.Structure):
    """ssdk_loss_params (include/ssdk.h)."""
    _fields_ = [('cls_kind', C.c_int), ('loc_kind', C.c_int), ('focal_gamma', C.c_float), ('focal_alpha', C.c_float),
                ('reduce_mean', C.c_int), ('soft_epsilon', C.c_float), ('classification_weight', C.c_float),
                ('localization_weight', C.c_float), ('xy_scale', C.c_float), ('wh_scale', C.c_float), ('eps', C.c_float),
                ('smooth_l1_beta', C.c_float)]


class ConvDesc(C.Structure):
    """ssdk_conv_desc (include/ssdk.h)."""
    _fields_ = [('x', C.c_void_p), ('hin', C.c_int), ('win', C.c_int), ('cin', C.c_int), ('w', C.c_void_p),
                ('bias', C.c_void_p), ('cout', C.c_int), ('ksize', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('relu', C.c_int), ('y', C.c_void_p), ('dy', C.c_void_p), ('dx', C.c_void_p), ('dw', C.c_void_p),
                ('db', C.c_void_p), ('w_t', C.c_void_p), ('stats', C.c_void_p)]


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """The loaded library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SsdkError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                            f'(or `make -C {CSRC}`); there is no fallback path')
        _preload_torch_hip_runtime()
        _lib = _Bound(C.CDLL(LIB_PATH))
    return _lib


def _preload_torch_hip_runtime():
    """libssdk must share torch's HIP runtime instance (device pointers and streams come from torch).  torch
    bundles its own libamdhip64.so.7; loading it first makes the dynamic linker resolve libssdk's NEEDED entry of
    the same SONAME to that instance instead of a second runtime from /opt/rocm."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


class _Bound(object):
    """Attribute access binds the prototype from _SIGNATURES on first use; unknown or missing symbols raise."""

    def __init__(self, handle):
        self._handle = handle

    def __getattr__(self, name):
        if name not in _SIGNATURES:
            raise AttributeError(f'{name} is not part of the libssdk C ABI (include/ssdk.h)')
        try:
            fn = getattr(self._handle, name)
        except AttributeError:
            raise SsdkError(f'{LIB_PATH} does not export {name}: stale build, rebuild with `make -C {CSRC}`')
        fn.restype, fn.argtypes = _SIGNATURES[name]
        setattr(self, name, fn)
        return fn


def check(status, what):
    if status != 0:
        msg = lib().ssdk_last_error_string().decode('utf-8', 'replace')
        if status < 0:
            raise ValueError(f'{what}: {msg} (status {status})')
        raise SsdkError(f'{what}: {msg} (hipError {status})')


# Opt-in reduced-precision mode of the forward GEMMs (detection.modules.heads.set_fast_mode; SSDK_FAST_MODE): None = exact fp32,
# 'bf16x3' = split-bf16 operands on the bf16 matrix cores.  Read by the head GEMM (heads.py) and by ops.conv2d.
fast_mode = os.environ.get('SSDK_FAST_MODE') or None


def raw_stream(device=None):
    """The current HIP stream of ``device`` (default: the current device) as an integer handle.  (``torch.cuda.current_stream()`` builds
    a Stream object behind three Python calls -- 8 us, and a training step asks ~40 times; the raw getter is a single C call.)"""
    import torch
    if device is None:
        index = torch.cuda.current_device()
    else:
        index = torch.device(device).index
        if index is None:
            index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(index)


def current_stream():
    return C.c_void_p(raw_stream())


def ptr(t):
    """Device (or host) pointer of a contiguous tensor, or NULL for None."""
    if t is None:
        return None
    import torch
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    assert dense, 'libssdk takes dense buffers (row-major, or channels_last for 4-d maps / weights)'
    return C.c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SsdkError('libssdk kernels run on the GPU only; got a tensor on ' + str(t.device) +
                            ' (there is no CPU fallback)')


_scratch = {}


def scratch(nbytes, device, tag, zeroed=False):
    """A grow-only uint8 device buffer per (device, tag) for workspaces that live only for the duration of one library call.
    Calls on one stream are ordered, so the next call may reuse the bytes; nothing that a backward pass reads later may live here.
    (A torch.empty per call is cheap, but it makes the step allocate -- which is what broke HIP-graph capture of the step.)
    ``zeroed``: zero-filled when it is created (workspaces whose state the library keeps from call to call, e.g. the ready flags of
    ssdk_heads_fwd's stream-K form)."""
    import torch
    key = (torch.device(device), tag, raw_stream(device))
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = (torch.zeros if zeroed else torch.empty)((max(int(nbytes), 256),), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


STREAMK_TAG = 'heads_fwd_streamk'


def streamk_poisoned():
    """True once a stream-K launch of this process has given up on a parked partial tile (the sticky pinned-host word; no
    synchronisation): every later head GEMM on that workspace, eager or replayed from a HIP graph, stores NaN."""
    return bool(lib().ssdk_streamk_poisoned())


def streamk_reset():
    """Recovery after a stream-K timeout: clears every ssdk_heads_fwd workspace this process made and the sticky host word
    (``ssdk_streamk_reset``; synchronises their streams).  The step that timed out is lost; later steps and replays are healthy again."""
    import torch
    for (dev, tag, stream), buf in list(_scratch.items()):
        if tag != STREAMK_TAG:
            continue
        with torch.cuda.device(dev):
            check(lib().ssdk_streamk_reset(C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(stream)), 'ssdk_streamk_reset')


def streamk_timeouts():
    """Stream-K fix-up waits that ran out, summed over every ssdk_heads_fwd workspace this process made (synchronises their streams).
    0 is the only healthy value: a non-zero count means an output tile was filled with NaN and ssdk_heads_fwd now refuses to run."""
    import torch
    total = 0
    word = C.c_uint(0)
    for (dev, tag, stream), buf in list(_scratch.items()):
        if tag != STREAMK_TAG:
            continue
        with torch.cuda.device(dev):
            check(lib().ssdk_heads_fwd_timeouts(C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(stream), C.byref(word)), 'ssdk_heads_fwd_timeouts')
        total += int(word.value)
    return total
