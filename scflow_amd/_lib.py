"""ctypes binding of libscflow_hip.so (C ABI declared in include/scflow_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared
object has not been built, ``load()`` raises with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libscflow_hip.so')

SCF_OK = 0
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
CONV_PLAIN, CONV_GRU_ZR, CONV_GRU_Q = 0, 1, 2
MAX_LEVELS = 12
ABI_MAJOR = 5            # SCF_ABI_MAJOR of include/scflow_hip.h this binding was written against

_fp = C.c_void_p  # device pointers travel as integers


class ConvDesc(C.Structure):
    """mirror of ``scf_conv_desc`` (include/scflow_hip.h)."""
    _fields_ = [
        ('in0', _fp), ('in1', _fp),
        ('C0', C.c_int32), ('C1', C.c_int32),
        ('in0_nstride', C.c_int64), ('in1_nstride', C.c_int64),
        ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
        ('wp', _fp), ('w_nstride', C.c_int64),
        ('Mld', C.c_int32), ('Cout', C.c_int32),
        ('KH', C.c_int32), ('KW', C.c_int32), ('stride', C.c_int32),
        ('pad_h', C.c_int32), ('pad_w', C.c_int32), ('KC', C.c_int32),
        ('out', _fp), ('out_nstride', C.c_int64),
        ('bias', _fp), ('scale', _fp), ('shift', _fp),
        ('res', _fp), ('res_nstride', C.c_int64),
        ('out_div', C.c_float),
        ('act', C.c_int32), ('act2', C.c_int32), ('act_split', C.c_int32),
        ('mode', C.c_int32),
        ('gru_h', _fp), ('gru_h_nstride', C.c_int64),
        ('gru_aux', _fp), ('gru_aux_nstride', C.c_int64),
        ('gru_z', _fp), ('gru_z_nstride', C.c_int64),
        ('wp_f16', _fp),
        ('wp_a4', _fp),
        ('a4_groups', C.c_int32),
        ('a4_mld', C.c_int32),
        ('wp_a4s', _fp),
        ('a4s_groups', C.c_int32),
        ('wp_thin', _fp),
        ('out_tile8x4', C.c_int32),
        ('wp_taps', _fp),
        ('wp_a4t', _fp),
        ('a4t_groups', C.c_int32),
        ('wp_wino1d', _fp),
        ('wp_wino', _fp),
        ('wp_wino1d4', _fp),
        ('k_slices', C.c_int32),
        ('out_slice_stride', C.c_int64),
    ]


class GruPass(C.Structure):
    """mirror of ``scf_gru_pass`` (include/scflow_hip.h)."""
    _fields_ = [
        ('KH', C.c_int32), ('KW', C.c_int32), ('pad_h', C.c_int32), ('pad_w', C.c_int32),
        ('wp_zr', _fp), ('bias_zr', _fp), ('wp_q', _fp), ('bias_q', _fp),
        ('wp_zr_a4', _fp), ('wp_q_a4', _fp), ('a4_groups', C.c_int32),
        ('wp_zr_f16', _fp), ('wp_q_f16', _fp),
        ('wp_zr_k32', _fp), ('wp_q_k32', _fp),
        ('wp_zr_a4s', _fp), ('wp_q_a4s', _fp), ('a4s_groups', C.c_int32),
        ('wp_zr_a4t', _fp), ('wp_q_a4t', _fp), ('a4t_groups', C.c_int32),
        ('wp_zr_wino1d', _fp), ('wp_q_wino1d', _fp),
        ('wp_zr_wino1d4', _fp), ('wp_q_wino1d4', _fp),
    ]


class ConvLogEntry(C.Structure):
    """mirror of ``scf_conv_log_entry`` (include/scflow_hip_prof.h)."""
    _fields_ = [(n, C.c_int32) for n in ('kernel', 'Cin', 'Cout', 'KH', 'KW', 'stride', 'Ho', 'Wo', 'N', 'mode')]


KERNEL_NAMES = {1: 'thin', 2: 'taps', 3: 'winograd', 4: 'winograd F(2,5)', 5: 'f16x3', 6: 'direct-dma',
                7: 'direct-mfma', 8: 'direct-mfma-ksplit', 9: 'winograd-q', 10: 'winograd F(4,5)'}


class FcDesc(C.Structure):
    """mirror of ``scf_fc_desc`` (include/scflow_hip.h)."""
    _fields_ = [('x', _fp), ('x_parts', C.c_int32), ('x_part_stride', C.c_int64),
                ('x_bias', _fp), ('x_relu', C.c_int32),
                ('gn_groups', C.c_int32), ('gn_hw', C.c_int32), ('gn_gamma', _fp), ('gn_beta', _fp), ('gn_eps', C.c_float),
                ('N', C.c_int32), ('K', C.c_int32),
                ('W', _fp), ('bias', _fp), ('y', _fp), ('O', C.c_int32),
                ('W2', _fp), ('bias2', _fp), ('y2', _fp), ('O2', C.c_int32),
                ('act', C.c_int32), ('slices', C.c_int32)]


class IterGN(C.Structure):
    """mirror of ``scf_iter_gn``."""
    _fields_ = [('gamma', _fp), ('beta', _fp), ('out', _fp),
                ('C', C.c_int32), ('HW', C.c_int32), ('G', C.c_int32), ('eps', C.c_float)]


class ScflowIter(C.Structure):
    """mirror of ``scf_scflow_iter`` (include/scflow_hip.h): one refinement iteration."""
    _fields_ = [
        ('struct_size', C.c_int32),
        ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
        ('L', C.c_int32), ('radius', C.c_int32), ('tiled_levels', C.c_uint32), ('corr_channels', C.c_int32),
        ('levels', _fp * MAX_LEVELS),
        ('flow_lr', _fp), ('corr', _fp),
        ('mask_flow', C.c_int32), ('mask_corr', C.c_int32), ('mask_prev', _fp), ('flow_masked', _fp),
        ('flow0', ConvDesc), ('flow1', ConvDesc), ('corr0', ConvDesc), ('corr1', ConvDesc), ('outn', ConvDesc),
        ('flow_copy_dst', _fp),
        ('hx', _fp), ('hx_nstride', C.c_int64), ('Ch', C.c_int32), ('Cc', C.c_int32), ('Cx', C.c_int32),
        ('npass', C.c_int32),
        ('gru', GruPass * 2), ('ctx', _fp * 2), ('ctx_nstride', C.c_int64), ('z', _fp), ('rh', _fp),
        ('heads', ConvDesc), ('fpred', ConvDesc), ('mpred', ConvDesc), ('menc0', ConvDesc), ('menc1', ConvDesc),
        ('denc0', ConvDesc), ('denc1', ConvDesc),
        ('pose', ConvDesc * 3), ('gn', IterGN * 3),
        ('fc1_w', _fp), ('fc1_b', _fp), ('fc1_out', _fp), ('fc1_K', C.c_int32), ('fc1_O', C.c_int32),
        ('fc2_w', _fp), ('fc2_b', _fp), ('fc2_out', _fp), ('fc2_O', C.c_int32),
        ('fc_fused', C.c_int32), ('fc1_slices', C.c_int32), ('fc2_slices', C.c_int32),
        ('rot_w', _fp), ('rot_b', _fp), ('rot_all', _fp), ('rot_O', C.c_int32),
        ('trans_w', _fp), ('trans_b', _fp), ('trans_all', _fp), ('trans_O', C.c_int32),
        ('label', _fp), ('num_class', C.c_int32), ('label_mode', C.c_int32),
        ('depth', _fp), ('K', _fp), ('R0', _fp), ('t0', _fp), ('invalid_flow_num', C.c_float),
        ('flow_in', _fp), ('R_in', _fp), ('t_in', _fp),
        ('flow_out', _fp), ('flow_pred', _fp), ('mask_up', _fp),
        ('R_out', _fp), ('t_out', _fp), ('d_rot', _fp), ('d_trans', _fp),
        ('side_stream', _fp), ('overlap_flow', C.c_int32), ('overlap_mask', C.c_int32), ('overlap_up', C.c_int32),
        ('lookup_timer', _fp),
    ]


# name -> (restype, argtypes); every symbol include/scflow_hip.h declares
SIGNATURES = {
    'scf_version': (C.c_int, []),
    'scf_error_string': (C.c_char_p, [C.c_int]),
    'scf_device_count': (C.c_int, []),
    'scf_corr_build': (C.c_int, [_fp, _fp, C.POINTER(_fp), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, _fp]),
    'scf_corr_lookup': (C.c_int, [C.POINTER(_fp), _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, _fp]),
    'scf_corr_build_ex': (C.c_int, [_fp, _fp, C.POINTER(_fp), C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_uint, _fp]),
    'scf_corr_lookup_ex': (C.c_int, [C.POINTER(_fp), _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_uint, _fp]),
    'scf_corr_level_floats': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'scf_corr_preferred_layout': (C.c_uint, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'scf_pose_error': (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    'scf_filter_flow_by_mask': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _fp]),
    'scf_cal_epe_workspace_bytes': (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    'scf_cal_epe': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int,
                              _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    'scf_timer_create': (C.c_int, [C.POINTER(_fp)]),
    'scf_timer_destroy': (C.c_int, [_fp]),
    'scf_timer_arm': (C.c_int, [_fp]),
    'scf_timer_elapsed_us': (C.c_int, [_fp, C.POINTER(C.c_float)]),
    'scf_conv2d': (C.c_int, [C.POINTER(ConvDesc), _fp]),
    'scf_conv2d_pair': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _fp]),
    'scf_conv2d_query': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int32)]),
    'scf_tune': (C.c_int, [C.c_int, C.c_int]),
    'scf_conv_log_enable': (C.c_int, [C.c_int]),
    'scf_conv_log_read': (C.c_int, [C.POINTER(ConvLogEntry), C.c_int]),
    'scf_pack_conv_weight_size': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'scf_pack_conv_weight': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'scf_pack_conv_weight_a4_size': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'scf_pack_conv_weight_a4': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'scf_pack_conv_weight_taps_size': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'scf_pack_conv_weight_taps': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'scf_pack_conv_weight_wino1d_size': (C.c_int64, [C.c_int32, C.c_int32]),
    'scf_pack_conv_weight_wino1d': (C.c_int, [_fp, C.c_int32, C.c_int32, _fp]),
    'scf_pack_conv_weight_wino1d4_size': (C.c_int64, [C.c_int32, C.c_int32]),
    'scf_pack_conv_weight_wino1d4': (C.c_int, [_fp, C.c_int32, C.c_int32, _fp]),
    'scf_pack_conv_weight_wino_size': (C.c_int64, [C.c_int32, C.c_int32]),
    'scf_pack_conv_weight_wino': (C.c_int, [_fp, C.c_int32, C.c_int32, _fp]),
    'scf_sepconv_gru': (C.c_int, [_fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(GruPass), C.c_int, _fp, _fp, _fp]),
    'scf_sepconv_gru_ctx': (C.c_int, [_fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(GruPass), C.c_int, C.POINTER(_fp),
                                      C.c_int64, _fp, _fp, _fp]),
    'scf_scflow_iteration': (C.c_int, [C.POINTER(ScflowIter), _fp]),
    'scf_instance_norm': (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int, C.c_float, C.c_int, _fp]),
    'scf_group_norm_relu': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_float, _fp]),
    'scf_conv_workspace': (C.c_int, [_fp, _fp, C.c_int64]),
    'scf_group_norm_relu_parts': (C.c_int, [_fp, C.c_int, C.c_int64, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_float, _fp]),
    'scf_fc_splitk': (C.c_int, [C.POINTER(FcDesc), _fp]),
    'scf_linear': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'scf_linear_pair': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int,
                                  C.c_int, _fp]),
    'scf_pose_update': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp,
                                  C.c_int, _fp]),
    'scf_reproject_flow': (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int,
                                     C.c_int, C.c_float, _fp]),
    'scf_unproject_depth': (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'scf_resize_bilinear': (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, _fp]),
    'scf_convex_upsample': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, _fp]),
    'scf_avgpool2x2': (C.c_int, [_fp, _fp, C.c_int64, C.c_int, C.c_int, _fp]),
    'scf_mul_mask': (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int64, C.c_int, C.c_int, C.c_int, _fp]),
    'scf_copy_strided': (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int, C.c_int64, _fp]),
}

_lib = None


class ScflowHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the HIP library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScflowHipError(
            f'{LIB_PATH} is missing: the HIP kernels are the product and there is no fallback. '
            'Build it with `python scflow_amd/csrc/build.py` (hipcc --offload-arch=gfx950).')
    lib = C.CDLL(LIB_PATH)
    lib.scf_version.restype = C.c_int
    have = lib.scf_version()
    if have // 100 != ABI_MAJOR:
        raise ScflowHipError(f'{LIB_PATH} was built with ABI major {have // 100}, this binding expects '
                             f'{ABI_MAJOR}: rebuild it (python scflow_amd/csrc/build.py --force)')
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != SCF_OK:
        msg = load().scf_error_string(code).decode()
        raise ScflowHipError(f'{what} failed: {msg} ({code})')
