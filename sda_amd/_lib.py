"""ctypes binding of libsda_hip.so (the C ABI declared in include/sda_hip.h).

The product path has NO fallback: if the library is missing or a symbol does not resolve,
loading raises -- nothing in sda_amd computes on the CPU or through the oracle.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import torch  # noqa: F401  -- must be imported first: the library binds to the HIP runtime torch loaded

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SDA_HIP_LIB') or os.path.join(HERE, 'lib', 'libsda_hip.so')     # (SDA_HIP_LIB: a tooling build, tools/ only)

ACT_IDS = {None: 0, 'none': 0, 'SiLU': 1, 'ReLU': 2, 'ELU': 3, 'GELU': 4, 'SELU': 5}

c_fp = c_void_p  # device pointers travel as integers


class ConvDesc(Structure):
    """Mirror of `struct sda_conv_desc` (include/sda_hip.h) -- field order and types must match."""
    _fields_ = [
        ('x', c_fp),
        ('x_sn_outer', c_int64), ('x_sn_inner', c_int64),
        ('n_inner', c_int32),
        ('x_n_off', c_int32),
        ('x_sc', c_int64), ('x_sy', c_int64), ('x_sx', c_int64),
        ('cx', c_int32),
        ('ctx', c_fp),
        ('ctx_sn', c_int64),
        ('cctx', c_int32),
        ('n', c_int32),
        ('hs', c_int32), ('ws', c_int32),
        ('up_h', c_int32), ('up_w', c_int32),
        ('zins_h', c_int32), ('zins_w', c_int32),
        ('mod', c_fp),
        ('mod_sn', c_int64),
        ('ln_mean', c_fp),
        ('ln_rstd', c_fp),
        ('act_in', c_int32),
        ('kh', c_int32), ('kw', c_int32), ('stride_h', c_int32), ('stride_w', c_int32), ('circular', c_int32),
        ('w', c_fp),
        ('cin_pad', c_int32), ('cout_pad', c_int32),
        ('bias', c_fp),
        ('out', c_fp),
        ('cout', c_int32), ('ho', c_int32), ('wo', c_int32),
        ('dact_z', c_fp),
        ('act_d', c_int32),
        ('res', c_fp),
        ('mt', c_int32),
        ('w_wino', c_fp),
        ('explicit_pad', c_int32), ('pad_h', c_int32), ('pad_w', c_int32),
        ('out_sn', c_int64), ('out_sc', c_int64), ('out_sy', c_int64), ('out_sx', c_int64),
        ('w_wino4', c_fp),
        ('pool_h', ctypes.c_int32), ('pool_w', ctypes.c_int32),
        ('w_wino4_zp', c_fp),
        ('w_h2', ctypes.c_void_p), ('w_h2_scale', ctypes.c_float), ('x_amax_static', ctypes.c_float),
        ('x_amax', c_fp), ('out_amax', c_fp),
    ]


class Conv3dDesc(Structure):
    """Mirror of `struct sda_conv3d_desc` (include/sda_hip.h)."""
    _fields_ = [
        ('x', c_fp), ('w', c_fp), ('bias', c_fp), ('z', c_fp), ('res', c_fp), ('out', c_fp),
        ('n', c_int32), ('cin', c_int32), ('cout', c_int32),
        ('in_size', c_int32 * 3), ('out_size', c_int32 * 3),
        ('k', c_int32 * 3), ('pad', c_int32 * 3), ('stride', c_int32 * 3), ('up', c_int32 * 3), ('dil', c_int32 * 3),
        ('circular', c_int32), ('act', c_int32), ('act_in', c_int32),
    ]


# name -> (restype, argtypes); exactly the symbols include/sda_hip.h declares
class Block1dDesc(Structure):
    """Mirror of `struct sda_block1d_desc` (include/sda_hip.h)."""
    _fields_ = [
        ('n', c_int32), ('c', c_int32), ('len', c_int32),
        ('circular', c_int32), ('act', c_int32), ('unbiased', c_int32),
        ('eps', c_float),
        ('k_pad', c_int32), ('m_pad', c_int32),
        ('a', c_fp),
        ('mod', c_fp),
        ('mod_sn', c_int64),
        ('w1', c_fp), ('b1', c_fp),
        ('w2', c_fp), ('b2', c_fp),
        ('z', c_fp),
        ('mean', c_fp), ('rstd', c_fp),
        ('y', c_fp),
        ('g', c_fp),
        ('gx', c_fp),
    ]


NET1D_MAXB = 8


class Net1dDesc(Structure):
    """Mirror of `struct sda_net1d_desc` (include/sda_hip.h)."""
    _fields_ = [
        ('n', c_int32), ('len', c_int32),
        ('cin', c_int32), ('c', c_int32), ('cout', c_int32),
        ('nblocks', c_int32),
        ('circular', c_int32), ('act', c_int32), ('unbiased', c_int32),
        ('eps', c_float),
        ('x', c_fp), ('x_sn', c_int64), ('x_sc', c_int64), ('x_sx', c_int64),
        ('out', c_fp), ('out_sn', c_int64), ('out_sc', c_int64), ('out_sx', c_int64),
        ('w', c_fp),
        ('bias', c_fp),
        ('mod', c_fp * NET1D_MAXB),
        ('mod_sn', c_int64),
        ('a_save', c_fp), ('z_save', c_fp), ('save_stride', c_int64),
        ('mean_save', c_fp), ('rstd_save', c_fp), ('stat_stride', c_int64),
    ]


MLP_MAXG = 32


class MlpDesc(Structure):
    """Mirror of `struct sda_mlp_desc` (include/sda_hip.h)."""
    _fields_ = [
        ('rows', c_int32), ('ngemm', c_int32),
        ('act', c_int32), ('unbiased', c_int32),
        ('eps', c_float),
        ('kind', c_int32 * MLP_MAXG),
        ('in_f', c_int32 * MLP_MAXG), ('out_f', c_int32 * MLP_MAXG),
        ('w_off', c_int32 * MLP_MAXG), ('b_off', c_int32 * MLP_MAXG),
        ('w', c_fp), ('bias', c_fp),
        ('x', c_fp), ('x_ld', c_int64),
        ('out', c_fp), ('out_ld', c_int64),
        ('a_save', c_fp), ('z_save', c_fp), ('save_stride', c_int64), ('save_ld', c_int32),
        ('mean_save', c_fp), ('rstd_save', c_fp), ('stat_stride', c_int64),
    ]


class MlpWin(Structure):
    """Mirror of `struct sda_mlp_win` (include/sda_hip.h)."""
    _fields_ = [
        ('nw', c_int32), ('len', c_int32), ('c', c_int32), ('emb_n', c_int32),
        ('x', c_fp), ('emb', c_fp),
        ('cx0', c_float), ('cx1', c_float), ('cn', c_float),
        ('coef', c_fp),
        ('y', c_fp), ('y_sn', c_int64),
        ('p_start', c_int32), ('p_step', c_int32), ('p_stop', c_int32),
        ('c_start', c_int32), ('c_step', c_int32), ('c_stop', c_int32),
        ('std', c_float), ('gamma', c_float),
        ('eps', c_fp), ('ghat', c_fp),
        ('gwin', c_fp),
    ]


class Net1dFuse(Structure):
    """Mirror of `struct sda_net1d_fuse` (include/sda_hip.h)."""
    _fields_ = [
        ('cx0', c_float), ('cx1', c_float), ('cn', c_float),
        ('coef', c_fp),
        ('y', c_fp), ('y_sn', c_int64),
        ('p_start', c_int32), ('p_step', c_int32), ('p_stop', c_int32),
        ('c_start', c_int32), ('c_step', c_int32), ('c_stop', c_int32),
        ('std', c_float), ('gamma', c_float),
        ('ghat', c_fp),
        ('eps', c_fp),
        ('mode', c_int32),
        ('xs', c_fp),
        ('step_coef', c_fp),
        ('partial', c_fp), ('partial_stride', c_int32),
    ]


SIGNATURES = {
    'sda_abi_version': (c_int, []),
    'sda_conv_igemm': (c_int, [POINTER(ConvDesc), c_void_p]),
    'sda_gauss_cotangent': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'sda_block1d_fwd': (c_int, [POINTER(Block1dDesc), c_void_p]),
    'sda_block1d_bwd': (c_int, [POINTER(Block1dDesc), c_void_p]),
    'sda_net1d_fwd': (c_int, [POINTER(Net1dDesc), c_void_p]),
    'sda_net1d_bwd': (c_int, [POINTER(Net1dDesc), c_void_p]),
    'sda_net1d_fwd_fused': (c_int, [POINTER(Net1dDesc), POINTER(Net1dFuse), c_void_p]),
    'sda_net1d_bwd_fused': (c_int, [POINTER(Net1dDesc), POINTER(Net1dFuse), c_void_p]),
    'sda_net1d_tiles': (c_int, [POINTER(Net1dDesc)]),
    'sda_step1d_prologue': (c_int, [c_fp, c_int, c_fp, c_fp, c_int, c_int, c_float, c_float, c_int, c_fp, c_int, c_fp, c_fp, c_int,
                                    c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_void_p]),
    'sda_pc_correct_keyed': (c_int, [c_fp, c_fp, c_int, c_int64, c_fp, c_int, c_float, c_float, c_fp, c_uint64, c_int64, c_fp, c_int64,
                                     c_int64, c_void_p]),
    'sda_conv_parity4': (c_int, [POINTER(ConvDesc), c_void_p]),
    'sda_conv_igemm_path': (c_int, [POINTER(ConvDesc)]),
    'sda_conv_igemm_lds_bytes': (c_int64, [POINTER(ConvDesc)]),
    'sda_pack_conv_weight': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_int, c_int, c_void_p]),
    'sda_pack_conv_weight_wino': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_int, c_void_p]),
    'sda_pack_conv_weight_wino4': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_int, c_void_p]),
    'sda_pack_conv_weight_wino4_zp': (c_int, [c_fp, c_int, c_int, c_fp, c_void_p]),
    'sda_wino4_zp_floats': (c_int64, [c_int, c_int]),
    'sda_ln_stats': (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_int64, c_float, c_int, c_fp, c_fp, c_void_p]),
    'sda_ln_apply': (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_int64, c_fp, c_fp, c_fp, c_void_p]),
    'sda_ln_bwd': (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int64, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp,
                           c_void_p]),
    'sda_ln_bwd_amax': (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int64, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp,
                           c_fp, c_void_p]),
    'sda_time_embed': (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_void_p]),
    'sda_linear_small': (c_int, [c_fp, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_void_p]),
    'sda_linear': (c_int, [c_fp, c_int, c_int, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_void_p]),
    'sda_mlp_fwd': (c_int, [POINTER(MlpDesc), c_void_p]),
    'sda_mlp_bwd': (c_int, [POINTER(MlpDesc), c_void_p]),
    'sda_mlp_slab_floats': (c_int, [c_int, c_int]),
    'sda_mlp_fwd_win': (c_int, [POINTER(MlpDesc), POINTER(MlpWin), c_void_p]),
    'sda_mlp_bwd_win': (c_int, [POINTER(MlpDesc), POINTER(MlpWin), c_void_p]),
    'sda_mc_finish': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_float, c_float, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_void_p]),
    'sda_row_ln': (c_int, [c_fp, c_int, c_int, c_float, c_int, c_fp, c_fp, c_fp, c_void_p]),
    'sda_row_ln_bwd': (c_int, [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_void_p]),
    'sda_obs_subsample': (c_int, [c_fp, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_fp, c_void_p]),
    'sda_obs_subsample_adjoint': (c_int, [c_fp, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_fp, c_void_p]),
    'sda_obs_pointwise': (c_int, [c_fp, c_int64, c_int, c_fp, c_void_p]),
    'sda_obs_pointwise_vjp': (c_int, [c_fp, c_fp, c_int64, c_int, c_fp, c_void_p]),
    'sda_obs_mask': (c_int, [c_fp, c_int64, c_fp, c_int64, c_fp, c_void_p]),
    'sda_obs_timediff': (c_int, [c_fp, c_int64, c_int, c_int64, c_int, c_int, c_fp, c_void_p]),
    'sda_obs_timediff_adjoint': (c_int, [c_fp, c_int64, c_int, c_int64, c_int, c_int, c_fp, c_void_p]),
    'sda_obs_subsample_guidance': (c_int, [c_fp, c_fp, c_fp, c_int64, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                           c_float, c_float, c_float, c_float, c_fp, c_fp, c_void_p]),
    'sda_obs_coarsen': (c_int, [c_fp, c_int64, c_int, c_int, c_int, c_fp, c_void_p]),
    'sda_obs_coarsen_adjoint': (c_int, [c_fp, c_int64, c_int, c_int, c_int, c_fp, c_void_p]),
    'sda_obs_vorticity': (c_int, [c_fp, c_int64, c_int, c_int, c_fp, c_void_p]),
    'sda_obs_vorticity_adjoint': (c_int, [c_fp, c_int64, c_int, c_int, c_fp, c_void_p]),
    'sda_fold': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_void_p]),
    'sda_fold_adjoint': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_void_p]),
    'sda_unfold_adjoint': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_int64, c_fp, c_void_p]),
    'sda_vp_schedule': (c_int, [c_fp, c_int, c_float, c_float, c_int, c_fp, c_void_p]),
    'sda_pc_predict': (c_int, [c_fp, c_fp, c_int64, c_float, c_float, c_fp, c_void_p]),
    'sda_sumsq_partial': (c_int, [c_fp, c_int, c_int64, c_fp, c_int, c_void_p]),
    'sda_pc_correct': (c_int, [c_fp, c_fp, c_fp, c_int, c_int64, c_fp, c_int, c_float, c_float, c_fp, c_void_p]),
    'sda_randn_rows': (c_int, [c_fp, c_int, c_int64, c_uint64, c_int64, c_int64, c_fp, c_int64, c_int64, c_void_p]),
    'sda_clock_probe': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'sda_conv_h2': (c_int, [POINTER(ConvDesc), c_void_p]),
    'sda_conv_h2_supported': (c_int, [POINTER(ConvDesc)]),
    'sda_pack_conv_weight_h2': (c_int, [c_fp, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'sda_conv_h2_packed_bytes': (c_int64, [c_int, c_int, c_int]),
    'sda_pack_conv_weight_h2_up': (c_int, [c_fp, c_int, c_int, c_float, c_void_p, c_void_p]),
    'sda_conv_h2_up_packed_bytes': (c_int64, [c_int, c_int]),
    'sda_pack_conv_weight_h2_rows': (c_int, [c_fp, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'sda_conv_h2_rows_packed_bytes': (c_int64, [c_int, c_int, c_int]),
    'sda_conv_h2_scale': (c_float, [c_float]),
    'sda_absmax': (c_int, [c_fp, c_int64, c_fp, c_void_p]),
    'sda_philox_words': (c_int, [c_fp, c_int64, c_uint64, c_uint32, c_uint32, c_uint32, c_void_p]),
    'sda_denoise': (c_int, [c_fp, c_fp, c_int64, c_float, c_float, c_fp, c_fp, c_void_p]),
    'sda_guided_combine': (c_int, [c_fp, c_fp, c_fp, c_int64, c_float, c_float, c_fp, c_fp, c_void_p]),
    'sda_pairwise_dist': (c_int, [c_fp, c_int, c_fp, c_int, c_int64, c_int, c_fp, c_void_p]),
    'sda_mmd_kernel_sums': (c_int, [c_fp, c_int64, c_void_p, c_int, c_void_p]),
    'sda_assignment_cost': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'sda_transport_cost': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'sda_conv3d': (c_int, [POINTER(Conv3dDesc), c_void_p]),
    'sda_conv3d_packed_floats': (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'sda_pack_conv3d_weight': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_void_p]),
    'sda_pool3d_sum': (c_int, [c_fp, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_void_p]),
}

_lib = None


class SdaHipError(RuntimeError):
    pass


def load():
    """Load libsda_hip.so and bind every declared symbol.  Raises (never falls back) if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdaHipError(
            f'{LIB_PATH} not found: the HIP extension is not built. Run `python -m sda_amd.build` '
            f'(or __graft_entry__.build()). sda_amd has no CPU fallback.')
    if os.environ.get('SDA_HIP_LIB'):
        import warnings
        warnings.warn(f'sda_amd: kernel library overridden by SDA_HIP_LIB={LIB_PATH} (a tooling build; bench.py records it as '
                      f'`kernel_library_override`)')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SdaHipError(f'libsda_hip.so does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        if rc < 0:
            msg = {-1: 'bad argument', -2: 'unsupported shape', -3: 'tile exceeds LDS'}.get(rc, 'error')
            raise SdaHipError(f'{what}: {msg} (SDA_E {rc})')
        raise SdaHipError(f'{what}: HIP launch error {rc}')
