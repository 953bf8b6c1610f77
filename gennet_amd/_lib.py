"""ctypes binding of libgennet_hip.so (C ABI: include/gennet_hip.h).

The HIP library is the product path: if it is missing this module raises at import of the first op -- there is
no CPU / eager-PyTorch fallback anywhere in gennet_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libgennet_hip.so')

_lib = None

vp, i32, f32, f64, u64, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_uint64, C.c_size_t

_SIGS = {
    'gn_conv1d_fwd': [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    'gn_conv1d_fwd_bf16x3': [vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp],
    'gn_conv1d_fwd_wino': [vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    'gn_conv1d_fwd_stats': [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'gn_conv1d_fwd_dropout': [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp],
    'gn_conv1d_transpose_w': [vp, vp, i32, i32, i32, vp],
    'gn_conv1d_dgrad': [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'gn_conv1d_dgrad_fused': [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, f32, f32, vp],
    'gn_dense_bwd_fused': [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, f32, f32, vp],
    'gn_conv1d_wgrad': [vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'gn_conv2d_w2_fold': [vp, vp, vp, vp, i32, i32, i32, vp],
    'gn_conv2d_w2_unfold_grad': [vp, vp, vp, vp, i32, i32, i32, vp],
    'gn_conv1d_tapfold_x': [vp, vp, i32, i32, i32, i32, i32, vp],
    'gn_conv1d_tapunfold_dx': [vp, vp, i32, i32, i32, i32, i32, vp],
    'gn_conv1d_tapfold_w': [vp, vp, i32, i32, i32, vp],
    'gn_conv1d_tapunfold_dw': [vp, vp, i32, i32, i32, vp],
    'gn_conv1d_tap_groups': [i32, vp, vp],
    'gn_conv1d_up2_fold': [vp, vp, vp, vp, i32, i32, i32, vp],
    'gn_conv1d_up2_unfold_grad': [vp, vp, vp, vp, i32, i32, i32, vp],
    'gn_dense_fwd': [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    'gn_dense_bwd': [vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, vp],
    'gn_act_fwd': [vp, vp, sz, i32, f32, vp],
    'gn_act_bwd': [vp, vp, vp, sz, i32, f32, vp],
    'gn_act_dropout_bwd': [vp, vp, vp, vp, sz, i32, f32, f32, vp],
    'gn_set_conv_math': [i32, vp, sz],
    'gn_conv_fold_bn': [vp, vp, vp, vp, vp, vp, sz, i32, vp],
    'gn_bn_apply_dropgen': [vp, vp, vp, vp, vp, sz, i32, i32, f32, f32, u64, u64, vp],
    'gn_prelu_fwd': [vp, vp, vp, i32, sz, vp],
    'gn_prelu_bwd': [vp, vp, vp, vp, vp, i32, sz, vp],
    'gn_dropout_mask': [vp, sz, f32, u64, u64, vp],
    'gn_dropout_apply': [vp, vp, vp, sz, f32, vp],
    'gn_upsample2_fwd': [vp, vp, i32, i32, i32, vp],
    'gn_upsample2_bwd': [vp, vp, i32, i32, i32, vp],
    'gn_maxpool_h2_fwd': [vp, vp, i32, i32, i32, vp],
    'gn_maxpool_h2_bwd': [vp, vp, vp, i32, i32, i32, vp],
    'gn_subtract_stack_fwd': [vp, vp, vp, i32, i32, vp],
    'gn_subtract_stack_bwd': [vp, vp, i32, i32, vp],
    'gn_affine_stack_fwd': [vp, vp, vp, f32, f32, vp, i32, i32, vp],
    'gn_affine_stack_bwd': [vp, f32, f32, vp, i32, i32, vp],
    'gn_assemble_d_batch': [vp, vp, vp, vp, vp, i32, i32, vp],
    'gn_fill_uniform': [vp, sz, f32, f32, u64, u64, vp],
    'gn_fill_normal': [vp, sz, f32, f32, u64, u64, vp],
    'gn_gather_rows': [vp, vp, vp, i32, i32, vp],
    'gn_axpy': [vp, vp, f32, sz, vp],
    'gn_bn_stats': [vp, sz, i32, vp, vp, sz, vp],
    'gn_bn_finalize': [vp, f64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp],
    'gn_bn_finalize_zero_debias': [vp, f64, vp, vp, f32, f32, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp],
    'gn_bn_infer_coeffs': [vp, vp, vp, vp, f32, vp, vp, i32, vp],
    'gn_bn_apply': [vp, vp, vp, vp, vp, sz, i32, i32, f32, f32, vp],
    'gn_bn_bwd_stats': [vp, vp, vp, vp, vp, vp, vp, vp, sz, sz, i32, i32, f32, f32, vp, vp, vp],
    'gn_bn_bwd_apply': [vp, vp, vp, vp, vp, vp, vp, vp, f64, vp, vp, vp, vp, sz, i32, i32, f32, f32, vp, vp, vp],
    'gn_bn_bwd_stats_conv1': [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, sz, i32, i32, f32, f32, vp, vp, vp],
    'gn_bn_bwd_apply_conv1': [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, f64, vp, vp, vp, vp, sz, i32, i32, f32, f32, vp, vp, vp],
    'gn_bce_loss': [vp, vp, vp, vp, i32, i32, vp],
    'gn_mse_loss': [vp, vp, vp, vp, i32, i32, vp],
    'gn_adam_step': [vp, vp, vp, vp, sz, f32, f32, f32, f32, vp],
    'gn_set_rng_base': [vp],
    'gn_adam_step_dyn': [vp, vp, vp, vp, sz, vp, f32, f32, f32, vp],
    'gn_fill_normal_dyn': [vp, sz, f32, vp, u64, u64, vp],
    'gn_bn_finalize_zero_debias_dyn': [vp, f64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp],
    'gn_prof_enable': [i32],
    'gn_prof_reset': [],
    'gn_prof_collect': [i32, vp],
    'gn_chirp_fd_whitened': [vp, vp, vp, vp, vp, i32, i32, f64, f64, f64, f64, f64, vp],
    'gn_irfft_f64': [vp, vp, vp, i32, i32, vp],
    'gn_rfft_f64': [vp, vp, vp, i32, i32, vp],
    'gn_mul_f64': [vp, vp, sz, sz, i32, vp],
    'gn_align_crop': [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f64, f64, f64, vp],
    'gn_noise_fd': [vp, vp, i32, i32, u64, u64, vp],
    'gn_synth_templates': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f64, f64, f64, f64, f64, f64, f64, f64, vp],
    'gn_synth_templates_prior': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f64, f64, f64, f64, f64, f64, f64, f64, u64, u64, i32, i32,
                                 f64, f64, vp],
    'gn_noise_whitened': [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f64, u64, u64, vp],
    'gn_synth_templates_noise': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f64, f64, f64, f64, f64, f64, f64, f64,
                                 u64, u64, i32, i32, f64, f64, u64, u64, vp, vp],
    'gn_kde2d_pdf': [vp, i32, vp, i32, f64, f64, f64, f64, vp, vp],
    'gn_scale_f64': [vp, f64, sz, vp],
    'gn_f64_to_f32': [vp, vp, f64, sz, vp],
}
_SIZE_FNS = {
    'gn_conv1d_bf16x3_workspace': [i32, i32, i32, i32, i32],
    'gn_conv1d_wino_workspace': [i32, i32],
    'gn_conv1d_wgrad_workspace': [i32, i32, i32, i32, i32, i32, i32],
    'gn_dense_bwd_workspace': [i32, i32, i32],
    'gn_bn_stats_workspace': [sz, i32],
    'gn_conv1d_fwd_stats_workspace': [i32, i32, i32],
}


class GennetHipError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises (loudly) when the .so has not been built: run `python -m gennet_amd.build`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            try:                            # a fresh checkout on a box that has hipcc: compile the kernels (~30 s), nothing else
                from . import build as _build
                _build.build(verbose=False)
            except Exception as e:          # noqa: BLE001
                raise GennetHipError('%s not found and building it failed (%s): run `python -m gennet_amd.build` '
                                     '(hipcc --offload-arch=gfx950); gennet_amd has no CPU fallback' % (LIB_PATH, e))
        # torch ships its own libamdhip64; it must be the HIP runtime of the process (device pointers and streams come from
        # torch), so torch is imported BEFORE this library is dlopen-ed and the library's NEEDED libamdhip64 resolves to the
        # already-loaded copy.  Loading in the other order gives the process two HIP runtimes and launches fail.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for table, restype in ((_SIGS, i32), (_SIZE_FNS, sz)):
            for name, args in table.items():
                try:
                    fn = getattr(L, name)
                except AttributeError:      # reported by call()/size() and by tests/test_capi_symbols.py
                    continue
                fn.argtypes = args
                fn.restype = restype
        L.gn_last_error.restype = C.c_char_p
        L.gn_version.restype = i32
        _lib = L
    return _lib


def exported_symbols():
    return sorted(list(_SIGS) + list(_SIZE_FNS) + ['gn_last_error', 'gn_version'])


def call(name, *args):
    L = lib()
    try:
        fn = getattr(L, name)
    except AttributeError:
        raise GennetHipError('%s is not exported by %s (stale build?)' % (name, LIB_PATH))
    rc = fn(*args)
    if rc != 0:
        raise GennetHipError('%s failed (%d): %s' % (name, rc, L.gn_last_error().decode()))


def size(name, *args):
    return int(getattr(lib(), name)(*args))
