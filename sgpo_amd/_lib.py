"""ctypes binding of libsgpo_hip.so (include/sgpo_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call
fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SGPO_HIP_LIB') or os.path.join(_HERE, 'libsgpo_hip.so')   # override: diagnostic builds only

NOISE_PHILOX, NOISE_EXPONENTIAL, NOISE_UNIFORM, NOISE_NONE = 0, 1, 2, 3
STEP_SOFT, STEP_SOFT_CG, STEP_HARD = 0, 1, 2
PROF_KINDS = 20


class SgpoError(RuntimeError):
    pass


class DitConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('vocab', 'hidden', 'cond_dim', 'freq_dim', 'n_blocks', 'n_heads',
                                       'mlp_hidden', 'max_len')]


_vp, _i, _f, _u64, _sz, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t, C.c_int64

# name -> (restype, argtypes); must list every symbol include/sgpo_hip.h declares
SIGNATURES = {
    'sgpo_init': (_i, [_i]),
    'sgpo_last_error': (C.c_char_p, []),
    'sgpo_version': (_i, []),
    'sgpo_dit_param_count': (_sz, [C.POINTER(DitConfig)]),
    'sgpo_dit_create': (_i, [C.POINTER(DitConfig), _vp, _sz, C.POINTER(_vp)]),
    'sgpo_dit_destroy': (None, [_vp]),
    'sgpo_dit_workspace_bytes': (_sz, [_vp, _i, _i]),
    'sgpo_dit_forward': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sgpo_reverse_step_mdlm': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _vp, _u64, _u64,
                                    _u64, _vp, _vp, _vp, _vp]),
    'sgpo_reverse_step_mdlm_devkey': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _i, _u64, _vp, _vp, _vp, _vp, _vp]),
    'sgpo_reverse_step_udlm': (_i, [_vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    'sgpo_guided_step_udlm': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _i, _vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp]),
    'sgpo_q_sample_udlm': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    'sgpo_q_sample_mdlm': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _u64, _u64, _u64, _vp, _vp]),
    'sgpo_categorical_sample': (_i, [_vp, _i64, _i, _i, _vp, _u64, _u64, _u64, _vp, _vp]),
    'sgpo_mlp_create': (_i, [_vp, _sz, _i, C.POINTER(_i), _i, _i, _i, C.POINTER(_vp)]),
    'sgpo_mlp_destroy': (None, [_vp]),
    'sgpo_mlp_copy': (_i, [_vp, _vp, _vp]),
    'sgpo_mlp_forward': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    'sgpo_mlp_workspace_bytes': (_sz, [_vp, _i]),
    'sgpo_guidance_ratio_mlp': (_i, [_vp, _vp, _f, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'sgpo_guidance_ratio_mlp_absorbing': (_i, [_vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'sgpo_guided_step_groups_mdlm': (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz,
                                          _vp, _sz, _vp, _vp, _vp]),
    'sgpo_guided_step_groups_mdlm_devkey': (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _f, _f, _u64, _vp, _vp, _vp, _sz,
                                                 _vp, _sz, _vp, _vp, _vp]),
    'sgpo_gemm_q_bf16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'sgpo_guidance_noalloc_max_bytes': (_sz, []),
    'sgpo_guidance_ratio_mlp_absorbing_noalloc': (_i, [_vp, _vp, _f, _i, _i, _i, _i, _vp, _vp]),
    'sgpo_fitness_create': (_i, [_vp, _sz, _i, _i, _i, C.POINTER(_vp)]),
    'sgpo_fitness_destroy': (None, [_vp]),
    'sgpo_fitness_workspace_bytes': (_sz, [_vp, _i]),
    'sgpo_fitness_forward': (_i, [_vp, _vp, _i, _vp, _i, C.c_double, _vp, _vp, _vp, _sz, _vp]),
    'sgpo_project_tokens': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    'sgpo_acq_create': (_i, [_i, _sz, _i, C.POINTER(_vp)]),
    'sgpo_acq_destroy': (None, [_vp]),
    'sgpo_acq_count': (C.c_longlong, [_vp]),
    'sgpo_acq_update': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_i), C.POINTER(_i), _vp]),
    'sgpo_acq_read': (_i, [_vp, C.c_longlong, C.c_longlong, _vp, _vp, _vp]),
    'sgpo_landscape_lookup': (_i, [_vp, _sz, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    'sgpo_mlp_trainer_param_count': (_sz, [_i, _i, _i]),
    'sgpo_mlp_trainer_create': (_i, [_vp, _i, _i, C.POINTER(_i), _i, _i, _i, C.POINTER(_vp)]),
    'sgpo_mlp_trainer_destroy': (None, [_vp]),
    'sgpo_mlp_trainer_step': (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_double, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    'sgpo_mlp_trainer_steps': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.c_double, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    'sgpo_mlp_trainer_read': (_i, [_vp, _vp, _vp]),
    'sgpo_nos_workspace_bytes': (_sz, [_vp, _i, _i]),
    'sgpo_nos_denoise': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _f, _f, _f, _f, _i, _i, _i, _vp, _u64, _u64, _u64,
                              _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    'sgpo_bytenet_param_count': (_sz, [_vp]),
    'sgpo_bytenet_create': (_i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    'sgpo_bytenet_destroy': (None, [_vp]),
    'sgpo_bytenet_workspace_bytes': (_sz, [_vp, _i, _i]),
    'sgpo_bytenet_forward': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'sgpo_reverse_step_d3pm': (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    'sgpo_q_sample_d3pm': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _u64, _u64, _u64, _vp, _vp]),
    'sgpo_daps_mh_workspace_bytes': (_sz, []),
    'sgpo_daps_mh': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _f, _f, _f, _i, _i, _vp, _vp, _vp, _u64, _u64, _u64,
                          _vp, _vp, _sz, _vp]),
    'sgpo_gemm_bf16': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'sgpo_f32_to_bf16': (_i, [_vp, _vp, _sz, _vp]),
    'sgpo_prof_enable': (_i, [_i, _i]),
    'sgpo_prof_reset': (_i, []),
    'sgpo_prof_read': (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    'sgpo_prof_kind_name': (C.c_char_p, [_i]),
}

_lib = None


def load():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SgpoError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                            f'or `make -C sgpo_amd/csrc`. There is no CPU fallback.')
        # torch first: libsgpo_hip.so must bind to the HIP runtime PyTorch-ROCm ships (same soname);
        # loading ours first pulls /opt/rocm's copy under torch and the GPU is no longer detected.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = load().sgpo_last_error().decode(errors='replace')
        raise SgpoError(f'{what} failed ({rc}): {msg}')


def ptr(t):
    """Device address of a torch tensor (None -> NULL).  Every tensor argument of the library is device memory: a
    host tensor here would hand a kernel a host pointer (a GPU memory fault), so it is refused."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SgpoError(f'host tensor {tuple(t.shape)} {t.dtype} passed where the library expects device memory')
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class BytenetConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('n_tokens', 'd_embedding', 'd_model', 'd_h', 'n_layers', 'kernel_size', 'r')]
