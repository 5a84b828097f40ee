"""ctypes binding of librepmode_hip.so (the C ABI declared in include/repmode_hip.h).

The library is built in-tree by ``repmode_amd/csrc/build.sh`` (``__graft_entry__.build()``).
There is NO fallback: if the shared object is missing or a call fails this module raises.
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: torch bundles its own libamdhip64.so.7; importing it before dlopen()ing
# librepmode_hip.so makes that the single HIP runtime of the process (same soname as /opt/rocm's copy).

_HERE = os.path.dirname(os.path.abspath(__file__))
# REPMODE_LIB: a developer's variant build of the kernel library (tools/ab_variant.sh, tools/conv_phase_timing.py)
LIB_PATH = os.environ.get('REPMODE_LIB') or os.path.join(_HERE, 'librepmode_hip.so')

F32, BF16 = 0, 1
DEFER = 1        # REPMODE_DEFER: queue the job for the next conv5 launch on the stream (include/repmode_hip.h)
ABI_VERSION = 11

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int

# name -> argtypes, in the order of include/repmode_hip.h
_SIGNATURES = {
    'repmode_abi_version': [],
    'repmode_device_arch': [_I, _c.c_char_p, _I],
    'repmode_set_deterministic': [_I],
    'repmode_get_deterministic': [],
    'repmode_set_reserve_cus': [_I],
    'repmode_get_reserve_cus': [],
    'repmode_set_conv_pipe': [_I],
    'repmode_get_conv_pipe': [],
    'repmode_conv5_elem_out': [_I] * 7,
    'repmode_adam_multi': [_I, _P, _P, _P, _P, _P, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_long, _P],
    'repmode_adam_expert_frags': [_I] + [_P] * 12 + [_c.c_double] * 4 + [_c.c_long, _P],
    'repmode_adam_hyper_dev': [_P, _P, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _P],
    'repmode_adam_multi_dev': [_I, _P, _P, _P, _P, _P, _P, _P],
    'repmode_adam_expert_frags_dev': [_I] + [_P] * 12 + [_P, _P],
    'repmode_set_wgrad_ws': [_I],
    'repmode_get_wgrad_ws': [],
    'repmode_set_wgrad_col': [_I],
    'repmode_set_wgrad_col_split': [_I],
    'repmode_get_wgrad_col_split': [],
    'repmode_set_bn_fused': [_I],
    'repmode_get_bn_fused': [],
    'repmode_get_wgrad_col': [],
    'repmode_padded_channels': [_I, _I, _I],
    'repmode_gate_softmax': [_P, _P, _P, _I, _I, _I, _P, _P],
    'repmode_gate_softmax_multi': [_I, _P, _P, _P, _P, _I, _I, _P, _P],
    'repmode_gatrep_fwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    'repmode_gatrep_fwd_multi': [_I] + [_P] * 10 + [_I, _I, _I] + [_P] * 4,
    'repmode_gatrep_fwd_gate': [_P] * 8 + [_I] * 5 + [_P] * 4,
    'repmode_conv5': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_ex': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_deep': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_deep_supported': [_I, _I, _I],
    'repmode_deep_mode_plan': [_I] * 8,
    'repmode_deep_mode_fwd': [_P] * 9 + [_I] * 6 + [_P],
    'repmode_deep_mode_fwd_ex': [_P] * 9 + [_I] * 7 + [_P, _P],
    'repmode_deep_mode_dgrad': [_P] * 9 + [_I] * 7 + [_P],
    'repmode_conv5_thin_in1': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'repmode_conv5_thin_out1': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_merged': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_pair': [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_epi': [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    'repmode_conv5_wgrad': [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_wgrad_ex': [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_wgrad_part': [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_wgrad_plan': [_I] * 9 + [_P],
    'repmode_conv5_wgrad_dual': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_conv5_wgrad_thin': [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    'repmode_shift5': [_P, _I, _P, _c.c_long, _I, _P],
    'repmode_thin_pack': [_P, _P, _I, _I, _I, _P],
    'repmode_unshift5': [_P, _P, _c.c_long, _I, _P],
    'repmode_gatrep_bwd': [_P] * 8 + [_I] * 4 + [_P] * 9,
    'repmode_gatrep_bwd_ex': [_P] * 8 + [_I] * 4 + [_P] * 8 + [_I, _P],
    'repmode_tail_flush': [_P],
    'repmode_tail_discard': [_P],
    'repmode_bn_relu_fwd': [_P] * 8 + [_c.c_long, _I, _c.c_float, _c.c_float, _I, _I, _I, _P],
    'repmode_bn_relu_fwd_ex': [_P] * 8 + [_c.c_long, _I, _c.c_float, _c.c_float, _I, _I, _I, _I, _P],
    'repmode_bn_relu_bwd': [_P] * 8 + [_c.c_long, _I, _I, _I, _I, _P],
    'repmode_k2s2': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_k2s2_wgrad': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'repmode_k2s2_wgrad_ex': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_k2_frags': [_P, _I, _I, _I, _I, _P, _P],
    'repmode_k2_frags2': [_P, _I, _I, _I, _I, _P, _P, _P],
    'repmode_k2_frags_multi': [_I, _P, _P, _P, _P, _I, _P, _P, _P],
    'repmode_expert_mix_fwd': [_P, _P, _P, _I, _c.c_long, _I, _P],
    'repmode_expert_mix_bwd': [_P, _P, _P, _P, _P, _P, _I, _c.c_long, _I, _I, _P],
    'repmode_expert_mix_bwd_ex': [_P, _P, _P, _P, _P, _P, _c.c_long, _I, _c.c_long, _I, _I, _P],
    'repmode_expert_mix_bwd_box': [_P, _P, _P, _P, _P, _P, _c.c_long, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'repmode_gemm3': [_P, _c.c_long, _c.c_long, _P, _c.c_long, _c.c_long, _P, _I, _I, _I, _I, _I, _I, _P],
    'repmode_box_expand': [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    'repmode_box_pair': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'repmode_concat_channels': [_P, _P, _P, _c.c_long, _I, _I, _P],
    'repmode_box_sum': [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    'repmode_box_sum_ex': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'repmode_tap_transpose': [_P, _P, _c.c_long, _I, _P],
    'repmode_tap_transpose_ex': [_P, _P, _c.c_long, _I, _I, _P],
    'repmode_gate_bwd': [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    'repmode_gate_bwd_ex': [_P, _P, _P, _I, _I, _I, _P, _P, _I, _P],
    'repmode_expert_frags': [_P, _P, _I, _I, _P, _P, _P],
    'repmode_crop_flip': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    'repmode_mse_loss': [_P, _P, _P, _I, _c.c_long, _I, _P, _P, _P, _P, _P, _P, _P],
    'repmode_patch_gather': [_P, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P],
    'repmode_patch_blend': [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P],
    'repmode_expert_frags_multi': [_I, _P, _P, _P, _P, _P, _P, _P],
    'repmode_expert_frags_refresh_multi': [_I, _P, _P, _P, _P, _P, _P, _P, _P],
    'repmode_prof_enable': [_I],
    'repmode_prof_pause': [_I],
    'repmode_prof_summary': [_I, _P, _P, _P],
    'repmode_prof_count': [],
    'repmode_prof_record': [_I, _P, _P, _P],
    'repmode_debug_conv5_naive': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'repmode_debug_wgrad_naive': [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
}
EXPORTS = sorted(list(_SIGNATURES) + ['repmode_last_error'])

_lib = None


class RepModeHipError(RuntimeError):
    pass


def load():
    """dlopen the library once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RepModeHipError(
            'librepmode_hip.so not found at %s -- build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950). '
            'repmode_amd has no CPU or eager fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _I
    lib.repmode_last_error.argtypes = []
    lib.repmode_last_error.restype = _c.c_char_p
    if lib.repmode_abi_version() != ABI_VERSION:
        raise RepModeHipError('ABI mismatch: library %d, binding %d' % (lib.repmode_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


TORCH_LIB_PATH = os.environ.get('REPMODE_TORCH_LIB') or os.path.join(_HERE, 'librepmode_torch.so')
_torch_ops_loaded = False


def load_torch_ops():
    """Load the operator seam (``torch.ops.repmode.*``: C++ ops + autograd over the C ABI, csrc/torch/repmode_ops.cpp).
    Raises -- never falls back -- when it has not been built."""
    global _torch_ops_loaded
    if _torch_ops_loaded:
        return
    lib = load()                # librepmode_hip.so first (the operator library links against it)
    if os.environ.get('REPMODE_LIB') and not os.environ.get('REPMODE_TORCH_LIB'):
        # the operator library links librepmode_hip.so BY NAME (rpath $ORIGIN): with only the ctypes handle redirected, the
        # network path would run the in-tree kernels next to a second copy of the library's state
        raise RepModeHipError('REPMODE_LIB selects a variant kernel library for the ctypes layer only; the operator seam needs '
                              'REPMODE_TORCH_LIB = a librepmode_torch.so built against it (csrc/build.sh with REPMODE_OUT / '
                              'REPMODE_TORCH_OUT in one directory)')
    if lib.repmode_abi_version() != ABI_VERSION:
        raise RepModeHipError('ABI mismatch: library %d, binding %d' % (lib.repmode_abi_version(), ABI_VERSION))
    if not os.path.exists(TORCH_LIB_PATH):
        raise RepModeHipError(
            'librepmode_torch.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"`. '
            'repmode_amd has no CPU or eager fallback.' % TORCH_LIB_PATH)
    torch.ops.load_library(TORCH_LIB_PATH)
    _torch_ops_loaded = True
    from .optim import install_foreign_step_hook      # (the store of expert operands kept across steps: see optim.py)
    install_foreign_step_hook()


def torch_ops_loaded():
    return _torch_ops_loaded


_FUNCS = {}     # name -> bound foreign function (the hot path makes ~500 calls per train step)


def call(name, *args):
    """Call an int-returning entry point; raise with the library's message on failure."""
    fn = _FUNCS.get(name)
    if fn is None:
        fn = _FUNCS[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RepModeHipError('%s failed (code %d): %s' % (name, rc, load().repmode_last_error().decode()))


def padded_channels(channels, dtype_code, is_reduction_dim):
    return load().repmode_padded_channels(channels, dtype_code, 1 if is_reduction_dim else 0)


def device_arch(dev=0):
    buf = ctypes.create_string_buffer(64)
    call('repmode_device_arch', dev, buf, 64)
    return buf.value.decode()


PROF_KINDS = {'conv5_igemm': 0, 'conv5_wgrad': 1, 'gatrep_fwd': 2, 'gatrep_bwd': 3, 'conv5_wgrad_thin': 4, 'conv5_deep': 5,
              'conv5_thin': 6, 'conv5_ws': 7, 'deep_mode': 8, 'helper': 9, 'deep_mode_dgrad': 10}


def prof_enable(on):
    """False/0: off; True/1: every kernel kind; 2: the MoDE convolution's MFMA kernels only (forward / data gradient / filter gradient)."""
    call('repmode_prof_enable', int(on))


def prof_pause(paused):
    call('repmode_prof_pause', int(bool(paused)))


def prof_summary(kind):
    """(launches, total_ms, total_work) of one kernel kind since prof_enable(True)."""
    n, ms, work = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
    call('repmode_prof_summary', PROF_KINDS[kind], ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work))
    return n.value, ms.value, work.value


def prof_records():
    """[(kind_name, ms, work)] of every recorded launch, in launch order."""
    lib = load()
    names = {v: k for k, v in PROF_KINDS.items()}
    out = []
    for i in range(lib.repmode_prof_count()):
        kind, ms, work = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
        call('repmode_prof_record', i, ctypes.byref(kind), ctypes.byref(ms), ctypes.byref(work))
        out.append((names[kind.value], ms.value, work.value))
    return out
