"""ctypes binding of libcaf.so (include/caf.h).

Follows the reference's own C-DLL idiom (cpuWola.py:38-70, cpuTone.py:28-47,
cython_ext/compareIntPreambles/compareIntPreambles.py:7-52): the shared library
sits next to the Python modules and is loaded with ``np.ctypeslib.load_library``;
every entry point returns an int32 status (0 = success) and writes into
caller-allocated buffers.

There is NO CPU fallback: if the library is missing or a call fails, the host
raises.  The oracle under ``oracle/`` is test infrastructure and is never
imported from here.
"""

import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.realpath(__file__))

CAF_OK = 0
CAF_ERR_INVALID = 1
CAF_ERR_HIP = 2
CAF_ERR_ROCFFT = 3
CAF_ERR_NOMEM = 4
CAF_ERR_NODEVICE = 5

CAF_FREQ_BINS = 0
CAF_FREQ_NORM = 1
CAF_ENGINE_AUTO = 0
CAF_ENGINE_ROCFFT = 1
CAF_ENGINE_FUSED = 2
CAF_ENGINE_PERSISTENT = 3
CAF_ENGINE_DIRECT = 4
ENGINE_IDS = {
    "auto": CAF_ENGINE_AUTO,
    "rocfft": CAF_ENGINE_ROCFFT,
    "fused": CAF_ENGINE_FUSED,
    "persistent": CAF_ENGINE_PERSISTENT,
    "direct": CAF_ENGINE_DIRECT,
}
ENGINE_NAMES = {v: k for k, v in ENGINE_IDS.items() if k != "auto"}
CAF_NUM_STAGES = 7
STAGE_NAMES = (
    "energy_prefix",
    "gather_blocks",
    "fft_forward",
    "spectral_conj_multiply",
    "fft_inverse(rocFFT)",
    "magsq_norm_argmax",
    "peak_reduce",
)


class CafPlanDesc(ct.Structure):
    _fields_ = [
        ("num_templates", ct.c_int32),
        ("template_len", ct.c_int32),
        ("h_templates", ct.c_void_p),
        ("auto_conj", ct.c_int32),
        ("num_groups", ct.c_int32),
        ("h_group_start", ct.c_void_p),
        ("h_group_len", ct.c_void_p),
        ("freq_mode", ct.c_int32),
        ("num_freqs", ct.c_int32),
        ("h_bins", ct.c_void_p),
        ("grid", ct.c_int32),
        ("h_freqs_norm", ct.c_void_p),
        ("max_rx_len", ct.c_int64),
        ("log2_block", ct.c_int32),
        ("blocks_per_batch", ct.c_int32),
        ("engine", ct.c_int32),
        ("reserved", ct.c_int32),
    ]


class CafOutputs(ct.Structure):
    _fields_ = [
        ("d_surface", ct.c_void_p),
        ("d_row_max", ct.c_void_p),
        ("d_row_arg", ct.c_void_p),
        ("d_peak_val", ct.c_void_p),
        ("d_peak_delay", ct.c_void_p),
        ("d_peak_freq", ct.c_void_p),
        ("d_cqf", ct.c_void_p),
    ]


class CafOutputs2(ct.Structure):
    _fields_ = [("base", CafOutputs), ("d_surface_t", ct.c_void_p), ("reserved", ct.c_void_p * 3)]


class CafZoomOutputs(ct.Structure):
    _fields_ = [
        ("d_count", ct.c_void_p),
        ("d_delay", ct.c_void_p),
        ("d_coarse_freq_index", ct.c_void_p),
        ("d_coarse_qf2", ct.c_void_p),
        ("d_fine_index", ct.c_void_p),
        ("d_fine_freq", ct.c_void_p),
        ("d_fine_qf2", ct.c_void_p),
        ("d_planes", ct.c_void_p),
    ]


_P = ct.c_void_p
_I32 = ct.c_int32
_I64 = ct.c_int64

# name -> argtypes; restype is always int32 (the reference DLL convention)
_SIGNATURES = {
    "caf_last_error": [ct.c_char_p, _I32],
    "caf_abi_version": [],
    "caf_xcorr_perdelay_one_kernel": [_I32],
    "caf_perdelay_jit_describe": [_I32, ct.c_char_p, ct.c_char_p, ct.c_char_p, _I32],
    "caf_device_count": [ct.POINTER(_I32)],
    "caf_set_device": [_I32],
    "caf_device_info": [_I32, ct.c_char_p, _I32, ct.POINTER(_I64), ct.POINTER(_I32)],
    "caf_malloc": [ct.POINTER(_P), _I64],
    "caf_free": [_P],
    "caf_pool_trim": [],
    "caf_pool_stats": [ct.POINTER(_I64), ct.POINTER(_I64), ct.POINTER(_I64), ct.POINTER(_I64)],
    "caf_memset": [_P, _I32, _I64, _P],
    "caf_h2d": [_P, _P, _I64, _P],
    "caf_d2h": [_P, _P, _I64, _P],
    "caf_d2d": [_P, _P, _I64, _P],
    "caf_d2h_transposed": [_P, _I32, _P, _I64, _I64, _I64, _I64, _P],
    "caf_d2h_f64": [_P, _P, _I64, _P],
    "caf_stream_sync": [_P],
    "caf_stream_create": [ct.POINTER(_P)],
    "caf_stream_destroy": [_P],
    "caf_plan_create": [ct.POINTER(_P), ct.POINTER(CafPlanDesc)],
    "caf_plan_destroy": [_P],
    "caf_plan_info": [_P, ct.POINTER(_I32), ct.POINTER(_I32), ct.POINTER(_I32), ct.POINTER(_I64)],
    "caf_plan_engine": [_P, ct.POINTER(_I32)],
    "caf_plan_execute": [_P, _P, _I64, _I64, _I64, ct.POINTER(CafOutputs), _P],
    "caf_plan_execute2": [_P, _P, _I64, _I64, _I64, ct.POINTER(CafOutputs2), _P],
    "caf_plan_watchdog": [_P, ct.POINTER(_I32)],
    "caf_plan_profile": [_P, _I32],
    "caf_plan_profile_get": [_P, ct.POINTER(ct.c_double), ct.POINTER(_I64)],
    "caf_plan_execute_host": [_P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P],
    "caf_xcorr_perdelay": [_P, _I32, _P, _I64, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _I64, _P],
    "caf_fft_rows": [_P, _P, _I64, _I64, _I32, _P],
    "caf_sliding_multiply_normalised": [_P, _I32, _P, _I64, _I64, _I64, ct.c_double, _P, _P],
    "caf_multi_template_sliding_dot": [_P, _P, _I32, _I32, _P, _I64, _I64, _I64, _P, _P, _P],
    "caf_multiply_slices_indexed_rows": [_P, _I64, _P, _I32, _I32, _P, _P, _P, _I32, _I64, _P, _P],
    "caf_complex_magnsq": [_P, _I64, _I32, _P, _I32, _P],
    "caf_argmax_abs_rows": [_P, _I64, _I64, _P, _P, _I32, _P],
    "caf_moving_average": [_P, _I64, _I64, _I32, _I32, _P, _P],
    "caf_complex_moving_sum": [_P, _I64, _I32, _P, _P],
    "caf_copy_slices_to_matrix": [_P, _I64, _P, _I32, _I64, _I64, _I32, _I64, _P, _P],
    "caf_copy_groups": [_P, _P, _P, _P, _P, _I32, _P],
    "caf_find_local_maxima": [_P, _I64, ct.c_float, _I32, _P, _P, _P],
    "caf_gather_b32": [_P, _I64, _P, _I64, _P, _P],
    "caf_gather_f32_f64": [_P, _I64, _P, _I64, _P, _P],
    "caf_fir_lfilter": [_P, _I64, _P, _I32, _P, _I32, _I32, _I32, _P, _I64, _P],
    "caf_upfirdn": [_P, _I64, _I64, _P, _I32, _I32, _I32, _P, _P, _I64, _P],
    "caf_czt_run_many": [_P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "caf_argmax3d_u32": [_P, _I64, _I32, _I32, _I32, _P, _P, _P],
    "caf_iq16_to_c64": [_P, _I64, ct.c_float, _P, _P],
    "caf_iq16_fir_decimate": [_P, _I64, ct.c_float, _P, _I32, _P, _I32, _I32, _I32, _P, _I64, _P],
    "caf_colmax_abs": [_P, _I32, _I64, _P, _P, _I32, _P],
    "caf_colmax_sqrt": [_P, _I32, _I64, _P, _P, _P],
    "caf_dot_tones": [_P, _I64, ct.c_double, ct.c_double, _I32, _P, _P],
    "caf_mul_conj": [_P, _P, _I64, _P, _P],
    "caf_steer_dot": [_P, _P, _I64, _I64, ct.c_double, _P, _P],
    "caf_sum_planes_qf2": [_P, _I32, _I64, _I32, _P, _I32, _P, ct.c_double, _P, _P],
    "caf_sum_groups_qf2": [_P, _I32, _I64, _I32, _P, _P, ct.c_double, _P, _P],
    "caf_comm_unique_id": [_P],
    "caf_comm_create": [ct.POINTER(_P), _I32, _I32, _P],
    "caf_comm_destroy": [_P],
    "caf_peak_table_allgather": [_P, _P, _I32, _P, _P],
    "caf_zoom_num_bins": [ct.c_double, ct.c_double, ct.POINTER(_I32)],
    "caf_zoom_czt": [_P, _I32, _P, _I64, _P, _P, _I64, _I64, _I32, ct.c_float, ct.c_double, ct.c_double,
                     ct.POINTER(CafZoomOutputs), _P],
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def _share_rocm_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so / librocfft.so under torch/lib and ask
    for them by file name ("libamdhip64.so", RPATH $ORIGIN), while libcaf.so asks for the soname ("libamdhip64.so.7").
    If torch is imported AFTER libcaf.so has pulled in the system ROCm, the dynamic linker does not recognise the two
    as the same library and the process ends up with two HIP + two HSA runtimes: torch then reports "no ROCm-capable
    device" and the interpreter aborts in free() while both tear down at exit.  With the bundled copies loaded first,
    libcaf's soname requests and torch's file-name requests resolve to the same objects in either import order.
    Nothing is imported from torch here; CAF_SYSTEM_ROCM=1 keeps the system runtime (a process that never loads torch).
    """
    import sys

    if os.environ.get("CAF_SYSTEM_ROCM") == "1" or "torch" in sys.modules:
        return  # torch already loaded: its runtime is in the process and libcaf's sonames bind to it
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so", "librocfft.so", "librccl.so"):  # rccl: caf_comm_* dlopens its soname
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            ct.CDLL(path)  # RTLD_LOCAL: global scope would interpose the amd::smi statics of libamd_smi


def load():
    """Load libcaf.so (once).  Raises OSError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    _share_rocm_runtime_with_torch()
    try:
        # CAF_LIBRARY=<name>: a development switch for A/B runs of two builds on one box (e.g. libcaf_r02, kept beside
        # libcaf.so); the product always loads libcaf.so
        lib = np.ctypeslib.load_library(os.environ.get("CAF_LIBRARY", "libcaf"), _HERE)
    except OSError as e:
        raise OSError(
            "pydsproutines_amd: libcaf.so (the HIP extension) is missing or cannot be loaded from %s "
            "(%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C pydsproutines_amd/csrc`. There is no CPU fallback." % (_HERE, e)
        ) from e
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _I32
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    buf = ct.create_string_buffer(1024)
    load().caf_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(rc, what=""):
    """Map a libcaf status to the reference's exception types
    (ValueError shape/range, MemoryError resource, RuntimeError 'DLL returned an error')."""
    if rc == CAF_OK:
        return
    msg = "%s: %s" % (what or "libcaf", last_error())
    if rc == CAF_ERR_INVALID:
        raise ValueError(msg)
    if rc == CAF_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError("DLL returned an error. " + msg)


def device_count():
    n = _I32(0)
    rc = load().caf_device_count(ct.byref(n))
    return n.value if rc == CAF_OK else 0


def require_device():
    """Fail loudly when no GPU is usable (no silent CPU path)."""
    if device_count() < 1:
        raise RuntimeError("pydsproutines_amd: no MI355X/HIP device visible: " + last_error())
