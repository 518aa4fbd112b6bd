"""ctypes loader for libcusmc_hip.so (the C ABI of include/cusmc_hip.h).

There is no Python or CPU fallback behind this module: if the shared library has not been built
(`make -C cusmc_amd/csrc`, or `python -c "import __graft_entry__ as g; g.build()"`), importing
the symbols fails loudly, and every compute entry point fails with CUSMC_ENODEVICE when no
gfx950 device is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CUSMC_LIBRARY: another build of the same ABI -- a sanitizer or calibration build; scripts/host_sanitizers.sh)
SO_PATH = os.environ.get("CUSMC_LIBRARY") or os.path.join(_HERE, "libcusmc_hip.so")

OK, EINVAL, ENOTSPD, EHIP, ENODEVICE, ERANGE = range(6)
MVN, MVT = 0, 1
OUT_LOG, OUT_DENSITY = 0, 1

# every symbol include/cusmc_hip.h declares: (name, restype, argtypes)
_vp, _i, _i64, _u32, _u64, _f, _d = (C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64,
                                     C.c_float, C.c_double)
_pp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("cusmc_version", C.c_char_p, []),
    ("cusmc_rng_contract", _i, []),
    ("cusmc_last_error", C.c_char_p, []),
    ("cusmc_device_count", _i, []),
    ("cusmc_stream_key", _u64, [_u64, _u64]),
    ("cusmc_ctx_create", _i, [_i, _pp]),
    ("cusmc_ctx_destroy", _i, [_vp]),
    ("cusmc_ctx_set_stream", _i, [_vp, _vp]),
    ("cusmc_ctx_synchronize", _i, [_vp]),
    ("cusmc_ctx_num_cus", _i, [_vp, C.POINTER(_i)]),
    ("cusmc_dist_create", _i, [_vp, _i, _vp, _vp, _i, _f, _pp]),
    ("cusmc_dist_destroy", _i, [_vp]),
    ("cusmc_dist_lognorm", _i, [_vp, C.POINTER(_d)]),
    ("cusmc_dist_logdet", _i, [_vp, C.POINTER(_d)]),
    ("cusmc_dist_pdf_dev", _i, [_vp, _vp, _i64, _i64, _vp, _i, _vp]),
    ("cusmc_dist_pdf_host", _i, [_vp, _vp, _i64, _i64, _vp, _i, _vp]),
    ("cusmc_dist_reweight_dev", _i, [_vp, _vp, _i64, _i64, _vp, _vp, _i, _vp]),
    ("cusmc_dist_reweight_host", _i, [_vp, _vp, _i64, _i64, _vp, _vp, _i, _vp]),
    ("cusmc_dist_pdf_multi_host", _i, [_vp, C.POINTER(_i), _i, _vp, _i64, _i64, _vp, _i, _vp]),
    ("cusmc_dist_reweight_multi_host", _i, [_vp, C.POINTER(_i), _i, _vp, _i64, _i64, _vp, _vp, _i, _vp]),
    ("cusmc_metropolis_dev", _i, [_vp, _vp, _u32, _u32, _u64, _u32, _u32, _u32, _vp]),
    ("cusmc_metropolis_host", _i, [_vp, _vp, _u32, _u32, _u64, _u32, _vp]),
    ("cusmc_metropolis_log_dev", _i, [_vp, _vp, _u32, _u32, _u64, _u32, _u32, _u32, _vp]),
    ("cusmc_metropolis_log_host", _i, [_vp, _vp, _u32, _u32, _u64, _u32, _vp]),
    ("cusmc_metropolis_multi_host", _i, [C.POINTER(_i), _i, _vp, _u32, _u32, _u64, _u32, _i, _vp]),
    ("cusmc_propagate_dev", _i, [_vp, _i, _f, _vp, _vp, _u32, _i, _vp, _vp, _d, _u64, _u32, _u32,
                                 _u32, _vp]),
    ("cusmc_initialize_dev", _i, [_vp, _i, _f, _vp, _vp, _i, _d, _u64, _u32, _u32, _vp]),
    ("cusmc_sample_host", _i, [_vp, _i, _f, _vp, _vp, _i, _d, _u64, _u32, _u32, _vp]),
    ("cusmc_eigen_sqrt", _i, [_vp, _i, _vp]),
    ("cusmc_chol_batched_dev", _i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    ("cusmc_chol_batched_host", _i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    ("cusmc_logpdf_percov_dev", _i, [_vp, _i, _f, _vp, _i64, _i64, _vp, _i64, _vp, _i, _i, _vp, _vp]),
    ("cusmc_logpdf_percov_host", _i, [_vp, _i, _f, _vp, _i64, _i64, _vp, _i64, _vp, _i, _i, _vp, _vp]),
    ("cusmc_pf_step_dev", _i, [_vp, _i, _f, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _d, _u64, _u32, _u32,
                               _u32, _vp, _vp, _vp, _i]),
    ("cusmc_pf_run_host", _i, [_vp, _vp, _u32, _i, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _f,
                               C.c_char_p, C.c_char_p, _u32, _d, _u64, _vp, _vp, _vp]),
    ("cusmc_pf_run_multi_host", _i, [C.POINTER(_i), _i, _vp, _u32, _i, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _f,
                                     C.c_char_p, C.c_char_p, _u32, _d, _u64, _vp, _vp, _vp]),
]

_lib = None


class CusmcError(RuntimeError):
    """A non-zero status from libcusmc_hip (the R glue turns the same status into Rcpp::stop)."""

    def __init__(self, code, message):
        super().__init__("cusmc error %d: %s" % (code, message))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "libcusmc_hip.so is not built (%s). Build it with `make -C cusmc_amd/csrc`; "
                "there is no CPU fallback." % SO_PATH)
        # One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 /
        # libhsa-runtime64; if this library pulls in /opt/rocm's copy first, torch's later
        # initialisation finds "No HIP GPUs".  Loaded in the other order both share torch's copy.
        # So when torch is installed (it provides the device tensors and streams of the
        # device-resident API anyway) it is imported before the first dlopen.
        if os.environ.get("CUSMC_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(SO_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the ABI and this table drift apart
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(status):
    if status != OK:
        raise CusmcError(status, lib().cusmc_last_error().decode())
