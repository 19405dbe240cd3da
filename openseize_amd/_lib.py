"""ctypes binding of libosz_hip.so (the C ABI declared in include/osz_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot
be loaded, every numerical entry point raises ``OszLibraryError``.
"""

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSZ_HIP_LIB points the binding at another build of the same C ABI
LIB_PATH = os.environ.get("OSZ_HIP_LIB") or os.path.join(_HERE, "lib", "libosz_hip.so")
CSRC = os.path.join(_HERE, "csrc")

OSZ_OK = 0
OSZ_ERR_INVALID = -1
OSZ_ERR_HIP = -2
OSZ_ERR_NOMEM = -3
OSZ_ERR_STATE = -4
OSZ_ERR_UNSUPPORTED = -5

SPEC_PSD_MEAN, SPEC_PSD_SEGMENTS, SPEC_DFT_SEGMENTS = 0, 1, 2
EW_ADD, EW_MUL, EW_DIV, EW_STANDARDIZE = 0, 1, 2, 3
BCAST_SCALAR, BCAST_ROW, BCAST_COL, BCAST_FULL = 0, 1, 2, 3
DETREND = {"constant": 0, "linear": 1}


class OszLibraryError(RuntimeError):
    """libosz_hip.so is missing or failed to load."""


c_dp = ctypes.POINTER(ctypes.c_double)
c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/osz_hip.h one to one
CHAIN_DEFER = 1

SIGNATURES = {
    "osz_version": (ctypes.c_int, []),
    "osz_last_error": (ctypes.c_char_p, []),
    "osz_device_info": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.c_char_p, ctypes.c_int]),
    "osz_malloc": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_size_t]),
    "osz_free": (ctypes.c_int, [c_vp]),
    "osz_memcpy_h2d": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "osz_memcpy_d2h": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "osz_memcpy_d2d": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "osz_memset": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_size_t, c_vp]),
    "osz_stream_sync": (ctypes.c_int, [c_vp]),
    "osz_event_create": (ctypes.c_int, [ctypes.POINTER(c_vp)]),
    "osz_event_destroy": (ctypes.c_int, [c_vp]),
    "osz_event_record": (ctypes.c_int, [c_vp, c_vp]),
    "osz_event_elapsed_ms": (ctypes.c_int, [c_vp, c_vp,
                                            ctypes.POINTER(ctypes.c_float)]),
    "osz_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "osz_profile_reset": (ctypes.c_int, []),
    "osz_profile_query": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(c_i64),
                                         ctypes.POINTER(ctypes.c_double)]),
    "osz_sos_create": (ctypes.c_int, [ctypes.POINTER(c_vp), c_dp, ctypes.c_int,
                                      ctypes.c_int]),
    "osz_sos_destroy": (ctypes.c_int, [c_vp]),
    "osz_sos_set_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_sos_get_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_sos_set_zi_unit": (ctypes.c_int, [c_vp, c_dp]),
    "osz_sos_set_state_scaled": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "osz_sos_forward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64,
                                       c_vp]),
    "osz_sos_warmup_len": (c_i64, [c_vp]),
    "osz_sos_set_warmup_len": (ctypes.c_int, [c_vp, c_i64]),
    "osz_sosfiltfilt_chunk": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp,
                                             c_i64, c_i64, c_vp, c_i64, c_vp]),
    "osz_sosfiltfilt_step": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64,
                                            c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                            c_vp, c_i64, c_vp]),
    "osz_chain_forward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "osz_chain_forward_route": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "osz_chain_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64,
                                      c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64,
                                      ctypes.c_int, c_vp]),
    "osz_chain_wait": (ctypes.c_int, [c_vp, c_vp]),
    "osz_chain_zp_lag": (c_i64, [c_vp, c_vp]),
    "osz_chain_zp_tolerance": (ctypes.c_int, [c_vp, c_vp, ctypes.c_double]),
    "osz_chain_zp_reach": (ctypes.c_int, [c_vp, c_vp, c_i64]),
    "osz_chain_zp_min_chunk": (c_i64, [c_vp, c_vp]),
    "osz_chain_zp_open": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "osz_chain_zp_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    "osz_chain_zp_seal": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp]),
    "osz_chain_zp_finish": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp]),
    "osz_sos_warm_len": (c_i64, [c_vp]),
    "osz_fir_create": (ctypes.c_int, [ctypes.POINTER(c_vp), c_dp, ctypes.c_int,
                                      ctypes.c_int]),
    "osz_fir_destroy": (ctypes.c_int, [c_vp]),
    "osz_fir_reset": (ctypes.c_int, [c_vp, c_vp]),
    "osz_fir_state_size": (c_i64, [c_vp]),
    "osz_fir_get_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_fir_set_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_fir_push": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64,
                                    c_i64, c_vp]),
    "osz_fir_flush": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "osz_poly_create": (ctypes.c_int, [ctypes.POINTER(c_vp), c_dp, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int]),
    "osz_poly_create_centred": (ctypes.c_int, [ctypes.POINTER(c_vp), c_dp, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "osz_poly_destroy": (ctypes.c_int, [c_vp]),
    "osz_poly_reset": (ctypes.c_int, [c_vp, c_vp]),
    "osz_poly_state_size": (c_i64, [c_vp]),
    "osz_poly_get_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_poly_set_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_poly_out_count": (c_i64, [c_vp, c_i64, ctypes.c_int]),
    "osz_poly_push": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.c_int,
                                     c_vp, c_i64, ctypes.POINTER(c_i64), c_vp]),
    "osz_spec_create": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, c_dp,
                                       ctypes.c_double, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int]),
    "osz_spec_destroy": (ctypes.c_int, [c_vp]),
    "osz_spec_reset": (ctypes.c_int, [c_vp, c_vp]),
    "osz_spec_seg_count": (c_i64, [c_vp, c_i64]),
    "osz_spec_push": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp,
                                     ctypes.POINTER(c_i64), c_vp]),
    "osz_spec_sum": (ctypes.c_int, [c_vp, ctypes.POINTER(c_vp),
                                    ctypes.POINTER(c_i64)]),
    "osz_spec_export_sum": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_i64), c_vp]),
    "osz_spec_mean": (ctypes.c_int, [c_vp, c_dp, ctypes.POINTER(c_i64), c_vp]),
    "osz_spec_mean_device": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_i64), c_vp]),
    "osz_spec_state_size": (c_i64, [c_vp]),
    "osz_spec_get_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_spec_set_state": (ctypes.c_int, [c_vp, c_dp, c_vp]),
    "osz_rccl_bind": (ctypes.c_int, [ctypes.c_char_p]),
    "osz_rccl_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "osz_rccl_comm_create": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_int, ctypes.c_int,
                                            ctypes.c_char_p]),
    "osz_rccl_comm_destroy": (ctypes.c_int, [c_vp]),
    "osz_rccl_comm_size": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_int)]),
    "osz_welch_reduce": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "osz_moments_create": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_int]),
    "osz_moments_destroy": (ctypes.c_int, [c_vp]),
    "osz_moments_reset": (ctypes.c_int, [c_vp, c_vp]),
    "osz_moments_push": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.c_int, c_vp]),
    "osz_moments_finish": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "osz_col_moments": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, ctypes.c_int, c_vp, c_vp,
                                       c_vp]),
    "osz_ew": (ctypes.c_int, [ctypes.c_int, c_vp, c_i64, ctypes.c_int, c_i64, c_vp, c_vp,
                              ctypes.c_int, c_i64, c_vp, c_i64, c_vp]),
    "osz_complex_join": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, ctypes.c_int, c_i64, c_vp,
                                        c_i64, c_vp]),
    "osz_magphase": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "osz_simpson": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_i64, ctypes.c_double,
                                   c_vp, c_vp]),
    "osz_host_copy2d": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "osz_take": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_i64, c_vp,
                                c_i64, c_vp]),
    "osz_edf_decode": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp,
                                      c_vp, c_vp, c_i64, c_i64, c_i64, ctypes.c_double,
                                      c_vp, c_i64, c_vp]),
    "osz_synth_normal": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64,
                                        ctypes.c_uint64, c_i64, c_i64, c_vp]),
    "osz_checksum": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64,
                                    ctypes.POINTER(ctypes.c_uint64),
                                    ctypes.POINTER(ctypes.c_double), c_vp]),
}

_lib = None


def build(verbose=False):
    """Compile libosz_hip.so for gfx950 with hipcc (make in csrc/)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode:
        raise OszLibraryError("building libosz_hip.so failed")


def load():
    """Returns the loaded library with argtypes set; raises OszLibraryError
    if it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OszLibraryError(
            f"{LIB_PATH} not found: build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc, gfx950). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise OszLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


_EXC = {OSZ_ERR_INVALID: ValueError, OSZ_ERR_HIP: RuntimeError,
        OSZ_ERR_NOMEM: MemoryError, OSZ_ERR_STATE: RuntimeError,
        OSZ_ERR_UNSUPPORTED: NotImplementedError}


def check(status):
    """Maps a C status code back to the Python exception types the reference
    raises at this boundary (SURVEY 8b 'Errors')."""
    if status == OSZ_OK:
        return
    msg = load().osz_last_error().decode("utf-8", "replace")
    raise _EXC.get(status, RuntimeError)(msg)
