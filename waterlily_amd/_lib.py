"""ctypes binding of libwlhip.so (the C ABI declared in include/wlhip.h).

There is NO fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WLHIP_LIB") or os.path.join(_HERE, "libwlhip.so")   # (WLHIP_LIB: an alternative build, A/B measurements)
HEADER = os.path.join(os.path.dirname(_HERE), "include", "wlhip.h")
_lib = None

WL_F32, WL_F64 = 0, 1
WL_E_ARG, WL_E_LEVELS, WL_E_NOGPU, WL_E_STATE = 10001, 10002, 10003, 10004


class WlError(RuntimeError):
    pass


class Grid(C.Structure):
    _fields_ = [("D", C.c_int32), ("n", C.c_int32 * 3), ("s", C.c_int64 * 3), ("sc", C.c_int64),
                ("nzg", C.c_int32), ("kz0", C.c_int32), ("own_lo", C.c_int32), ("own_hi", C.c_int32),
                ("zring", C.c_int32)]


class LevelDesc(C.Structure):
    _fields_ = [("g", Grid)] + [(k, C.c_void_p) for k in ("L", "D", "iD", "x", "eps", "r", "z")]


class FlowDesc(C.Structure):
    _fields_ = [("g", Grid)] + [(k, C.c_void_p) for k in ("u", "u0", "f", "p", "sigma", "V", "mu0", "mu1")] + [
        ("nu", C.c_double), ("exitBC", C.c_int32), ("perdir_mask", C.c_int32)]


class BodyDesc(C.Structure):
    """wl_body_desc (include/wlhip.h): a parametric body at one instant"""
    _fields_ = [("family", C.c_int32), ("identity_map", C.c_int32), ("p", C.c_double * 8), ("A", C.c_double * 9),
                ("b", C.c_double * 3), ("dA", C.c_double * 9), ("db", C.c_double * 3), ("Ainv", C.c_double * 9),
                ("op", C.c_int32), ("count", C.c_int32)]


WL_BODY_SPHERE, WL_BODY_TORUS, WL_BODY_PLATE, WL_BODY_CYLINDER = 0, 1, 2, 3
WL_BODY_OP_UNION, WL_BODY_OP_MINUS, WL_BODY_OP_INTERSECT = 0, 1, 2
WL_BODY_MAXLEAF = 6

SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)


def build(force: bool = False) -> str:
    """Compile waterlily_amd/libwlhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [HEADER]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-C", csrc], capture_output=True, text=True)
        if r.returncode != 0:
            raise WlError("building libwlhip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def declared_symbols() -> list[str]:
    """Every function the public header declares (used by the CPU-side export test)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wl_[a-zA-Z0-9_]+)\s*\(", txt)))


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WlError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(waterlily_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, gp, dp, i, d, i64 = C.c_void_p, C.POINTER(Grid), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int64
    ip = C.POINTER(C.c_int)
    sig = {
        "wl_abi_version": (i, []),
        "wl_last_error": (C.c_char_p, []),
        "wl_device_count": (i, [ip]),
        "wl_set_device": (i, [i]),
        "wl_set_stream": (i, [vp]),
        "wl_sync": (i, []),
        "wl_malloc": (i, [C.POINTER(vp), C.c_size_t]),
        "wl_free": (i, [vp]),
        "wl_h2d": (i, [vp, vp, C.c_size_t]),
        "wl_d2h": (i, [vp, vp, C.c_size_t]),
        "wl_memset0": (i, [vp, C.c_size_t]),
        "wl_h2d_2d": (i, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]),
        "wl_d2h_2d": (i, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]),
        "wl_comm_unique_id": (i, [vp]),
        "wl_comm_init_rccl": (i, [vp, i, i]),
        "wl_comm_init_host": (i, [i, i, SENDRECV_FN, ALLREDUCE_FN, ALLGATHER_FN, vp]),
        "wl_comm_init_loopback": (i, [i, i]),
        "wl_comm_mailbox": (i, [C.c_char_p, i]),
        "wl_comm_mailbox_off": (i, []),
        "wl_comm_mailbox_active": (i, [ip]),
        "wl_comm_finalize": (i, []),
        "wl_comm_rank": (i, [ip, ip]),
        "wl_halo_exchange": (i, [i, gp, vp, i, i]),
        "wl_allreduce": (i, [dp, i, i]),
        "wl_bc_vec": (i, [i, gp, vp, dp, i, i]),
        "wl_bc_per": (i, [i, gp, vp, i]),
        "wl_exit_bc": (i, [i, gp, vp, vp, dp, d]),
        "wl_L2_inside": (i, [i, gp, vp, dp]),
        "wl_dot": (i, [i, gp, vp, vp, dp]),
        "wl_sum": (i, [i, gp, vp, dp]),
        "wl_max": (i, [i, gp, vp, dp]),
        "wl_conv_diff": (i, [i, gp, vp, vp, vp, d, i]),
        "wl_accelerate": (i, [i, gp, vp, dp]),
        "wl_bdim": (i, [i, gp, vp, vp, vp, vp, vp, vp, d]),
        "wl_scale_u": (i, [i, gp, vp, d]),
        "wl_div": (i, [i, gp, vp, vp]),
        "wl_cfl": (i, [i, gp, vp, vp, d, dp]),
        "wl_set_diag": (i, [i, gp, vp, vp, vp]),
        "wl_restrictL": (i, [i, gp, vp, gp, vp, i]),
        "wl_restrict": (i, [i, gp, vp, gp, vp]),
        "wl_prolongate": (i, [i, gp, vp, gp, vp]),
        "wl_mg_create": (i, [C.POINTER(vp), i, i, C.POINTER(LevelDesc), i]),
        "wl_mg_destroy": (i, [vp]),
        "wl_mg_update": (i, [vp]),
        "wl_mg_update_changed": (i, [vp, vp]),
        "wl_mg_uniform_rows": (i, [vp, i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        "wl_mg_mult": (i, [vp, i, vp]),
        "wl_mg_residual": (i, [vp, i]),
        "wl_mg_increment": (i, [vp, i]),
        "wl_mg_jacobi": (i, [vp, i, i]),
        "wl_mg_pcg": (i, [vp, i, i, ip]),
        "wl_mg_L2": (i, [vp, i, dp]),
        "wl_mg_Linf": (i, [vp, i, dp]),
        "wl_mg_log": (i, [vp, i]),
        "wl_mg_log_read": (i, [vp, dp, i, ip]),
        "wl_mg_vcycle": (i, [vp, i]),
        "wl_mg_solve": (i, [vp, d, i, ip]),
        "wl_flow_create": (i, [C.POINTER(vp), i, C.POINTER(FlowDesc)]),
        "wl_flow_destroy": (i, [vp]),
        "wl_flow_update": (i, [vp]),
        "wl_measure_rows": (i, [vp, C.POINTER(BodyDesc), d, C.POINTER(i64)]),
        "wl_measure_fill": (i, [vp, C.POINTER(BodyDesc), d, vp]),
        "wl_body_nds": (i, [gp, C.POINTER(BodyDesc), vp, i64, vp]),
        "wl_project": (i, [vp, vp, d, d, ip]),
        "wl_mom_step": (i, [vp, vp, d, dp, dp, dp, dp, ip]),
        "wl_metric": (i, [i, gp, i, vp, vp, i, dp, dp]),
        "wl_pforce": (i, [i, gp, vp, vp, vp, i64, dp]),
        "wl_vforce": (i, [i, gp, vp, vp, vp, i64, d, dp]),
        "wl_pmoment": (i, [i, gp, vp, vp, vp, i64, dp, dp]),
        "wl_snapshot_pack": (i, [i, gp, vp, i, i, i, i, vp]),
        "wl_snapshot_unpack": (i, [i, gp, vp, i, i, i, i, vp]),
        "wl_set_option": (i, [i, i]),
        "wl_get_option": (i, [i, ip]),
        "wl_kernel_name": (C.c_char_p, [i]),
        "wl_prof_select": (i, [i, i64]),
        "wl_prof_reset": (i, []),
        "wl_prof_overlapped": (i, [C.POINTER(i64)]),
        "wl_prof_counts": (i, [i, C.POINTER(i64), C.POINTER(i64)]),
        "wl_prof_allocs": (i, [C.POINTER(i64), C.POINTER(i64)]),
        "wl_prof_comm": (i, [C.POINTER(i64)]),
        "wl_prof_reset_comm": (i, []),
        "wl_prof_allreduce_us": (i, [i, dp]),
        "wl_prof_timed": (i, [C.POINTER(i64), C.POINTER(i64), dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.wl_abi_version() != 6:
        raise WlError("libwlhip.so ABI version mismatch; rebuild it")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().wl_last_error().decode(errors="replace")
        if rc == WL_E_LEVELS:
            raise AssertionError("MultiLevelPoisson requires size=a2ⁿ, where n>2")
        raise WlError(f"libwlhip call failed ({rc}): {msg}")


def d3(v):
    import numpy as np
    v = list(np.asarray(v, dtype=np.float64).ravel()) + [0.0] * 3
    return (C.c_double * 3)(*v[:3])
