"""CPU-side checks of the drop-in boundary: libwlhip.so loads without a GPU and exports every symbol that
include/wlhip.h declares; argument validation works without touching the device."""
import ctypes as C

import pytest

from waterlily_amd import _lib


def test_header_symbols_exported():
    names = _lib.declared_symbols()
    assert len(names) >= 45
    L = C.CDLL(_lib.build())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in wlhip.h but not exported: {missing}"


def test_binding_covers_header():
    L = _lib.lib()
    for n in _lib.declared_symbols():
        assert getattr(L, n).argtypes is not None, f"{n} has no ctypes signature in _lib.py"


def test_abi_version_and_names():
    L = _lib.lib()
    assert L.wl_abi_version() == 6
    assert L.wl_kernel_name(14) == b"pcg_mult_dot"


def test_argument_validation_without_gpu():
    L = _lib.lib()
    g = _lib.Grid()
    g.D = 4
    rc = L.wl_bc_per(_lib.WL_F32, C.byref(g), None, 0)
    assert rc != 0 and b"grid.D" in L.wl_last_error()
    h = C.c_void_p()
    lv = (_lib.LevelDesc * 2)()
    assert L.wl_mg_create(C.byref(h), _lib.WL_F32, 2, lv, 0) == _lib.WL_E_LEVELS
    with pytest.raises(AssertionError, match="MultiLevelPoisson requires size=a2ⁿ, where n>2"):
        _lib.check(_lib.WL_E_LEVELS)


def test_new_entry_points_validate_without_gpu():
    """ABI v5 additions reject bad calls before touching the device: the mailbox needs a communicator and a POSIX name,
    the log reader and L-inf need their outputs, composite bodies are bounded."""
    L = _lib.lib()
    assert L.wl_comm_mailbox(b"/wlhip-test", 1) == _lib.WL_E_STATE and b"communicator" in L.wl_last_error()
    on = C.c_int(7)
    assert L.wl_comm_mailbox_active(C.byref(on)) == 0 and on.value == 0
    assert L.wl_comm_mailbox_off() == 0
    v = (C.c_int64 * 6)(*[9] * 6)
    assert L.wl_prof_comm(v) == 0 and list(v) == [0] * 6
    assert L.wl_prof_reset_comm() == 0
    assert L.wl_mg_log(None, 1) != 0 and L.wl_mg_log_read(None, None, 0, None) != 0
    assert L.wl_mg_Linf(None, 0, None) != 0
    assert L.wl_set_option(31, 0) == 0 and L.wl_set_option(31, 1) == 0 and L.wl_set_option(32, 0) != 0
    for retired in (11, 12, 20, 21, 24, 25, 28, 29):       # round 4: keys whose alternative was a recorded loss are gone
        assert L.wl_set_option(retired, 1) == _lib.WL_E_ARG and b"no such option" in L.wl_last_error()
    v = C.c_int()
    assert L.wl_get_option(26, C.byref(v)) == 0 and v.value == 600 and L.wl_get_option(24, C.byref(v)) == _lib.WL_E_ARG
    # body descriptors: family / composite checks happen on the host
    bd = (_lib.BodyDesc * 2)()
    bd[0].family, bd[0].count = 9, 1
    nb = C.c_int64()
    assert L.wl_measure_rows(None, bd, 1.0, C.byref(nb)) != 0          # null flow handle
    g = _lib.Grid()
    g.D = 3
    g.n[:] = [8, 8, 8]
    g.s[:] = [1, 8, 64]
    g.sc = 512
    assert L.wl_body_nds(C.byref(g), bd, None, 4, None) != 0 and b"family" in L.wl_last_error()
    bd[0].family, bd[0].count = _lib.WL_BODY_SPHERE, _lib.WL_BODY_MAXLEAF + 1
    assert L.wl_body_nds(C.byref(g), bd, None, 4, None) != 0 and b"WL_BODY_MAXLEAF" in L.wl_last_error()
    bd[0].count = 2
    bd[1].family, bd[1].op = _lib.WL_BODY_CYLINDER, 5
    assert L.wl_body_nds(C.byref(g), bd, None, 4, None) != 0 and b"operation" in L.wl_last_error()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from waterlily_amd import sim
    with pytest.raises(_lib.WlError, match="no CPU fallback"):
        sim.Flow((16, 16), (1.0, 0.0))


def test_abi_v6_entry_points_validate_without_gpu():
    """ABI v6 additions reject bad calls before touching the device: pitched copies whose row is wider than a pitch, snapshot
    staging with a bad tuple / plane range, the loopback communicator's rank, option queries of retired keys."""
    L = _lib.lib()
    assert L.wl_h2d_2d(None, 8, None, 16, 32, 1) == _lib.WL_E_ARG and b"wider than a pitch" in L.wl_last_error()
    assert L.wl_d2h_2d(None, 64, None, 16, 32, 1) == _lib.WL_E_ARG
    g = _lib.Grid()
    g.D = 3
    g.n[:] = [8, 8, 8]
    g.s[:] = [1, 8, 64]
    g.sc = 512
    buf = (C.c_float * 4)()
    assert L.wl_snapshot_pack(_lib.WL_F32, C.byref(g), buf, 3, 2, 0, 7, buf) == _lib.WL_E_ARG and b"ntuple" in L.wl_last_error()
    assert L.wl_snapshot_pack(_lib.WL_F32, C.byref(g), buf, 3, 3, 2, 9, buf) == _lib.WL_E_ARG and b"plane range" in L.wl_last_error()
    assert L.wl_snapshot_unpack(_lib.WL_F32, C.byref(g), None, 1, 1, 0, 0, buf) == _lib.WL_E_ARG
    assert L.wl_comm_init_loopback(3, 2) == _lib.WL_E_ARG
    n, b = C.c_int64(-1), C.c_int64(-1)
    assert L.wl_prof_allocs(C.byref(n), C.byref(b)) == 0 and n.value >= 0 and b.value >= 0
    assert L.wl_prof_allocs(None, None) == _lib.WL_E_ARG
