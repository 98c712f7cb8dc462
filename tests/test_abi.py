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
    assert L.wl_abi_version() == 5
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


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from waterlily_amd import sim
    with pytest.raises(_lib.WlError, match="no CPU fallback"):
        sim.Flow((16, 16), (1.0, 0.0))
