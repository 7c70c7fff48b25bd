"""CPU-side checks of the drop-in boundary: libpfmi.so builds for gfx950, loads, and exports every
symbol include/pfmi.h declares; and the product path fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "pfmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pfmi_[a-z0-9_]+)\s*\(", txt)) - {"pfmi_logp_fn"})


def test_library_exports_every_declared_symbol():
    import pfmi
    so = pfmi.build()
    lib = ctypes.CDLL(so)
    syms = _header_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    from pfmi._lib import SYMBOLS
    assert sorted(SYMBOLS) == syms
    lib.pfmi_version.restype = ctypes.c_int32
    assert lib.pfmi_version() >= 100


def test_no_cpu_fallback_without_gpu():
    import pfmi
    lib = pfmi.lib()
    n = ctypes.c_int32(-1)
    rc = lib.pfmi_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pfmi.PfmiError):
        pfmi.Engine(0)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "pathfinder.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "pf_oracle" not in src.replace("oracle/pf_oracle.c:pfo_philox4x32_10", ""), os.path.join(dp, f)
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dp, f)
