"""The C-ABI library loads and exports every symbol include/alfd/*.h declares;
no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from fictitious_domain_al_preconditioners_amd import _abi, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    return built


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "alfd", "alfd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(alfd_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    lib = solver.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(solver.ABI_SYMBOLS) == declared     # the Python mirror binds all of them


def test_version_strerror_and_default_config_mirror():
    lib = solver.load_library()
    assert lib.alfd_abi_version() == 5
    assert b"NoConvergence" in lib.alfd_strerror(_abi.E_NO_CONVERGENCE_INNER)
    assert lib.alfd_strerror(_abi.OK) == b"ok"
    for variant in (_abi.AL2, _abi.AL_STOKES, _abi.AL_ELL_MODIFIED, _abi.RATIONAL):
        c = _abi.Config()
        lib.alfd_default_config(C.byref(c), variant)
        py = _abi.default_config(variant)
        assert bytes(c) == bytes(py)                  # same layout, same defaults
    # reference defaults: restart 30 (50 for elliptic_interface.cc:863), inner 100 / 1e-2 abs
    c = _abi.default_config(_abi.AL_STOKES)
    assert (c.restart, c.inner.max_steps, c.inner.tol, c.inner.kind) == (30, 100, 1e-2, _abi.CTRL_ABS)
    assert _abi.default_config(_abi.AL_ELL_MODIFIED).restart == 50
    assert C.sizeof(_abi.Config) == 216 and C.sizeof(_abi.Result) == 72
    assert C.sizeof(_abi.MatrixInfo) == 88


def test_argument_validation_without_gpu():
    lib = solver.load_library()
    assert lib.alfd_destroy(None) == _abi.E_INVALID
    assert lib.alfd_setup(None) == _abi.E_INVALID
    assert lib.alfd_comm_unique_id(None, 0) == _abi.E_INVALID
    assert lib.alfd_last_error(None) == b"null context"


def test_no_cpu_fallback():
    """Without a HIP device the product path fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(solver.AlfdError) as e:
        solver.Context(0)
    assert e.value.status == _abi.E_HIP
    assert "no CPU path" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fictitious_domain_al_preconditioners_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                # no import, dlopen, include or path of anything under oracle/
                hits = re.findall(r"(import\s+oracle|from\s+oracle|liboracle|oracle/|orc_[a-z_]+\s*\()", txt)
                assert not hits, (os.path.join(dirpath, f), hits)
