"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/egx_gp.h declares, its host-only helpers match the oracle, and it refuses to compute without a
GPU (no fallback).  No device compute is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "egx_gp.h")


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="module")
def egx():
    import egobox_amd
    return egobox_amd


def _declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(egx_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_symbols_are_exported_and_bound(egx):
    declared = _declared_functions()
    assert len(declared) >= 20
    lib = C.CDLL(egx._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/egx_gp.h but not exported"
    bound = {s[0] for s in egx._lib.SIGNATURES}
    assert set(declared) == bound, (set(declared) ^ bound)
    out = subprocess.run(["nm", "-D", "--defined-only", egx._lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (egx_[a-z0-9_]+)", out))
    assert exported == set(declared), (exported ^ set(declared))


def test_abi_version_and_default_config(egx):
    lib = egx._lib.load()
    assert lib.egx_abi_version() == 2
    cfg = egx._lib.GpConfig()
    lib.egx_gp_config_default(C.byref(cfg))
    assert cfg.corr == 0 and cfg.mean == 0 and cfg.n_workspaces == 1 and cfg.device == -1
    assert cfg.nugget == 100.0 * np.finfo(float).eps  # crates/gp/src/parameters.rs:118


def test_host_normalize_matches_oracle(egx):
    from oracle import gp_oracle as O
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 4)) * [1, 10, 0.1, 5] + [0, 3, -2, 100]
    x[:, 2] = 7.0  # zero std column -> std 1 (utils.rs:50)
    xn, m, s = egx.normalize(x)
    xo, mo, so = O.normalize(x)
    np.testing.assert_allclose(xn, xo, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(m, mo, rtol=1e-14)
    np.testing.assert_allclose(s, so, rtol=1e-13)
    assert s[2] == 1.0
    with pytest.raises(egx.InvalidValueError):
        egx.normalize(np.zeros((1, 3)))


@pytest.mark.parametrize("mean,name", [(0, "Constant"), (1, "Linear"), (2, "Quadratic")])
def test_host_regression_basis_matches_oracle(egx, mean, name):
    from oracle import gp_oracle as O
    x = np.random.default_rng(1).standard_normal((7, 3))
    np.testing.assert_array_equal(egx.regression_basis(mean, x), O.regression_value(name, x))
    assert egx._lib.load().egx_regression_ncols(mean, 3) == O.regression_value(name, x).shape[1]


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback(egx):
    assert egx._lib.load().egx_device_count() == 0
    with pytest.raises(egx.NoDeviceError):
        egx.GpHandle(np.random.rand(10, 2), np.random.rand(10))
    with pytest.raises(egx.NoDeviceError):
        egx.corr_matrix(0, np.random.rand(4, 2), [1.0, 1.0])
    with pytest.raises(egx.NoDeviceError):
        egx.potrf(np.eye(3))
    with pytest.raises(egx.NoDeviceError):
        egx.Kriging.params().fit(np.random.rand(10, 2), np.random.rand(10))
    with pytest.raises(egx.NoDeviceError):
        egx.SgpHandle(np.random.rand(10, 2), np.random.rand(10), np.random.rand(3, 2))
    with pytest.raises(egx.InvalidValueError):  # validated before the device is touched
        egx.SgpHandle(np.random.rand(10, 2), np.random.rand(10), np.random.rand(11, 2))
    with pytest.raises(egx.InvalidValueError):
        egx.SgpHandle(np.random.rand(10, 2), np.random.rand(10), np.random.rand(3, 2), method=7)


def test_argument_validation_before_device(egx):
    """Invalid inputs are rejected with InvalidValueError whether or not a GPU is present."""
    good_x, good_y = np.random.rand(10, 2), np.random.rand(10)
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(good_x[:1], good_y[:1])              # n < 2
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(np.zeros((0, 2)), np.zeros(0))       # empty
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(good_x, good_y[:9])                  # ragged
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(np.zeros((10, 0)), good_y)           # d < 1 (round 4: no upper bound on d any more)
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(good_x, good_y, corr=9)
    bad = good_x.copy()
    bad[3, 1] = np.nan
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(bad, good_y)
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(good_x, good_y, w_star=np.ones((2, 3)))  # kpls_dim > d (algorithm.rs:798-807)
    with pytest.raises(egx.InvalidValueError):
        egx.GpHandle(good_x[:3], good_y[:3], mean=2)      # p >= n


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under egobox_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "egobox_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "gp_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_header_is_valid_c_and_cpp(tmp_path):
    """include/egx_gp.h must compile as plain C (the Rust/cgo-style FFI consumers see it as C) and as C++."""
    c = tmp_path / "t.c"
    c.write_text('#include "egx_gp.h"\nint main(void) { egx_gp_config c; egx_timings t; (void)c; (void)t; '
                 'return EGX_GP_ABI_VERSION == 2 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", f"-I{inc}", str(c)], check=True)
    cpp = tmp_path / "t.cpp"
    cpp.write_text(c.read_text())
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", f"-I{inc}", str(cpp)], check=True)
    hpp = tmp_path / "t2.cpp"  # the header-only C++ mirror of the reference's builder and the host tests that use it
    hpp.write_text('#include "egx_gp.hpp"\nint main() { auto p = egobox::Kriging::params(); (void)p; return 0; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", f"-I{inc}", str(hpp)], check=True)
    for src in ("golden_a_driver.c", "sweep_driver.c", "moe_driver.c", "reference_style_tests.cpp"):
        comp = ["gcc", "-std=c99"] if src.endswith(".c") else ["g++", "-std=c++17"]
        subprocess.run(comp + ["-Wall", "-Werror", "-fsyntax-only", f"-I{inc}", os.path.join(ROOT, "tests", "c_host", src)],
                       check=True)


def test_c_program_links_against_the_library(tmp_path):
    """A C program using only the header links against libegx_gp_hip.so and can call the host-only entry points."""
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include "egx_gp.h"
int main(void) {
    double x[4] = {1, 2, 3, 4}, xn[4], m[2], s[2];
    if (egx_abi_version() != EGX_GP_ABI_VERSION) return 1;
    if (egx_normalize(x, 2, 2, xn, m, s) != EGX_SUCCESS) return 2;
    if (m[0] != 2.0 || m[1] != 3.0) return 3;
    if (egx_regression_ncols(EGX_MEAN_QUADRATIC, 3) != 10) return 4;
    if (egx_normalize(x, 1, 2, xn, m, s) != EGX_ERR_INVALID_VALUE) return 5;
    printf("%s\n", egx_last_error());
    return 0;
}
''')
    exe = tmp_path / "host"
    libdir = os.path.join(ROOT, "egobox_amd", "lib")
    subprocess.run(["gcc", "-std=c99", f"-I{os.path.join(ROOT, 'include')}", str(src), f"-L{libdir}", "-legx_gp_hip",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out
    assert "n >= 2" in out.stdout


def test_sweep_argument_validation_needs_no_device(egx):
    """egx_sweep_create rejects an inconsistent (rank, world) / a missing unique id before it touches a device."""
    lib = egx._lib.load()
    x = np.random.default_rng(0).random((10, 2))
    y = x.sum(axis=1)
    h = C.c_void_p()
    dp = egx._lib.dptr
    for rank, world, idp in ((2, 2, None), (-1, 1, None), (0, 0, None), (0, 2, None)):
        rc = lib.egx_sweep_create(None, dp(x), dp(y), 10, 2, idp, rank, world, C.byref(h))
        assert rc == egx._lib.ERR_INVALID_VALUE and not h.value, (rank, world)
    assert b"rank" in lib.egx_last_error() or b"unique id" in lib.egx_last_error()
    assert lib.egx_sweep_info(None, None, None, None, None, None) == egx._lib.ERR_INVALID_VALUE
    lib.egx_sweep_destroy(None)  # no-op
