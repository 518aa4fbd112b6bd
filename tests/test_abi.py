"""CPU suite, part 2: the C-ABI boundary without a GPU -- the library loads, exports every
symbol include/cusmc_hip.h declares, fails loudly without a device, and never routes through
the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cusmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cusmc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared_symbols()
    for must in ("cusmc_dist_pdf_dev", "cusmc_dist_reweight_dev", "cusmc_metropolis_dev",
                 "cusmc_propagate_dev", "cusmc_pf_run_host", "cusmc_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from cusmc_amd import _lib
    L = _lib.lib()  # raises if libcusmc_hip.so is missing or a bound symbol is absent
    for name in _declared_symbols():
        assert hasattr(L, name), "libcusmc_hip.so does not export %s" % name
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert bound == set(_declared_symbols())  # the ctypes table and the header agree


def test_exports_are_exactly_the_abi():
    from cusmc_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    leaked = {s for s in exported if not s.startswith("cusmc_") and not s.startswith("_")}
    assert not leaked, leaked
    assert {s for s in exported if s.startswith("cusmc_")} == set(_declared_symbols())


def test_version_and_error_strings():
    from cusmc_amd import _lib
    L = _lib.lib()
    assert b"gfx950" in L.cusmc_version()
    assert isinstance(L.cusmc_last_error(), bytes)


def _has_gpu():
    from cusmc_amd import _lib
    return _lib.lib().cusmc_device_count() > 0


def test_no_device_fails_loudly_not_silently():
    """On a box without a GPU every compute entry point must refuse -- there is no CPU path."""
    if _has_gpu():
        pytest.skip("a GPU is present")
    import cusmc_amd
    from cusmc_amd import _lib
    with pytest.raises(cusmc_amd.CusmcError) as e:
        cusmc_amd.Context()
    assert e.value.code == _lib.ENODEVICE
    with pytest.raises(cusmc_amd.CusmcError):
        cusmc_amd.MVNPDF([0.0, 0.0], [0.0, 0.0], np.eye(2))
    with pytest.raises(cusmc_amd.CusmcError):
        cusmc_amd.metropolis_hastings([0.0, 0.0], 2, 10)
    I = np.eye(2)
    with pytest.raises(cusmc_amd.CusmcError) as e:  # the multi-device entry validates, then refuses too
        cusmc_amd.run(8, 2, 3, np.zeros((2, 3)), np.zeros(2), I, I, I, I, I, 0.0, "metropolis", "mvn", seed=1, devices=[0, 0])
    assert e.value.code == _lib.ENODEVICE
    with pytest.raises(cusmc_amd.CusmcError) as e:
        cusmc_amd.run(8, 2, 3, np.zeros((2, 3)), np.zeros(2), I, I, I, I, I, 0.0, "bootstrap", "mvn", seed=1, devices=[0, 0])
    assert e.value.code == _lib.EINVAL and "resampler" in str(e.value)


def test_null_arguments_are_rejected_without_a_device():
    from cusmc_amd import _lib
    L = _lib.lib()
    assert L.cusmc_ctx_create(0, None) == _lib.EINVAL
    assert L.cusmc_dist_create(None, 0, None, None, 2, C.c_float(0), None) == _lib.EINVAL
    assert L.cusmc_eigen_sqrt(None, 2, None) == _lib.EINVAL
    assert b"null" in L.cusmc_last_error() or b"bad" in L.cusmc_last_error()


def test_eigen_sqrt_is_host_side():
    """eigenSolver (src/linear_algebra.cpp:10-23) is host code and works without a device."""
    import cusmc_amd
    rng = np.random.default_rng(0)
    A = rng.standard_normal((7, 7))
    S = A @ A.T + np.eye(7)
    Q = cusmc_amd.eigenSolver(S)
    assert np.allclose(Q @ Q.T, S, atol=1e-11)


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or name anything under oracle/."""
    pkg = os.path.join(ROOT, "cusmc_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line
                    assert "libcusmc_oracle" not in code, (f, line)
                    assert not re.search(r"^\s*(from|import)\s+oracle", code), (f, line)
    out = subprocess.run(["ldd", os.path.join(pkg, "libcusmc_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_writeOutput_layout(tmp_path):
    """src/io.cpp:17-32: headers `y` / `w,x`, trailing commas in y_t.csv, w_t[i][0] then x[p]."""
    from cusmc_amd.io import writeOutput
    T, N, d, p = 3, 4, 2, 1
    y = np.arange(T * d, dtype=float).reshape(T, d)
    w = np.arange(T * N, dtype=float).reshape(T, N) / 10
    X = np.arange(T * N * d, dtype=float).reshape(T, N, d)
    writeOutput(y, w, X, N, d, T, p, directory=str(tmp_path))
    assert open(tmp_path / "y_t.csv").read().splitlines() == ["y", "0,1,", "2,3,", "4,5,"]
    assert open(tmp_path / "x_t_N1.csv").read().splitlines() == ["w,x", "0,2,3", "0.4,10,11", "0.8,18,19"]


def test_stream_key_is_splitmix64():
    """cusmc_stream_key(seed, call) = output call+1 of SplitMix64(seed): the published test vector for
    seed 0 (Vigna's splitmix64.c), and distinct keys across calls and seeds."""
    from cusmc_amd import _lib
    k = _lib.lib().cusmc_stream_key
    assert [k(0, c) for c in range(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    keys = {k(s, c) for s in (0, 1, 2 ** 63, 2 ** 64 - 1) for c in range(1000)}
    assert len(keys) == 4000


def test_device_wrappers_validate_before_the_abi():
    """the device-tensor wrappers must refuse wrong dtypes / shapes / strides with a
    ValueError instead of handing raw pointers to the kernels (CPU tensors stand in here: the
    checks run before any pointer is taken)."""
    import torch
    from cusmc_amd import api
    x = torch.zeros(8, 4, dtype=torch.float64)
    a = torch.zeros(8, dtype=torch.int32)
    for call in (lambda: api.propagate_dev(x, a, np.eye(4), np.eye(4), x.clone(), ctx=object()),
                 lambda: api.initialize_dev(np.zeros(4), np.eye(4), x, ctx=object()),
                 lambda: api.Sampler.metropolis_hastings_dev(x[:, 0], a, ctx=object())):
        with pytest.raises(ValueError):
            call()
