"""The priority transform's power, `np.float32(x) ** alpha` (reference
pfrl/replay_buffers/prioritized.py:47-55), is libm's powf.  csrc/powf_glibc.h restates
glibc's algorithm so that the device computes the same bits; here the restatement is pinned
against this host's libm / NumPy (CPU) and the device kernel against the restatement (GPU)."""
import ctypes
import ctypes.util

import numpy as np
import pytest

from pfrl_amd import ops


def _libm_powf():
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    return libm.powf


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("alpha", [0.5, 0.6, 0.7, 0.4, 1.0])
def test_host_probe_finds_a_variant(alpha):
    assert ops.powf_host_variant(alpha) in (ops.POW_GLIBC, ops.POW_GLIBC_FMA)


@pytest.mark.parametrize("mode", [ops.POW_GLIBC, ops.POW_GLIBC_FMA])
def test_restatement_equals_numpy_scalar_power(mode):
    """10^6 priorities' worth of (clip(err) + eps) in (0.01, 1.01], alpha 0.5 / 0.6: the values
    NumPy's scalar power gives (the reference's expression), 0 ulp."""
    rs = np.random.RandomState(0)
    for alpha in (0.5, 0.6):
        x = (rs.rand(500000).astype(np.float32) + np.float32(0.01)).astype(np.float32)
        want = np.array([v ** alpha for v in x], dtype=np.float32)   # np.float32 ** float
        assert want.dtype == np.float32
        got = ops.powf_host(x, alpha, mode)
        assert np.array_equal(_bits(got), _bits(want))


def test_restatement_special_cases_equal_libm():
    powf = _libm_powf()
    xs = [0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 1e-45,
          3.4e38, 1.0000001, 0.99999994, 7.0, -7.0, 1e-20, 1e20]
    ys = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, 3.0, -3.0, 0.6, np.inf, -np.inf, np.nan, 1e-40,
          100.0, -100.0, 4e9, 1 / 3]
    for y in ys:
        x = np.array(xs, dtype=np.float32)
        want = np.array([powf(float(v), float(np.float32(y))) for v in x], dtype=np.float32)
        for mode in (ops.POW_GLIBC, ops.POW_GLIBC_FMA):
            got = ops.powf_host(x, y, mode)
            same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
            assert same.all(), (y, x[~same], got[~same], want[~same])


@pytest.mark.gpu
@pytest.mark.parametrize("alpha", [0.5, 0.6])
def test_device_power_is_the_hosts_powf_0_ulp(alpha):
    """>= 10^7 inputs over the transform's whole domain [2^-7, 2): the device evaluates the
    number this host's libm (= NumPy) gives, bit for bit, in the mode the buffers use."""
    import torch

    dev = torch.device("cuda:0")
    mode = ops.powf_host_variant(alpha)
    assert mode is not None
    powf = _libm_powf()
    lo, hi = 0x3c000000, 0x40000000
    rs = np.random.RandomState(1)
    bits = rs.randint(lo, hi, size=1 << 24).astype(np.uint32)
    bits[:4096] = np.arange(lo, lo + 4096, dtype=np.uint32)
    x = bits.view(np.float32)
    got = ops.powf_device(torch.from_numpy(x).to(dev), alpha, mode).cpu().numpy()
    want = ops.powf_host(x, alpha, mode)
    assert np.array_equal(_bits(got), _bits(want))
    # and the host restatement against libm itself on a slice of the same inputs
    idx = rs.randint(0, x.size, size=20000)
    ref = np.array([powf(float(v), float(np.float32(alpha))) for v in x[idx]], dtype=np.float32)
    assert np.array_equal(_bits(got[idx]), _bits(ref))
    # the other build of the same code differs, if at all, by <= 1 ulp
    other = ops.POW_GLIBC if mode == ops.POW_GLIBC_FMA else ops.POW_GLIBC_FMA
    got2 = ops.powf_device(torch.from_numpy(x).to(dev), alpha, other).cpu().numpy()
    d = np.abs(_bits(got2).astype(np.int64) - _bits(got).astype(np.int64))
    assert d.max() <= 1
