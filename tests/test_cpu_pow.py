"""x^y of the power-law family (test/ileqg_test.jl:151-155): the reference's Julia evaluates Float64 ^ Float64 with openlibm's pow --
fdlibm's e_pow.c -- which is neither the host libm's (glibc) nor the device math library's.  Both sides restate that published
algorithm: oracle/fdlibm_pow.h (checker) and ratilqr.jl_amd/csrc/rat_pow.h (product, compiled for the device by hipcc).  Here, on
the CPU: the two restatements agree bit for bit, each stays within 1 ulp of glibc (two different < 1 ulp algorithms), the IEEE
special cases are glibc's, and the published constants are self-consistent (hexadecimal words == decimal literals)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def product_pow(tmp_path_factory):
    """csrc/rat_pow.h compiled for the host by gcc (same flags as the oracle: no contraction), as a tiny shared object."""
    d = tmp_path_factory.mktemp("ratpow")
    src = d / "p.c"
    src.write_text('#include "%s"\nvoid rat_pow_array(const double *x, const double *y, long n, double *o) '
                   '{ for (long i = 0; i < n; ++i) o[i] = rat_pow(x[i], y[i]); }\n' % os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "rat_pow.h"))
    so = d / "p.so"
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", str(so), str(src), "-lm"])
    lib = C.CDLL(str(so))

    def f(x, y):
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        out = np.empty_like(x)
        dp = C.POINTER(C.c_double)
        lib.rat_pow_array(x.ctypes.data_as(dp), y.ctypes.data_as(dp), C.c_long(x.size), out.ctypes.data_as(dp))
        return out
    return f


def oracle_pow(x, y):
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    orc.lib().orc_pow_array(x.ctypes.data_as(dp), y.ctypes.data_as(dp), C.c_long(x.size), out.ctypes.data_as(dp))
    return out


def arguments(n, seed):
    rng = np.random.default_rng(seed)
    xs = [rng.uniform(0, 3, n), np.exp(rng.uniform(-20, 20, n)), rng.uniform(0, 2, n), 1.0 + rng.uniform(-5e-4, 5e-4, n), np.exp(rng.uniform(-700, 700, n)),
          -rng.uniform(0, 50, n)]
    ys = [rng.uniform(0.1, 3.1, n), rng.uniform(-10, 10, n), rng.choice([1.3, 1.5, 2.5, 0.3, 0.5, -0.7, 2.0, 3.0], n), rng.uniform(-5e5, 5e5, n),
          rng.uniform(-2, 2, n), rng.choice([1.0, 2.0, 3.0, -3.0, 7.0, 0.5, 1.3], n)]
    return np.concatenate(xs), np.concatenate(ys)


def ulps(a, b):
    ia, ib = a.view(np.int64), b.view(np.int64)
    d = np.abs(ia - ib)
    d[(a == b) | (np.isnan(a) & np.isnan(b))] = 0
    return d


def test_the_two_restatements_agree_bit_for_bit(product_pow):
    x, y = arguments(200_000, 1)
    a, b = oracle_pow(x, y), product_pow(x, y)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    assert np.array_equal(a[ok].view(np.int64), b[ok].view(np.int64))


def test_within_one_ulp_of_the_host_libm_and_not_identical_to_it():
    x, y = arguments(200_000, 2)
    with np.errstate(all="ignore"):
        ref = np.power(x, y)
    a = oracle_pow(x, y)
    assert np.array_equal(np.isnan(a), np.isnan(ref))
    fin = np.isfinite(ref) & (ref != 0)
    d = ulps(a[fin], ref[fin])
    assert d.max() <= 1, (d.max(), x[fin][d.argmax()], y[fin][d.argmax()])
    same = float(np.mean(d == 0))
    assert 0.85 < same < 0.999, same                # fdlibm (< 0.70 ulp) and glibc (< 0.52 ulp) are different algorithms: they do differ


def test_ieee_special_cases():
    sx = np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-310, -8.0, 3.0, 1e300, 1e-300, 4.0, 0.25])
    sy = np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 3.0, -3.0, 1.0 / 3.0, 1e10, -1e10, 1e20, 1074.0, -1074.0, 2.5, 1.3])
    X, Y = np.meshgrid(sx, sy)
    X, Y = X.ravel(), Y.ravel()
    with np.errstate(all="ignore"):
        ref = np.power(X, Y)
    a = oracle_pow(X, Y)
    assert np.array_equal(np.isnan(a), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(np.signbit(a[ok]), np.signbit(ref[ok]))
    assert ulps(a[ok], ref[ok]).max() <= 1
    assert oracle_pow([-8.0], [1.0 / 3.0])[0] != oracle_pow([-8.0], [1.0 / 3.0])[0]      # (x < 0)^non-integer: NaN -> DomainError in Julia


def test_published_constants_hex_equals_decimal():
    """fdlibm prints every constant twice -- decimal literal and hexadecimal words; a slip in either would show here."""
    table = {"dp_h": (0x3FE2B803, 0x40000000, 5.84962487220764160156e-01), "dp_l": (0x3E4CFDEB, 0x43CFD006, 1.35003920212974897128e-08),
             "L1": (0x3FE33333, 0x33333303, 5.99999999999994648725e-01), "L2": (0x3FDB6DB6, 0xDB6FABFF, 4.28571428578550184252e-01),
             "L3": (0x3FD55555, 0x518F264D, 3.33333329818377432918e-01), "L4": (0x3FD17460, 0xA91D4101, 2.72728123808534006489e-01),
             "L5": (0x3FCD864A, 0x93C9DB65, 2.30660745775561754067e-01), "L6": (0x3FCA7E28, 0x4A454EEF, 2.06975017800338417784e-01),
             "P1": (0x3FC55555, 0x5555553E, 1.66666666666666019037e-01), "P2": (0xBF66C16C, 0x16BEBD93, -2.77777777770155933842e-03),
             "P3": (0x3F11566A, 0xAF25DE2C, 6.61375632143793436117e-05), "P4": (0xBEBBBD41, 0xC5D26BF1, -1.65339022054652515390e-06),
             "P5": (0x3E663769, 0x72BEA4D0, 4.13813679705723846039e-08), "lg2": (0x3FE62E42, 0xFEFA39EF, 6.93147180559945286227e-01),
             "lg2_h": (0x3FE62E43, 0x00000000, 6.93147182464599609375e-01), "lg2_l": (0xBE205C61, 0x0CA86C39, -1.90465429995776804525e-09),
             "cp": (0x3FEEC709, 0xDC3A03FD, 9.61796693925975554329e-01), "cp_h": (0x3FEEC709, 0xE0000000, 9.61796700954437255859e-01),
             "cp_l": (0xBE3E2FE0, 0x145B01F5, -7.02846165095275826516e-09), "ivln2": (0x3FF71547, 0x652B82FE, 1.44269504088896338700e+00),
             "ivln2_h": (0x3FF71547, 0x60000000, 1.44269502162933349609e+00), "ivln2_l": (0x3E54AE0B, 0xF85DDF44, 1.92596299112661746887e-08)}
    for name, (hi, lo, dec) in table.items():
        assert struct.unpack(">d", struct.pack(">II", hi, lo))[0] == dec, name
    import re
    for f in ("oracle/fdlibm_pow.h", "ratilqr.jl_amd/csrc/rat_pow.h"):          # every decimal literal of the table appears in both files
        lits = {float(m) for m in re.findall(r"-?\d\.\d{15,}e[-+]\d+", open(os.path.join(ROOT, f)).read())}
        for name, (_, _, dec) in table.items():
            assert dec in lits, (f, name)


def test_oracle_power_law_model_uses_it():
    """x_{t+1} = x^1.3 + u^1.5 through the oracle's simulate_dynamics equals the elementwise orc_pow (not numpy's power)."""
    from ratilqr.jl_amd.problems import PowerLawRiskSensitiveProblem
    prob = PowerLawRiskSensitiveProblem(n=2, N=10, W=0.01 * np.eye(2))
    x0, u = np.array([0.37, 1.91]), 0.1 * np.ones((10, 2)) + 0.01 * np.arange(20).reshape(10, 2)
    rc, x = orc.simulate_open(orc.Problem(prob), x0, u)
    assert rc == 0
    xx = x0.copy()
    for t in range(10):
        xx = oracle_pow(xx, np.full(2, 1.3)) + oracle_pow(u[t], np.full(2, 1.5))
        assert np.array_equal(xx, x[t + 1])
