"""The Box-Muller transform of the device noise generators (csrc/rat_normal.h: ln, sqrt, sin / cos of pi t written out for their argument
ranges) evaluated on the host (oracle/normal_check.c includes the very header) against libm / numpy: each part to a few ulp, the
normals to 4e-16 of the radius, at random 53-bit uniforms and at the edges of every reduction."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib():
    so = os.path.join(ROOT, "oracle", "libnormal_check.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libnormal_check.so"])
    return C.CDLL(so)


def parts(u1, u2):
    P = C.POINTER(C.c_double)
    out = [np.zeros(u1.size) for _ in range(6)]
    lib().orc_normal_parts(u1.ctypes.data_as(P), u2.ctypes.data_as(P), C.c_long(u1.size), *[o.ctypes.data_as(P) for o in out])
    return out


def test_box_muller_parts_against_libm():
    rng = np.random.default_rng(3)
    n = 400_000
    u1 = rng.integers(0, 2 ** 53, n).astype(np.float64) * 2.0 ** -53
    u2 = rng.integers(0, 2 ** 53, n).astype(np.float64) * 2.0 ** -53
    u1[:9] = [0.0, 1 - 2.0 ** -53, 0.5, 2.0 ** -53, 0.25, 0.75, 1 - 2.0 ** -30, 0.2928932188134524, 0.2928932188134525]   # 1 - u1 around 1, 2^-53, sqrt(1/2)
    u2[:12] = [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1 - 2.0 ** -53, 0.12499999999999999, 0.1250000000000001, 0.62499999999999989]
    lg, rt, sn, cs, z0, z1 = parts(u1, u2)
    ref_l = np.log(1.0 - u1)
    assert np.all(np.isfinite(lg)) and lg[0] == 0.0 and np.all(lg <= 0.0)
    assert np.max(np.abs(lg - ref_l) / np.maximum(np.spacing(np.abs(ref_l)), 5e-324)) <= 2.0
    assert np.max(np.abs(rt - np.sqrt(-2.0 * lg)) / np.spacing(np.maximum(rt, 1e-300))) <= 1.0 and rt[0] == 0.0
    # sin / cos of 2 pi u2: numpy after the same exact reduction (2 u2 = q / 2 + r, |r| <= 1/4)
    t = 2.0 * u2
    q = np.rint(2.0 * t)
    r = t - 0.5 * q
    sp, cp = np.sin(np.pi * r), np.cos(np.pi * r)
    k = q.astype(int) & 3
    ref_s = np.where(k == 0, sp, np.where(k == 1, cp, np.where(k == 2, -sp, -cp)))
    ref_c = np.where(k == 0, cp, np.where(k == 1, -sp, np.where(k == 2, -cp, sp)))
    assert np.max(np.abs(sn - ref_s)) <= 4e-16 and np.max(np.abs(cs - ref_c)) <= 4e-16
    assert np.max(np.abs(sn * sn + cs * cs - 1.0)) <= 5e-16
    assert (sn[0], cs[0]) == (0.0, 1.0) and (sn[2], abs(cs[2])) == (1.0, 0.0) and (abs(sn[4]), cs[4]) == (0.0, -1.0)
    rad = np.sqrt(-2.0 * ref_l)
    assert np.max(np.abs(z0 - rad * ref_c) - 6e-16 * rad) <= 0 and np.max(np.abs(z1 - rad * ref_s) - 6e-16 * rad) <= 0


def test_box_muller_moments():
    rng = np.random.default_rng(4)
    n = 1_000_000
    u1 = rng.integers(0, 2 ** 53, n).astype(np.float64) * 2.0 ** -53
    u2 = rng.integers(0, 2 ** 53, n).astype(np.float64) * 2.0 ** -53
    _, _, _, _, z0, z1 = parts(u1, u2)
    z = np.concatenate([z0, z1])
    assert abs(z.mean()) < 4 / np.sqrt(z.size) and abs(z.var() - 1) < 6 * np.sqrt(2 / z.size)
    assert abs(np.mean(z ** 4) - 3) < 0.05 and abs(np.mean(z0 * z1)) < 4 / np.sqrt(n)
