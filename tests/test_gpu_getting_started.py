"""The reference's own getting-started example (docs/source/getting-started.md:40-118): a 2-D single integrator x' = x + dt u with cost
1/2 x'Qx + 1/2 u'Ru, terminal cost 1/2 x'Qx, W = 0.1 dt I, N = 10, solved by CrossEntropyBilevelOptimizationSolver() with its DEFAULT
parameters from x_0 = [5, 5], u = 0, kl_bound = 0.1 -- here as an LQ-family problem on the device, with the solver's N(0,1) draws injected
so that the oracle (the restated reference) can be run on the same stream.  What a first-time user of the reference runs must work and agree."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import cross_entropy as ce
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def single_integrator(dt=0.1, N=10):
    Q, R = np.eye(2), 0.01 * np.eye(2)
    prob = rat.LQRiskSensitiveProblem(np.eye(2), dt * np.eye(2), Q=Q, R=R, N=N, W=0.1 * dt * np.eye(2), Qf=Q)
    return prob, np.array([5.0, 5.0]), np.zeros((N, 2))


@pytest.mark.parametrize("devices", [1, 3])
def test_getting_started_example(devices, monkeypatch):
    prob, x0, u = single_integrator()
    z = np.random.default_rng(12345).standard_normal(20000)
    solver = rat.CrossEntropyBilevelOptimizationSolver()              # defaults: num_samples 10, num_elite 3, iter_max 5   (:70-127)
    th, x, l, L, value, tmin, tmax = ce.solve_(solver, prob, x0, u, z, kl_bound=0.1)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z)
    rc, th_o, x_o, l_o, L_o, val_o, tmin_o, tmax_o = oc.solve(orc.Problem(prob), x0, u, 0.1)
    assert rc == 0
    assert abs(th - th_o) <= 1e-9 * abs(th_o) and abs(value - val_o) <= 1e-9 * abs(val_o) and tmin == tmin_o and tmax == tmax_o
    assert np.abs(x - x_o).max() < 1e-9 and np.abs(l - l_o).max() < 1e-9 and np.abs(L - L_o).max() < 1e-9
    # the policy is what the documentation promises: it steers the state to the origin and feeds back on the state
    assert np.linalg.norm(x[-1]) < 0.2 * np.linalg.norm(x0) and th > 0 and tmin <= th <= tmax
    assert np.all(np.linalg.norm(x[1:], axis=1) < np.linalg.norm(x[:-1], axis=1)) and np.abs(L).max() > 0.1
    if devices > 1:
        # the same solve with the CE batches (10 samples: ragged over 3 devices) sharded behind the C ABI
        import ctypes as C
        from ratilqr.jl_amd import _native as nv
        monkeypatch.setenv("RATILQR_MULTI_LOGICAL", "1")
        mc = rat.MultiContext(prob, max_batch=10, devices=tuple(range(devices)))
        c = nv.CeSolver()
        nv.lib().rat_ce_default(C.byref(c))
        mc.set_stream(z)
        got = mc.ce_solve(c, x0, u, 0.1)
        assert got[0] == th and got[4] == value and np.array_equal(got[1], x) and np.array_equal(got[3], L) and mc.allgathers == c.n_solves // 10
