"""Pins the oracle's Nelder-Mead (RAT iLQR++) restatement with the reference's known-answer test
(/root/reference/test/nelder_mead_bilevel_optimization_test.jl:10-32, K16) and the stale-state quirk of
nelder_mead_bilevel_optimization.jl:164-168 / 283-304.  CPU only."""
import numpy as np

import ratilqr.jl_amd as rat
from oracle import oracle as orc


def nonlinear():
    prob = rat.PowerLawRiskSensitiveProblem(2, 10, 0.01 * np.eye(2))
    return orc.Problem(prob), np.zeros(2), 0.1 * np.ones((10, 2))


def test_K16_nm_optimum_not_worse_than_initial_vertices():          # nm_test.jl:22-32
    P, x0, u = nonlinear()
    nm = orc.NelderMeadBilevelOptimizationSolver(iter_max=20, eps=1e-3, theta_high_init=10.0, theta_low_init=1e-8)
    rc, th, x, l, L, c_opt = nm.solve(P, x0, u, 1.0)
    assert rc == 0 and np.isfinite(c_opt) and not np.isnan(th)
    c_low_init = nm.compute_cost(P, x0, u, nm.c.theta_low_init, 1.0)
    c_high_init = nm.compute_cost(P, x0, u, nm.c.theta_high_init, 1.0)
    assert np.isfinite(c_low_init) and np.isfinite(c_high_init)
    assert c_opt <= c_low_init and c_opt <= c_high_init


def test_kl_zero_is_ilqg():                                         # nm.jl:330-333, 349-351
    P, x0, u = nonlinear()
    nm = orc.NelderMeadBilevelOptimizationSolver()
    rc, th, x, l, L, val = nm.solve(P, x0, u, 0.0)
    assert rc == 0 and th == 0.0 and np.isclose(val, 1.0029075497782471, rtol=1e-12)
    assert not nm.c.has_c_high and not nm.c.has_c_low


def test_infeasible_theta_high_is_halved_and_init_persists():       # nm.jl:283-293
    prob, x0, u = rat.synthetic_lq_problem()
    P = orc.Problem(prob)
    nm = orc.NelderMeadBilevelOptimizationSolver(theta_high_init=40.0)     # breakdown near 12.5: 40 -> 20 -> 10
    rc, th, *_rest, val = nm.solve(P, x0, u, 0.1)
    assert rc == 0 and nm.c.theta_high_init == 10.0 and np.isfinite(val) and 0 < th < 12.6
    # value + kl/theta is flat around its minimum (CE golden: theta = 1.59, 14.0071); NM stops at stdev < 1e-2
    assert val < nm.compute_cost(P, x0, u, 10.0, 0.1) and val < 14.03


def test_c_high_c_low_are_stale_across_solve_calls():               # initialize! :164-168 does not reset them
    P, x0, u = nonlinear()
    nm = orc.NelderMeadBilevelOptimizationSolver()
    nm.solve(P, x0, u, 1.0)
    ch, cl, th_h, th_l = nm.c.c_high, nm.c.c_low, nm.c.theta_high, nm.c.theta_low
    nm.initialize()
    # vertices are back at their initial guesses, but the costs of the previous call's final simplex are still attached
    assert (nm.c.theta_high, nm.c.theta_low) == (nm.c.theta_high_init, nm.c.theta_low_init) != (th_h, th_l)
    assert nm.c.has_c_high and nm.c.has_c_low and (nm.c.c_high, nm.c.c_low) == (ch, cl)
