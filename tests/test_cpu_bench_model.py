"""The byte / flop model behind bench.py's `roofline` object equals SURVEY.md section 8(d)'s per-unit figures (no GPU needed)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_per_unit():
    a = bench.algo_bytes()
    tile = 50 * 417 + 157
    assert tile * 8 == 168056                                   # tile bundle T (21,007 doubles)
    assert a["rollout_candidate"] == 4224 * 8                   # 33.8 KB
    assert a["linearise"] == (812 + tile) * 8                   # 174.6 KB
    assert a["sweep_eval"] == (tile + 2400 + 1) * 8 == 187264   # 187.3 KB
    assert a["sweep_gain"] == (tile + 2400 + 200 + 1) * 8 == 188864
    assert a["candidate"] == 395608 and a["init"] == 368312     # 395.6 KB / 368.3 KB
    # the 2-iteration, 2-evaluation LQ solve: 1.54 MB
    assert bench.algo_bytes_of_solves(np.array([2]), np.array([2])) == 368312 + 2 * 188864 + 2 * 395608 == 1537256


def test_algorithmic_flops_per_solve():
    # 5 sweeps x 50 steps x 22.2 kflop + 3 trajectories x (24 + 20) kflop = 5.68 Mflop
    assert abs(bench.algo_flops_of_solves(np.array([2]), np.array([2])) - 5.682e6) < 1.0


def test_theta_draw_is_positive_and_seeded():
    t1, t2 = bench.draw_theta(1024, 7), bench.draw_theta(1024, 7)
    assert np.array_equal(t1, t2) and t1.size == 1024 and np.all(t1 > 0)
