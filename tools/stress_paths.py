"""Every single-launch solve kernel against the round-based path, bit for bit, over random problems the fixed tests do not combine:
time-varying cost and noise tables, cubic drift, the power-law family, horizons around the staging limit, batch sizes from 1 to 3000
(block kernel up to 512, paired fused kernel up to 1024, beyond it the two-samples-per-SIMD tile-free fused kernel for the LQ family).
  STRESS_S=120 python tools/stress_paths.py      (on an MI355X)"""
import os
os.environ.setdefault("RATILQR_SPEC_FORCE", "1")     # handles of width E > 1 run the speculative kernels here (spec_eps is otherwise an upper bound)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat


def lq(rng):
    n, m, N = int(rng.integers(1, 13)), int(rng.integers(1, 5)), int(rng.choice([5, 20, 50, 50, 52, 53, 60]))
    tv = bool(rng.integers(0, 2))
    A = (0.7 + 0.25 * rng.random()) * np.linalg.qr(rng.standard_normal((n, n)))[0]
    B = rng.standard_normal((n, m)) / np.sqrt(n)

    def spd(k, scale):
        G = rng.standard_normal((k, k))
        return scale * (np.eye(k) + 0.2 * G @ G.T / k)
    if tv:
        Q = np.stack([spd(n, 0.5 + rng.random()) for _ in range(N)])
        R = np.stack([spd(m, 0.1 + 0.3 * rng.random()) for _ in range(N)])
        P = 0.03 * rng.standard_normal((N, m, n))
        qv, rv, q0 = 0.1 * rng.standard_normal((N, n)), 0.1 * rng.standard_normal((N, m)), rng.standard_normal(N)
    else:
        Q, R, P = spd(n, 1.0), spd(m, 0.2), 0.03 * rng.standard_normal((m, n))
        qv, rv, q0 = 0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(m), float(rng.standard_normal())
    W = np.stack([spd(n, 1e-3 * (0.5 + rng.random())) for _ in range(N)]) if rng.integers(0, 2) else spd(n, 1e-3)
    if rng.integers(0, 3) == 0:
        W = np.diag(1e-3 * 10.0 ** rng.uniform(-0.5, 0.5, n))           # diagonal W: the inv(W)-folded sweep arithmetic
    prob = rat.LQRiskSensitiveProblem(A, B, Q=Q, R=R, P=P, qv=qv, rv=rv, q0=q0, N=N, W=W, Qf=spd(n, 1.0), qvf=0.2 * rng.standard_normal(n),
                                      q0f=0.3, kappa=float(rng.choice([0.0, 0.02, -0.02, 0.04])))
    return prob, rng.uniform(0.3, 1.0) * rng.standard_normal(n), 0.1 * rng.standard_normal((N, m)), 10.0 ** rng.uniform(-2, 2.2)


def powerlaw(rng):
    n, N = int(rng.integers(1, 5)), int(rng.integers(3, 25))
    prob = rat.PowerLawRiskSensitiveProblem(n, N, 0.01 * np.eye(n), a=float(rng.choice([1.0, 1.3])), b=float(rng.choice([1.0, 1.5])),
                                            p=float(rng.choice([2.0, 2.5])), hconst=1.0)
    return prob, np.abs(0.3 * rng.standard_normal(n)), 0.1 + 0.05 * rng.random((N, n)), 2.0


def ctx_for(prob, B, env, E=1):
    for k, v in env.items():
        os.environ[k] = v
    try:
        return rat.Context(prob, max_batch=B, spec_eps=E)
    finally:
        for k in env:
            del os.environ[k]


def main():
    budget = float(os.environ.get("STRESS_S", "120"))
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "2")))
    t0, launches, bad, kinds = time.time(), 0, 0, {}
    while time.time() - t0 < budget:
        prob, x0, u, scale = powerlaw(rng) if rng.random() < 0.2 else lq(rng)
        B = int(rng.choice([1, 5, 100, 300, 512, 513, 800, 1024, 1025, 1500, 2100, 3000]))
        theta = np.concatenate([[0.0], np.abs(rng.normal(0.0, scale, B - 1))]) if B > 1 else np.array([scale])
        # speculation widths too: the default path (block kernel within one generation of workgroups, beyond it rounds whose candidates
        # carry no tile records and roll out in one wavefront per sample) against rounds with materialised tiles and one wave per candidate
        E = int(rng.choice([1, 1, 1, 2, 4, 8, 3])) if B <= 1024 else 1
        dflt, ref = ctx_for(prob, B, {"RATILQR_BLOCK_PSW": "0"}, E), ctx_for(prob, B, {"RATILQR_FUSED": "0", "RATILQR_FLY": "0"}, E)   # (bit-identity: the sequential-sweep paths)
        dflt.profile(True)
        a, b = dflt.solve_batch(x0, u, theta), ref.solve_batch(x0, u, theta)
        for k, p in dflt.profile_get().items():
            if p["launches"]:
                kinds[k] = kinds.get(k, 0) + 1
        launches += 1
        if not all(np.array_equal(p, q, equal_nan=True) for p, q in zip(a, b)):
            bad += 1
            print("MISMATCH", type(prob).__name__, dict(n=prob.n, m=prob.m, N=prob.N, B=B, E=E), flush=True)
    print(f"stress done: {launches} batches on the default path compared with the round-based path, {bad} mismatches, kernels {kinds}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
