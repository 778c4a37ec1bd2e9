"""NumPy model of the TIME-PARALLEL closed-loop rollout that csrc/kernels.hip implements as rollprl_body -- design note, not product code.

simulate_dynamics (ileqg.jl:62-87) under a feedback policy is x_{t+1} = f(x_t, u_t), u_t = l_t + eps dl_t + L_t (x_t - xbar_t): a chain of
N dependent steps.  For the LQ family with kappa = 0 (f = A x + B u) and (xbar, l) a trajectory of the same dynamics, the DEVIATION
dx_t = x_t - xbar_t obeys the affine recursion dx_{t+1} = (A + B L_t) dx_t + eps B dl_t, and affine maps compose.  The horizon is cut
into four segments [cut_w, cut_{w+1}), one per wavefront:
    wave 0     runs the ordinary step over segment 0 from x_0 at once;
    wave w>=1  builds the MAP of segment w - 1, dx_end = Phi dx_start + c, by running the deviation recursion on thirteen columns at once
               (columns 0..11: the unit vectors -> Phi; column 12: zero state + the affine input -> c), takes the deviation at that
               segment's start from wave w - 1 (zero for w = 1), "hops" (Phi dx + c), hands the result on, and runs the ordinary step
               over its own segment from xbar + dx.
Every x_t, u_t comes from the ordinary step's arithmetic on a start state that differs from the sequential one by the hop's rounding.
`cuts` restates the host's cost model (driver.cpp: rollprl_cuts): in units of one ordinary step an element step costs e, a hop h, the
terminal tile epi (last wave); T_w = max(e n_{w-1}, T_{w-1}) + h is when wave w starts its segment, all waves end together at F.
Checked by tests/test_cpu_rollprl_model.py; the device's cuts against `cuts` by tests/test_gpu_psweep.py."""
import numpy as np

WAVES = 4


def cuts(N, e=0.45, h=0.90, epi=1.00):
    def lens(F):
        n, T = [F], 0.0
        for w in range(1, WAVES):
            T = max(e * n[w - 1], T) + h
            n.append(max(F - T - (epi if w == WAVES - 1 else 0.0), 1.0))
        return n
    lo, hi = 1.0, float(N)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if sum(lens(mid)) < N:
            lo = mid
        else:
            hi = mid
    n, t, c = lens(hi), 0.0, [0]
    for w in range(WAVES):
        t += n[w]
        c.append(int(t + 0.5))
    c[WAVES] = N
    for w in range(1, WAVES + 1):
        c[w] = max(c[w], c[w - 1] + 1)
    c[WAVES] = N
    for w in range(WAVES - 1, 0, -1):
        c[w] = min(c[w], c[w + 1] - 1)
    return c


def finish_times(c, e=0.45, h=0.90, epi=1.00):
    """When each wave is done under the cost model (ordinary steps)."""
    n = [c[w + 1] - c[w] for w in range(WAVES)]
    out, T = [float(n[0])], 0.0
    for w in range(1, WAVES):
        T = max(e * n[w - 1], T) + h
        out.append(T + n[w] + (epi if w == WAVES - 1 else 0.0))
    return out


def sequential(A, B, L, l, dl, xbar, x0, eps):
    """The ordinary closed-loop rollout (ileqg.jl:74-85, kappa = 0)."""
    N = L.shape[0]
    x, u = np.zeros((N + 1, A.shape[0])), np.zeros((N, B.shape[1]))
    x[0] = x0
    for t in range(N):
        u[t] = (l[t] + eps * dl[t]) + L[t] @ (x[t] - xbar[t])
        x[t + 1] = A @ x[t] + B @ u[t]
    return x, u


def segment_map(A, B, L, dl, t0, t1, eps):
    """[Phi | c] of segment [t0, t1): thirteen columns through the deviation recursion at once."""
    n = A.shape[0]
    D = np.hstack([np.eye(n), np.zeros((n, 1))])
    for t in range(t0, t1):
        aff = np.zeros((B.shape[1], n + 1)); aff[:, n] = eps * dl[t]
        D = (A + B @ L[t]) @ D + B @ aff
    return D[:, :n], D[:, n]


def time_parallel(A, B, L, l, dl, xbar, x0, eps, c):
    N = L.shape[0]
    x, u = np.zeros((N + 1, A.shape[0])), np.zeros((N, B.shape[1]))
    dx = np.zeros(A.shape[0])
    for w in range(WAVES):
        if w >= 1:
            Phi, cc = segment_map(A, B, L, dl, c[w - 1], c[w], eps)
            dx = Phi @ dx + cc                                   # the hop (dx at cut_{w-1} is wave w - 1's start deviation)
        xs = xbar[c[w]] + dx if w else np.array(x0, dtype=float)
        x[c[w]] = xs
        for t in range(c[w], c[w + 1]):
            u[t] = (l[t] + eps * dl[t]) + L[t] @ (x[t] - xbar[t])
            x[t + 1] = A @ x[t] + B @ u[t]
    return x, u
