"""First-principles risk-sensitive value of an AFFINE policy on a linear-quadratic-Gaussian problem -- test aid, NumPy only.

This does not restate anything from the reference or from the oracle: no Riccati recursion, no M = inv(W) - theta S, no D.  It writes
the closed-loop trajectory as an affine map of the stacked noise w = (w_0 .. w_{N-1}) ~ N(0, Sigma), Sigma = blkdiag(W(k)), so that the
total cost is ONE quadratic form  J(w) = c0 + b'w + 1/2 w'H w,  and evaluates the Gaussian integral in closed form:

    (1/theta) log E exp(theta J) = c0 + theta/2 b'(Sigma^-1 - theta H)^-1 b - 1/(2 theta) logdet(I - theta Sigma H)     (theta > 0)
    E J                          = c0 + 1/2 tr(Sigma H)                                                                 (theta = 0)

For LQ problems the quadratic model of approximate_model (ileqg.jl:258-322) is exact, so this is what solve_approximate_dp(!)
(ileqg.jl:341-465) must return as s_array[1] for the policy u_k = l_k + dl_k + L_k (x_k - xbar_k), and what solve! returns as `value`
for the policy (l_array, L_array) around x_array.  It pins every theta > 0 term of the recursion (D, theta/2 s'M^-1 s, logdet(W M)) and,
through stationarity, the gain formula L = -H^-1 G with D S -- which the reference's own tests pin only at theta = 0 (K5)."""
import numpy as np


def _tv(tab, k, tv):
    return tab[k] if tv else tab


def exact_value(prob, x0, l, dl, L, xbar, theta):
    """Value of u_k = l_k + dl_k + L_k (x_k - xbar_k) from x_0 on `prob` (an LQRiskSensitiveProblem with kappa = 0).
    Returns (value, feasible): feasible is False when Sigma^-1 - theta H is not positive definite (the integral diverges)."""
    n, m, N = prob.n, prob.m, prob.N
    assert prob.kappa == 0.0
    A, B = prob.A, prob.B
    l, L, xbar = np.asarray(l, float), np.asarray(L, float), np.asarray(xbar, float)
    dl = np.zeros((N, m)) if dl is None else np.asarray(dl, float)
    off = l + dl - np.einsum("kij,kj->ki", L, xbar[:N])          # u_k = off_k + L_k x_k
    # x = xd + Mx w,  u = ud + Mu w   (Mx: n(N+1) x nN, block lower triangular)
    xd = np.zeros((N + 1, n))
    xd[0] = x0
    Mx = np.zeros((N + 1, n, N, n))
    for k in range(N):
        Acl = A + B @ L[k]
        xd[k + 1] = Acl @ xd[k] + B @ off[k]
        Mx[k + 1] = np.einsum("ij,jab->iab", Acl, Mx[k])
        Mx[k + 1, :, k, :] += np.eye(n)
    ud = off + np.einsum("kij,kj->ki", L, xd[:N])
    Mx2 = Mx.reshape(N + 1, n, N * n)
    Mu2 = np.einsum("kij,kjc->kic", L, Mx2[:N])
    c0, b, H = 0.0, np.zeros(N * n), np.zeros((N * n, N * n))
    for k in range(N):
        Q, R, P = _tv(prob.Q, k, prob.cost_tv), _tv(prob.R, k, prob.cost_tv), _tv(prob.P, k, prob.cost_tv)
        qv, rv, q0 = _tv(prob.qv, k, prob.cost_tv), _tv(prob.rv, k, prob.cost_tv), _tv(prob.q0, k, prob.cost_tv)
        x, u, X, U = xd[k], ud[k], Mx2[k], Mu2[k]
        c0 += 0.5 * x @ Q @ x + 0.5 * u @ R @ u + u @ P @ x + qv @ x + rv @ u + float(q0)
        b += X.T @ (Q @ x + qv + P.T @ u) + U.T @ (R @ u + rv + P @ x)
        H += X.T @ Q @ X + U.T @ R @ U + U.T @ P @ X + X.T @ P.T @ U
    x, X = xd[N], Mx2[N]
    c0 += 0.5 * x @ prob.Qf @ x + prob.qvf @ x + prob.q0f
    b += X.T @ (prob.Qf @ x + prob.qvf)
    H += X.T @ prob.Qf @ X
    H = 0.5 * (H + H.T)
    Sig = np.zeros((N * n, N * n))
    for k in range(N):
        Sig[k * n:(k + 1) * n, k * n:(k + 1) * n] = prob.W(k)
    if theta == 0.0:
        return c0 + 0.5 * np.trace(Sig @ H), True
    # Sigma^-1 - theta H = C^-T (I - theta C' H C) C^-1 with Sigma = C C'
    C = np.linalg.cholesky(Sig)
    K = np.eye(N * n) - theta * (C.T @ H @ C)
    K = 0.5 * (K + K.T)
    ev = np.linalg.eigvalsh(K)
    if ev.min() <= 0.0:
        return np.inf, False
    cb = C.T @ b
    quad = cb @ np.linalg.solve(K, cb)                          # b'(Sigma^-1 - theta H)^-1 b
    logdet = float(np.sum(np.log(ev)))                           # logdet(I - theta Sigma H)
    return c0 + 0.5 * theta * quad - logdet / (2.0 * theta), True


def random_lq(n, m, N, seed, w_scale=1e-2, tv=True):
    """A well-posed LQ problem with everything the recursion can see switched on: time-varying symmetric Q_k, R_k > 0, cross term
    P_k != 0, linear terms, time-varying non-diagonal W(k), terminal linear term."""
    import ratilqr.jl_amd as rat
    rng = np.random.default_rng(seed)
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = 0.9 * Qo
    B = rng.standard_normal((n, m)) / np.sqrt(n)

    def spd(k, lo, size):
        M = rng.standard_normal((size, k, k))
        return np.einsum("tij,tkj->tik", M, M) / k + lo * np.eye(k)

    T = N if tv else 1
    Q, R = spd(n, 0.5, T), spd(m, 0.3, T)
    P = 0.1 * rng.standard_normal((T, m, n))
    qv, rv, q0 = 0.2 * rng.standard_normal((T, n)), 0.2 * rng.standard_normal((T, m)), rng.standard_normal(T)
    W = w_scale * spd(n, 0.5, N)
    sq = (lambda a: a) if tv else (lambda a: a[0])
    prob = rat.LQRiskSensitiveProblem(A, B, Q=sq(Q), R=sq(R), N=N, W=W, P=sq(P), qv=sq(qv), rv=sq(rv), q0=sq(q0),
                                      Qf=spd(n, 0.5, 1)[0], qvf=0.2 * rng.standard_normal(n), q0f=0.7)
    return prob, rng.standard_normal(n), 0.1 * rng.standard_normal((N, m))


def breakdown_theta(prob, x0, l, L, xbar, hi=1e4):
    """Largest feasible theta of the given policy (bisection on the feasibility of the Gaussian integral), to place test thetas."""
    lo_t, hi_t = 0.0, hi
    if exact_value(prob, x0, l, None, L, xbar, hi)[1]:
        return hi
    for _ in range(40):
        mid = 0.5 * (lo_t + hi_t)
        if exact_value(prob, x0, l, None, L, xbar, mid)[1]:
            lo_t = mid
        else:
            hi_t = mid
    return lo_t
