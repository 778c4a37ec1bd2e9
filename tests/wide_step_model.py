"""NumPy model of the backward step that csrc/wide.hip (sweep) implements for problems beyond the 12 + 4 MFMA tile.

Design note (not product code).  ileqg.jl:361-391 writes the step with D = I + theta S M^-1, M = inv(W) - theta S.  The general-size
kernel never forms D or M^-1: with the Cholesky factor M = U'U and ONE forward substitution [Z | z] = U^-T [S | s_vec] (a lane per
column on the device)

    D S       = S + theta Z'Z                 (symmetric)          D s_vec = s_vec + theta Z'z
    s_vec' M^-1 s_vec = z'z                                        logdet(W M) = logdet W + 2 sum log U_kk

and isposdef(M) is "every Cholesky pivot > 0".  H = Uh'Uh decides isposdef(H); [L | dl] = -H^-1 [G | g] by two substitutions.
Checked against the CPU oracle by tests/test_cpu_wide_step_model.py (no GPU needed)."""
import numpy as np
from scipy.linalg import solve_triangular


def step(S, sv, s1, tile, W, theta, mu, L_given=None, dl_given=None):
    q, qv, Q, r, R, P, A, B = tile
    n = S.shape[0]
    M = np.linalg.inv(W) - theta * S
    M = np.triu(M) + np.triu(M, 1).T                        # Symmetric(...): the upper triangle rules
    try:
        U = np.linalg.cholesky(M).T                         # M = U'U
    except np.linalg.LinAlgError:
        return None, "M"
    if theta == 0.0:
        Z, z = np.zeros((n, n)), np.zeros(n)                # D = I exactly
    else:
        Zz = solve_triangular(U, np.column_stack([S, sv]), trans="T", lower=False)
        Z, z = Zz[:, :n], Zz[:, n]
    DS = S + theta * Z.T @ Z
    dsv = sv + theta * Z.T @ z
    T, F = DS @ A, DS @ B
    g = r + B.T @ dsv
    G = P + B.T @ T
    H = R + B.T @ F + mu * np.eye(R.shape[0])
    H = np.triu(H) + np.triu(H, 1).T
    if L_given is None:
        try:
            Uh = np.linalg.cholesky(H).T
        except np.linalg.LinAlgError:
            return None, "H"
        X = -solve_triangular(Uh, solve_triangular(Uh, np.column_stack([G, g]), trans="T", lower=False), lower=False)
        L, dl = X[:, :n], X[:, n]
    else:
        L, dl = L_given, (np.zeros(R.shape[0]) if dl_given is None else dl_given)
    hv = H @ dl + g
    s0 = q + s1 + dl @ (0.5 * (H @ dl) + g)
    if theta == 0.0:
        s0 += 0.5 * np.trace(W @ S)
    else:
        s0 += 0.5 * theta * (z @ z) - (np.linalg.slogdet(W)[1] + 2.0 * np.sum(np.log(np.diag(U)))) / (2.0 * theta)
    sv0 = qv + A.T @ dsv + L.T @ hv + G.T @ dl
    S0 = Q + A.T @ T + L.T @ (H @ L + G) + G.T @ L
    S0 = np.triu(S0) + np.triu(S0, 1).T
    return (S0, sv0, s0, L, dl, g, G, H), None


def sweep(a, N, Wk, theta, mu, L=None, dl=None):
    """a: dict of ApproximationResult arrays (time first); Wk(t) -> W(t).  Returns L (N,m,n), dl (N,m), s (N+1), S_0, status."""
    S, sv, s1 = np.triu(a["Q"][N]) + np.triu(a["Q"][N], 1).T, a["qv"][N].copy(), float(a["q"][N])
    m, n = a["P"][0].shape
    Ls, dls, s = np.zeros((N, m, n)), np.zeros((N, m)), np.zeros(N + 1)
    s[N] = s1
    for t in reversed(range(N)):
        tile = (a["q"][t], a["qv"][t], a["Q"][t], a["r"][t], a["R"][t], a["P"][t], a["A"][t], a["B"][t])
        out, why = step(S, sv, s1, tile, Wk(t), theta, mu, None if L is None else L[t], None if dl is None else dl[t])
        if out is None:
            return Ls, dls, s, S, why
        S, sv, s1, Ls[t], dls[t] = out[0], out[1], out[2], out[3], out[4]
        s[t] = s1
    return Ls, dls, s, S, None
