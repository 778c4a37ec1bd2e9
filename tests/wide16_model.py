"""NumPy model of the backward step of csrc/wide16.h (general sizes with n <= 16, m <= 4 in registers on the matrix pipe).

Design note (not product code).  The kernel carries every quantity as 16 x 16 tiles -- "1": the n state columns, "2": columns 0..3 the m
controls, column 4 the affine part -- and its one product is P(a, b) = a'b.  The step (ileqg.jl:361-391), padded to 16 states / 4 controls:

    M  = inv(W) - theta S          (unit diagonal on the padding)         Mi = -M^-1 by the symmetric sweep (here: np.linalg.inv)
    X1 = S A       X2 = S [B | 0] + [0 | s_vec]                          Y = theta M^-1 X          T = X + S Y = (D S)[A | B | S^-1 s_vec]
    F11 = Q + A'T1      F12 = A'T2 + [0 | q_vec]: columns 0..3 = (G - P)', column 4 = q_vec + A'D s_vec
    F22 = [B | 0]'T2 + [R + mu I | r_vec]: [H | g]          G = P + (F12[:, :4])'
    H [L | dl] = -[G | g];       S' = F11 + L'(H L + G) + G'L;       s_vec' = F12[:, 4] + L'(H dl + g) + G'dl
    scalars: q + 0.5 theta s_vec'M^-1 s_vec (= 0.5 s_vec . Y2[:, 4]) - (logdet W + logdet M) / (2 theta) + dl'(0.5 H dl + g)
             (theta = 0: 0.5 tr(W S); D = I)

isposdef(M) <=> all leading minors > 0 (the 2 x 2 block pivots of the sweep); isposdef(H) <=> LDL' pivots > 0.
Checked against the CPU oracle by tests/test_cpu_wide16_model.py (no GPU needed)."""
import numpy as np

NP, MP, AFF = 16, 4, 4          # padded states, padded controls, affine column of the "2" tile


def _pad(X, r, c, diag=0.0):
    out = np.zeros((r, c))
    out[:X.shape[0], :X.shape[1]] = X
    if diag:
        for i in range(min(X.shape[0], X.shape[1]), min(r, c)):
            out[i, i] = diag
    return out


def leading_minors_positive(M):
    return all(np.linalg.det(M[:k, :k]) > 0.0 for k in range(1, M.shape[0] + 1))


def step(S, sv, tile, W, theta, mu, L_given=None):
    """S (n, n), sv (n): value function behind the step; returns (S', sv', scalar increment, L, dl) or (None, why)."""
    q, qv, Q, r, R, P, A, B = tile
    n, m = S.shape[0], R.shape[0]
    Sp, svp = _pad(S, NP, NP), np.concatenate([sv, np.zeros(NP - n)])
    Ap, Qp = _pad(A, NP, NP), _pad(Q, NP, NP)
    Z2 = np.zeros((NP, 16)); Z2[:n, :m] = B
    Pn = np.zeros((MP, NP)); Pn[:m, :n] = P
    Rn = _pad(R, MP, MP, 1.0) + mu * np.eye(MP)
    qvp, rvp = np.concatenate([qv, np.zeros(NP - n)]), np.concatenate([r, np.zeros(MP - m)])
    Winv = _pad(np.linalg.inv(W), NP, NP, 1.0)
    X1 = Sp @ Ap
    X2 = Sp @ Z2
    X2[:, AFF] += svp
    inc = 0.0
    if theta != 0.0:
        M = Winv - theta * Sp
        M = np.triu(M) + np.triu(M, 1).T
        if not leading_minors_positive(M):
            return None, "M"
        Minv = np.linalg.inv(M)
        Y1, Y2 = theta * Minv @ X1, theta * Minv @ X2
        T1, T2 = X1 + Sp @ Y1, X2 + Sp @ Y2
        inc += 0.5 * svp @ Y2[:, AFF] - (np.linalg.slogdet(W)[1] + np.linalg.slogdet(M)[1]) / (2.0 * theta)
    else:
        if not np.all(np.isfinite(S)):
            return None, "M"
        T1, T2 = X1, X2
        inc += 0.5 * np.trace(W @ S)
    F11 = Qp + Ap.T @ T1
    F12 = Ap.T @ T2
    F12[:, AFF] += qvp
    F22 = Z2.T @ T2                                           # rows 0..3: B'(D S)B | B'D s_vec
    H = Rn + F22[:MP, :MP]
    H = np.triu(H) + np.triu(H, 1).T
    g = rvp + F22[:MP, AFF]
    G = Pn + F12[:, :MP].T                                    # (A'(D S)B)' = B'(D S)A
    if L_given is None:
        if not leading_minors_positive(H):
            return None, "H"
        X = -np.linalg.solve(H, np.column_stack([G, g]))
        L, dl = X[:, :NP], X[:, NP]
    else:
        L, dl = _pad(L_given, MP, NP), np.zeros(MP)
    Sn = F11 + L.T @ (H @ L + G) + G.T @ L
    svn = F12[:, AFF] + L.T @ (H @ dl + g) + G.T @ dl
    inc += q + dl @ (0.5 * (H @ dl) + g)
    return (Sn[:n, :n], svn[:n], inc, L[:m, :n], dl[:m]), None


def sweep(a, N, Wk, theta, mu, L=None):
    """a: dict of ApproximationResult arrays (time first); Wk(t) -> W(t).  Returns L (N, m, n), dl (N, m), s_0, S_0, status."""
    Qn = a["Q"][N]
    S, sv, s = np.triu(Qn) + np.triu(Qn, 1).T, a["qv"][N].copy(), float(a["q"][N])
    m, n = a["R"][0].shape[0], S.shape[0]
    Ls, dls = np.zeros((N, m, n)), np.zeros((N, m))
    for t in range(N - 1, -1, -1):
        tile = (float(a["q"][t]), a["qv"][t], np.triu(a["Q"][t]) + np.triu(a["Q"][t], 1).T, a["r"][t],
                np.triu(a["R"][t]) + np.triu(a["R"][t], 1).T, a["P"][t], a["A"][t], a["B"][t])
        out, why = step(S, sv, tile, Wk(t), theta, mu, None if L is None else L[t])
        if out is None:
            return None, None, None, None, why
        S, sv, inc, Ls[t], dls[t] = out
        s += inc
    return Ls, dls, s, S, None
