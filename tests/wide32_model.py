"""NumPy model of the backward step of csrc/wide32.h (general sizes in registers on BLOCKS of 16 x 16 tiles: n <= 32, m <= 32).

Design note (not product code).  The kernel pads the n states to 16 NT and the m controls to 16 MT (NT, MT in {1, 2}), keeps the affine parts
as separate one-column blocks, and has ONE way of inverting: the symmetric sweep with 2 x 2 block pivots over the padded matrix -- for
M = inv(W) - theta S (unit diagonal on the padding) and, unlike wide16.h, for H = R + B'(D S)B + mu I as well (m up to 32: no per-lane LDL').
`sweep_inverse` below IS that elimination, round by round as the kernel runs it (M' = M o mask - t'(Bk t) with the pivot rows t carrying -I in
the pivot block; rounds whose pivot rows lie beyond the matrix skipped; isposdef <=> every block's leading minors p11 > 0, det P > 0; the
determinant as the running product of the block determinants), so the model checks the ALGORITHM, not just the algebra:

    M  -> Mi = -M^-1,  logdet M = sum log det P_k
    X = S [A | B | .] + [0 | 0 | s_vec]      Y = theta M^-1 X      T = X + S Y
    F11 = Q + A'T_A   F1a = q_vec + A'T_a   G = P + B'T_A   H = R + mu I + B'T_B   g = r + B'T_a
    H -> Hi = -H^-1 (the same elimination),  L = Hi G,  dl = Hi g
    S' = F11 + L'(H L + G) + G'L,   s_vec' = F1a + L'(H dl + g) + G'dl
    scalars: q + 0.5 theta s_vec'M^-1 s_vec - (logdet W + logdet M) / (2 theta) + dl'(0.5 H dl + g)   (theta = 0: 0.5 tr(W S))

Checked against the CPU oracle by tests/test_cpu_wide32_model.py (no GPU needed)."""
import numpy as np


def _pad(X, r, c, diag=0.0):
    out = np.zeros((r, c))
    out[:X.shape[0], :X.shape[1]] = X
    if diag:
        for i in range(min(X.shape[0], X.shape[1]), min(r, c)):
            out[i, i] = diag
    return out


def sweep_inverse(M, size):
    """-M^-1 of the padded symmetric matrix M by 2 x 2 block-pivot rounds over rows 2k, 2k + 1 < size (the kernel's elim32_rounds).
    Returns (Mi, positive definite?, log det of the swept part)."""
    M = M.copy()
    p = M.shape[0]
    pd, logdet = True, 0.0
    for k in range(0, p, 2):
        if k >= size:                                   # pivot rows beyond the matrix: unit diagonal, nothing coupled -- skipped
            continue
        K = [k, k + 1]
        P = M[np.ix_(K, K)]
        p11, p12, p22 = P[0, 0], P[0, 1], P[1, 1]
        det = p11 * p22 - p12 * p12
        pd = pd and (p11 > 0.0) and (det > 0.0)
        logdet += np.log(det) if det > 0 else np.nan
        Bk = np.array([[p22, -p12], [-p12, p11]]) / det
        t = M[K, :].copy()                              # the pivot rows, -I in the pivot block
        t[:, K] = -np.eye(2)
        mask = np.ones_like(M)
        mask[K, :] = 0.0
        mask[:, K] = 0.0
        M = M * mask - t.T @ (Bk @ t)
    return M, pd, logdet


def step(S, sv, tile, W, theta, mu, L_given=None):
    """S (n, n), sv (n): value function behind the step; returns ((S', sv', scalar increment, L, dl), None) or (None, why)."""
    q, qv, Q, r, R, P, A, B = tile
    n, m = S.shape[0], R.shape[0]
    NP, MP = 16 * (1 if (n <= 16 and m <= 16) else 2), 16 * (1 if m <= 16 else 2)
    Sp, svp = _pad(S, NP, NP), np.concatenate([sv, np.zeros(NP - n)])
    Ap, Qp, Bp = _pad(A, NP, NP), _pad(Q, NP, NP), _pad(B, NP, MP)
    Pm, Rp = _pad(P, MP, NP), _pad(R, MP, MP, 1.0)
    qvp, rvp = np.concatenate([qv, np.zeros(NP - n)]), np.concatenate([r, np.zeros(MP - m)])
    X1, XB, Xa = Sp @ Ap, Sp @ Bp, svp.copy()
    inc = 0.0
    if theta != 0.0:
        M = _pad(np.linalg.inv(W), NP, NP, 1.0) - theta * Sp
        Mi, pd, ldm = sweep_inverse(M, n)
        if not pd or not np.isfinite(ldm):
            return None, "M"
        Y1, YB, Ya = -theta * (Mi @ X1), -theta * (Mi @ XB), -theta * (Mi @ Xa)
        T1, TB, Ta = X1 + Sp @ Y1, XB + Sp @ YB, Xa + Sp @ Ya
        inc += 0.5 * svp @ Ya - (np.linalg.slogdet(W)[1] + ldm) / (2.0 * theta)
    else:
        if not np.all(np.isfinite(S)):
            return None, "M"
        T1, TB, Ta = X1, XB, Xa
        inc += 0.5 * np.trace(W @ S)
    F11, F1a = Qp + Ap.T @ T1, qvp + Ap.T @ Ta
    G, H, g = Pm + Bp.T @ T1, Rp + mu * np.eye(MP) + Bp.T @ TB, rvp + Bp.T @ Ta
    if L_given is None:
        Hi, pd, _ = sweep_inverse(H, m)
        if not pd:
            return None, "H"
        L, dl = Hi @ G, Hi @ g
        L[m:, :] = 0.0; dl[m:] = 0.0                     # (rows of the skipped rounds: the kernel's products meet zero rows of G, g there)
    else:
        L, dl = _pad(L_given, MP, NP), np.zeros(MP)
    Sn = F11 + L.T @ (H @ L + G) + G.T @ L
    svn = F1a + L.T @ (H @ dl + g) + G.T @ dl
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
