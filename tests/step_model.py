"""NumPy model of the augmented 16x16 Riccati step that csrc/kernels.hip (sweep_body) implements.

Design note (not product code): shows that one backward step of ileqg.jl:361-391 for n<=12, m<=4 is
14 products of 16x16 tiles on the *augmented* value matrix V = [[S, sv],[sv', 2s]] plus one 12x12 SPD
inverse and one 4x4 solve.  Checked against the CPU oracle by tests/test_cpu_step_model.py.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NP, MP, PD, AUG = 12, 4, 16, 12


def pad_tiles(a, t, n, m):
    """Z (12x16) = [A|B], C (16x16) = [[Q,P'],[P,R]], qr (16) = [qv|r], q.  Padded controls get R = I."""
    Z = np.zeros((PD, PD)); C = np.zeros((PD, PD)); qr = np.zeros(PD)
    Z[:n, :n] = a["A"][t]; Z[:n, NP:NP + m] = a["B"][t]
    C[:n, :n] = a["Q"][t]; C[NP:NP + m, :n] = a["P"][t]; C[:n, NP:NP + m] = a["P"][t].T
    C[NP:NP + m, NP:NP + m] = a["R"][t]
    for g in range(m, MP):
        C[NP + g, NP + g] = 1.0
    qr[:n] = a["qv"][t]; qr[NP:NP + m] = a["r"][t]
    return Z, C, qr, a["q"][t]


def step(V, Z, C, qr, q, Winv, W, logdetW, theta, mu, n, m, L_given=None, dl_given=None):
    S = V[:NP, :NP]
    if theta != 0.0:
        M = Winv - theta * S
        pd_ok = np.all(np.linalg.eigvalsh(M) > 0)
        Minv = np.zeros((PD, PD)); Minv[:NP, :NP] = np.linalg.inv(M)
        Y = Minv @ V                                   # 3 MFMA (K = 12)
        Vt = V + V[:, :NP] @ (theta * Y[:NP, :])        # 3 MFMA
        risk = -1.0 / (2 * theta) * (logdetW + np.linalg.slogdet(M)[1])
    else:
        pd_ok = True
        Vt = V.copy()
        risk = 0.5 * np.sum(W * S)
    T = Vt[:, :NP] @ Z[:NP, :]                          # 3 MFMA ; row 12 = s~' Z
    F = Z[:NP, :].T @ T[:NP, :] + C                     # 3 MFMA with C as accumulator input
    f = T[AUG, :] + qr
    H = F[NP:, NP:] + mu * np.eye(MP)
    Gaug = np.zeros((MP, PD)); Gaug[:, :NP] = F[NP:, :NP]; Gaug[:, AUG] = f[NP:]
    if L_given is None:
        h_ok = np.all(np.linalg.eigvalsh(H) > 0)
        Laug = -np.linalg.solve(H, Gaug)
    else:
        h_ok = True
        Laug = np.zeros((MP, PD)); Laug[:m, :n] = L_given
        if dl_given is not None:
            Laug[:m, AUG] = dl_given
    Uaug = H @ Laug + Gaug
    Fx = np.zeros((PD, PD))
    Fx[:NP, :NP] = F[:NP, :NP]; Fx[:NP, AUG] = f[:NP]; Fx[AUG, :NP] = f[:NP]
    Fx[AUG, AUG] = 2 * q + Vt[AUG, AUG] + 2 * risk
    Vn = Fx + Laug.T @ Uaug + Gaug.T @ Laug             # 2 MFMA (K = 4 each)
    Vn[AUG + 1:, :] = 0; Vn[:, AUG + 1:] = 0
    return Vn, Laug, pd_ok, h_ok


def sweep(a, n, m, N, W, theta, mu, L=None):
    Winv = np.eye(NP); Winv[:n, :n] = np.linalg.inv(W)
    Wp = np.zeros((NP, NP)); Wp[:n, :n] = W
    logdetW = np.linalg.slogdet(W)[1]
    V = np.zeros((PD, PD))
    V[:n, :n] = a["Q"][N]; V[:n, AUG] = a["qv"][N]; V[AUG, :n] = a["qv"][N]; V[AUG, AUG] = 2 * a["q"][N]
    Ls = np.zeros((N, m, n)); dls = np.zeros((N, m)); s = np.zeros(N + 1); s[N] = a["q"][N]
    for t in reversed(range(N)):
        Z, C, qr, q = pad_tiles(a, t, n, m)
        V, Laug, ok1, ok2 = step(V, Z, C, qr, q, Winv, Wp, logdetW, theta, mu, n, m,
                                 None if L is None else L[t])
        assert ok1 and ok2
        Ls[t] = Laug[:m, :n]; dls[t] = Laug[:m, AUG]; s[t] = V[AUG, AUG] / 2
    return Ls, dls, s, V
