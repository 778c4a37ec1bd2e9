"""NumPy model of the TIME-PARALLEL (segment-parallel) Riccati sweep that csrc/psweep.hip implements -- design note, not product code.

The backward recursion of ileqg.jl:352-391 / :429-460 is a chain of N dependent steps.  Split the horizon into P segments
[t_s, t_{s+1}).  The LAST segment runs the recursion itself from the terminal condition.  Every other segment builds, concurrently and
with no knowledge of the value function at its end, the COMPOSITE of its steps -- the map (S, s_vec) at t_{s+1} -> (S, s_vec) at t_s --
in the conditional-value-function form of an associative LQ scan (Sarkka & Garcia-Fernandez, "Temporal parallelization of dynamic
programming and linear quadratic control"), carried in the augmented homogeneous coordinates of the kernels (V = [[S, s_vec], [s_vec',
2 s]], x_hat = [x; 1]):

    element of a segment:   Jv   (13 x 13)  the value function the segment's own recursion yields from a ZERO terminal value
                            K    (13 x 13)  transpose of the segment's closed-loop transition  A_c = [[Abar, bbar], [0, 1]]
                            Cbar (12 x 12)  = Ubar - theta Sigbar: accumulated control authority (gain sweeps) minus theta x the
                                            accumulated noise covariance
    prepending step i (tile Z = [A|B], gains La = [L | dl] -- given for a policy evaluation, the step's own optimum on Jv for a gain sweep):
        M = inv(W) - theta Jv_S,  N = M^-1                                 (the step's own inverse: nothing extra to invert)
        G     = N K[:12]
        T_K   = [inv(W) G ; K[12] + theta s_vec' G]                        (= (I + theta Jv E N E')' K)
        K    <- [A' T_K[:12] ; T_K[12]] + La' (B' T_K[:12])                (= A_i' T_K with the closed-loop A_i = [[A + B L, B dl], [0, 1]])
        Sigbar <- Sigbar + K[:12]' G
        Ubar   <- Ubar + (B' T_K[:12])' H^-1 (B' T_K[:12])                 (gain sweeps only; H = R + B' D S B + mu I of the step)
        Jv   <- step(Jv)                                                   (the ordinary backward step)
    applying the element to the TRUE value V_b at the segment's end ("hop", information form: two SPD inversions, no pivoting):
        Xt = (S_b^-1 + Cbar)^-1,  w = S_b^-1 s_b,  Aa = [Abar | bbar + w]
        [S | s_vec] at t_s = Jv[:12, :13] + (Aa' Xt Aa)[:12, :13]

With the true boundary values known, every segment re-runs the ORDINARY step from its own boundary (phase 3): the gains, the isposdef
tests and the scalars (logdet(W M), theta s'M^-1 s, the additive part of s) are then produced by the sequential sweep's own arithmetic;
only the boundary values differ from the sequential sweep's, by rounding.  Critical path with P waves: a + (P - 2) hops + b steps instead
of N (a: last segment, b: first segment).  Checked against tests/step_model.py (the sequential model) by tests/test_cpu_psweep_model.py."""
import numpy as np

from step_model import NP, MP, PD, AUG, pad_tiles, step


def terminal(a, n, N):
    V = np.zeros((PD, PD))
    V[:n, :n] = a["Q"][N]; V[:n, AUG] = a["qv"][N]; V[AUG, :n] = a["qv"][N]; V[AUG, AUG] = 2 * a["q"][N]
    return V


def P(a, b):
    """the one product the MFMA accumulator layout offers without data movement: a' b"""
    return a.T @ b


class Composite:
    def __init__(self):
        self.Jv = np.zeros((PD, PD))
        self.K = np.zeros((PD, PD)); self.K[:AUG + 1, :AUG + 1] = np.eye(AUG + 1)
        self.Sig = np.zeros((PD, PD))
        self.U = np.zeros((PD, PD))
        self.ok = True

    def prepend(self, tile, Winv, Wp, logdetW, theta, mu, n, m, L_given=None, dl_given=None, gain=False):
        Z, C, qr, q = tile
        S = self.Jv[:NP, :NP]
        sv = self.Jv[:NP, AUG]
        M = Winv - theta * S
        Nm = np.linalg.inv(M)
        Jn, Laug, ok1, ok2 = step(self.Jv, Z, C, qr, q, Winv, Wp, logdetW, theta, mu, n, m,
                                  None if gain else L_given, None if gain else dl_given)
        self.ok = self.ok and ok1 and ok2
        G = np.zeros((PD, PD)); G[:NP, :] = P(Nm, self.K[:NP, :])                   # 3 MFMA
        TK = np.zeros((PD, PD)); TK[:NP, :] = Winv @ G[:NP, :]                      # row scaling when W is diagonal
        TK[AUG, :] = self.K[AUG, :] + theta * (sv @ G[:NP, :])                      # per-lane dot + row reduction
        FK = P(Z[:NP, :], TK[:NP, :])                                              # 3 MFMA: rows 0..11 A'T, rows 12..15 B'T
        GK = np.zeros((MP, PD)); GK[:, :] = FK[NP:, :]
        Kn = np.zeros((PD, PD)); Kn[:NP, :] = FK[:NP, :]; Kn[AUG, :] = TK[AUG, :]
        Kn += P(Laug, GK)                                                          # 1 MFMA (K = 4)
        self.Sig = self.Sig + P(self.K[:NP, :], G[:NP, :])                          # 3 MFMA
        if gain:
            # H of the step on Jv (the step above formed and factorised it): F = Z'(DS)Z + C, H = F[12:,12:] + mu I
            Vt = self.Jv + self.Jv[:, :NP] @ (theta * (Nm @ self.Jv[:NP, :])) if theta != 0.0 else self.Jv
            T = Vt[:, :NP] @ Z[:NP, :]
            F = Z[:NP, :].T @ T[:NP, :] + C
            H = F[NP:, NP:] + mu * np.eye(MP)
            self.U = self.U + P(GK, np.linalg.solve(H, GK))                        # 4 x 4 solves per lane + 1 MFMA
        self.K = Kn
        self.Jv = Jn
        return Laug

    @property
    def Cbar(self):
        return self.U[:NP, :NP], self.Sig[:NP, :NP]


def hop(comp, Vb, theta, n=NP):
    """(S, s_vec) at the segment's start from the true value at its end; the [12][12] entry (the additive scalar) is not propagated.
    Problems with n < 12 are embedded with zero padding: the padded diagonal of S_b is set to 1 for the inversions (the padded rows of
    the transition are zero, so nothing of it reaches the result) and the padded rows / columns of Cbar are cleared."""
    U, Sig = comp.Cbar
    Cb = U - theta * Sig
    Cb[n:, :] = 0.0; Cb[:, n:] = 0.0            # (the padding's unit "noise" -- inv(W) is padded with 1 -- is not part of the problem)
    Sb, sb = Vb[:NP, :NP].copy(), Vb[:NP, AUG]
    Sb[n:, n:] += np.eye(NP - n)
    Sinv = np.linalg.inv(Sb)                                                       # SPD inversion 1 (six 2 x 2 block-pivot rounds)
    pd1 = np.all(np.linalg.eigvalsh(0.5 * (Sb + Sb.T)) > 0)
    Y = Sinv + Cb
    pd2 = np.all(np.linalg.eigvalsh(0.5 * (Y + Y.T)) > 0)
    Xt = np.linalg.inv(Y)                                                          # SPD inversion 2
    w = Sinv @ sb
    Ac = comp.K.T
    Aa = np.zeros((PD, PD)); Aa[:NP, :] = Ac[:NP, :]; Aa[:NP, AUG] += w
    Vt = comp.Jv + P(Aa[:NP, :], Xt @ Aa[:NP, :])                                  # 3 + 3 MFMA
    Vt[AUG + 1:, :] = 0; Vt[:, AUG + 1:] = 0
    Vt[AUG, AUG] = 0.0
    return Vt, pd1 and pd2


def hop_noise_form(comp, Vb, theta):
    """policy evaluation only (Ubar = 0): M_bar = Sigbar^-1 - theta S_b, one inversion on the chain (Sigbar^-1 is formed off it)."""
    _, Sig = comp.Cbar
    Sb = Vb[:NP, :NP]
    Sig = Sig + np.diag((np.diag(Sig) == 0.0).astype(float))                     # padded coordinates (n < 12): no noise reaches them
    Mb = np.linalg.inv(Sig) - theta * Sb
    Nb = np.zeros((PD, PD)); Nb[:NP, :NP] = np.linalg.inv(Mb)
    Ac = comp.K.T
    X = P(Vb, Ac)                                                                  # 4 MFMA (13 rows)
    T = X + Vb[:, :NP] @ (theta * (Nb[:NP, :NP] @ X[:NP, :]))                      # 3 + 3
    Vt = comp.Jv + P(Ac, T)                                                        # 4
    Vt[AUG + 1:, :] = 0; Vt[:, AUG + 1:] = 0
    Vt[AUG, AUG] = 0.0
    return Vt


def boundaries(N, P_, hop_cost=1.3, comp_cost=1.25):
    """cuts of P waves' P + 1 segments, 0 = cut[0] < ... < cut[P + 1] = N (driver.cpp: psweep_cuts).  Wave P-1 runs the recursion over the
    last segment (a steps), posts its value and carries on through segment P-1; wave w <= P-2 meanwhile builds the element of segment
    w+1, hops when the boundary value arrives and runs the ordinary recursion over segment w.  All waves end together when
    b_w = b_0 + w hop, and the first element must be ready when the last segment is done: a = comp b_{P-1}."""
    P = P_
    while P >= 2:
        b0 = (N - comp_cost * (P - 1) * hop_cost - hop_cost * P * (P - 1) / 2.0) / (P + comp_cost)
        if b0 >= 1.0 and N >= 2 * (P + 1):
            break
        P -= 1
    if P < 2:
        return [0, N, N]
    cuts, t = [0], 0.0
    for w in range(P):
        t += b0 + w * hop_cost
        cuts.append(int(t + 0.5))
    cuts.append(N)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1] + 1)
    cuts[-1] = N
    for i in range(len(cuts) - 2, 0, -1):
        cuts[i] = min(cuts[i], cuts[i + 1] - 1)
    return cuts


def psweep(a, n, m, N, W, theta, mu, cuts, L=None, dl=None, noise_form=False):
    """segment-parallel sweep over the segments [cuts[s], cuts[s+1]) (P = len(cuts) - 2 waves); L given: policy evaluation, else gain
    sweep.  Returns (Ls, dls, V_0 with V[12][12] = the summed additive scalar, the values handed along the chain, ok)."""
    gain = L is None
    Winv = np.eye(NP); Winv[:n, :n] = np.linalg.inv(W)
    Wp = np.zeros((NP, NP)); Wp[:n, :n] = W
    logdetW = np.linalg.slogdet(W)[1]
    S_ = len(cuts) - 1                         # segments 0 .. S_-1; the last TWO are the last wave's own recursion
    tiles = [pad_tiles(a, t, n, m) for t in range(N)]
    # phase 1: the elements of segments 1 .. S_-2 (built by waves 0 .. P-2 while the last wave runs segment S_-1)
    comps = {}
    for s in range(1, S_ - 1):
        c = Composite()
        for t in reversed(range(cuts[s], cuts[s + 1])):
            c.prepend(tiles[t], Winv, Wp, logdetW, theta, mu, n, m, None if gain else L[t],
                      None if (gain or dl is None) else dl[t], gain=gain)
        comps[s] = c
    ok = all(c.ok for c in comps.values())
    Ls = np.zeros((N, m, n)); dls = np.zeros((N, m))

    def recurse(V, s):
        nonlocal ok
        for t in reversed(range(cuts[s], cuts[s + 1])):
            Z, C, qr, q = tiles[t]
            V, Laug, ok1, ok2 = step(V, Z, C, qr, q, Winv, Wp, logdetW, theta, mu, n, m, None if gain else L[t],
                                     None if (gain or dl is None) else dl[t])
            ok = ok and ok1 and ok2
            Ls[t] = Laug[:m, :n]; dls[t] = Laug[:m, AUG]
        return V

    # the last wave: segment S_-1 from the terminal condition, posts, carries on through segment S_-2
    Vb = {}
    V = recurse(terminal(a, n, N), S_ - 1)
    Vb[S_ - 1] = V.copy()                       # the value at cuts[S_-1]: handed to the wave that owns the element of segment S_-2
    scal = 0.0
    if S_ >= 2:
        V = recurse(V, S_ - 2)
    scal += V[AUG, AUG]
    V_first = V
    # the chain: wave w hops over segment w+1 and runs the ordinary recursion over segment w
    for s in reversed(range(1, S_ - 1)):
        Vh, okh = (hop_noise_form(comps[s], Vb[s + 1], theta), True) if (noise_form and not gain) else hop(comps[s], Vb[s + 1], theta, n)
        ok = ok and okh
        Vb[s] = Vh
        V = recurse(Vh.copy(), s - 1)
        scal += V[AUG, AUG]
        if s == 1:
            V_first = V
    V0 = V_first.copy(); V0[AUG, AUG] = scal
    return Ls, dls, V0, Vb, ok
