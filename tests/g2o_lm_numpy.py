"""TEST AID: g2o's Levenberg-Marquardt over BlockSolver<6,3> + LinearSolverDense (Eigen::LDLT), written a SECOND time -- numpy, dense
matrices, rotation matrices instead of quaternions, no shared code with oracle/ -- from the same published sources (SURVEY.md Appendix A.3;
optimization_algorithm_levenberg.cpp, block_solver.hpp, linear_solver_dense.h, Eigen LDLT.h).  tests/test_oracle_ba.py runs it beside the
C++ oracle: two independent transcriptions must take the same decisions (which solves fail, which steps are accepted, the same dampings)
until rounding -- the windows are gauge-free, hence chaotic -- separates them.  Parity with g2o itself stays unpinned."""
import numpy as np

def eigen_ldlt(A, b):
    """Eigen::LDLT<MatrixXd>(A).isPositive() ? solve(b) : None -- written from LDLT.h independently of the oracle (numpy, full matrices)."""
    A = np.array(A, float); n = len(A)
    tr = list(range(n)); sign = 0
    for k in range(n):
        big = k + int(np.argmax(np.abs(np.diag(A)[k:])))
        tr[k] = big
        if big != k:
            A[[k, big], :] = A[[big, k], :]
            A[:, [k, big]] = A[:, [big, k]]
        if k > 0:
            temp = np.diag(A)[:k] * A[k, :k]
            A[k, k] -= A[k, :k] @ temp
            A[k + 1:, k] -= A[k + 1:, :k] @ temp
        akk = A[k, k]
        if akk != 0:
            A[k + 1:, k] /= akk
        if sign == 1 and akk < 0: sign = 2
        elif sign == -1 and akk > 0: sign = 2
        elif sign == 0: sign = 1 if akk > 0 else (-1 if akk < 0 else 0)
    if sign not in (0, 1):
        return None
    x = np.array(b, float)
    for k in range(n):
        x[[k, tr[k]]] = x[[tr[k], k]]
    L = np.tril(A, -1) + np.eye(n)
    x = np.linalg.solve(L, x) if n else x
    d = np.diag(A)
    x = np.where(np.abs(d) > np.finfo(float).tiny, x / np.where(d == 0, 1, d), 0.0)
    x = np.linalg.solve(L.T, x) if n else x
    for k in reversed(range(n)):
        x[[k, tr[k]]] = x[[tr[k], k]]
    return x

def se3_exp(u):
    om, up = u[:3], u[3:]
    th = np.linalg.norm(om)
    O = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-5:
        R = np.eye(3) + O + O @ O; V = R
    else:
        R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th**2 * O @ O
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * O + (th - np.sin(th)) / th**3 * O @ O
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = V @ up
    return T

def lm(pb, max_it=50, delta=1.0):
    """g2o's OptimizationAlgorithmLevenberg over BlockSolver<6,3> + LinearSolverDense on the window `pb` (free points, no fixed vertex):
    returns the trace rows (lambda, chi2 as the loop sees it, accepted)."""
    F, L = len(pb["poses0"]), len(pb["points0"])
    T = [np.linalg.inv(P) for P in pb["poses0"]]          # world -> camera
    X = pb["points0"].astype(float).copy()
    ep, el, uv = pb["edge_pose"], pb["edge_point"], pb["edge_uv"]
    f, cx, cy = pb["focal"], pb["cx"], pb["cy"]
    def errs(T, X):
        out = []
        for e in range(len(ep)):
            pc = T[ep[e]][:3, :3] @ X[el[e]] + T[ep[e]][:3, 3]
            out.append(uv[e] - (f * pc[:2] / pc[2] + [cx, cy]))
        return np.array(out)
    def robust(T, X):
        c = (errs(T, X) ** 2).sum(1)
        return np.where(c <= delta**2, c, 2 * np.sqrt(c) * delta - delta**2).sum()
    x = np.zeros(6 * F + 3 * L)                            # the solver's x: survives failed solves
    lam, ni, trace = 0.0, 2.0, []
    for it in range(max_it):
        cur = robust(T, X)
        H = np.zeros((6 * F + 3 * L,) * 2); b = np.zeros(6 * F + 3 * L)
        for e in range(len(ep)):
            p, l = ep[e], el[e]
            R = T[p][:3, :3]; pc = R @ X[l] + T[p][:3, 3]
            xx, y, z = pc; z2 = z * z
            err = uv[e] - (f * pc[:2] / z + [cx, cy])
            chi = err @ err
            w = 1.0 if chi <= delta**2 else delta / np.sqrt(chi)
            Jx = -1.0 / z * np.array([[f, 0, -xx / z * f], [0, f, -y / z * f]]) @ R
            Jp = np.array([[xx * y / z2 * f, -(1 + xx * xx / z2) * f, y / z * f, -1.0 / z * f, 0, xx / z2 * f],
                           [(1 + y * y / z2) * f, -xx * y / z2 * f, -xx / z * f, 0, -1.0 / z * f, y / z2 * f]])
            J = np.zeros((2, 6 * F + 3 * L)); J[:, 6 * p:6 * p + 6] = Jp; J[:, 6 * F + 3 * l:6 * F + 3 * l + 3] = Jx
            H += w * J.T @ J; b -= w * J.T @ err
        if it == 0:
            lam = 1e-5 * np.abs(np.diag(H)).max(); ni = 2.0
        qmax, rho = 0, 0.0
        while True:
            Hd = H + lam * np.eye(len(H))
            Hpp, Hpl, Hll = Hd[:6 * F, :6 * F], Hd[:6 * F, 6 * F:], Hd[6 * F:, 6 * F:]
            Dinv = np.zeros_like(Hll)
            for l in range(L):
                Dinv[3 * l:3 * l + 3, 3 * l:3 * l + 3] = np.linalg.inv(Hll[3 * l:3 * l + 3, 3 * l:3 * l + 3])
            S = Hpp - Hpl @ Dinv @ Hpl.T
            g = b[:6 * F] - Hpl @ Dinv @ b[6 * F:]
            xp = eigen_ldlt(S, g)
            ok = xp is not None
            if ok:
                x = np.concatenate([xp, Dinv @ (b[6 * F:] - Hpl.T @ xp)])
            Tn = [se3_exp(x[6 * p:6 * p + 6]) @ T[p] for p in range(F)]
            Xn = X + x[6 * F:].reshape(-1, 3)
            chi_state = robust(Tn, Xn)
            temp = chi_state if ok else np.finfo(float).max
            with np.errstate(over="ignore"):
                rho = (cur - temp) / (x @ (lam * x + b) + 1e-3)
            acc = rho > 0 and np.isfinite(temp)
            trace.append((lam, temp, float(acc)))
            if acc:
                with np.errstate(over="ignore", invalid="ignore"):
                    alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha); ni = 2.0; T, X = Tn, Xn
            else:
                lam *= ni; ni *= 2
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        if qmax == 10 or rho == 0:
            break
    return np.array(trace)
