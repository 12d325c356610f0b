"""TEST-ONLY stand-in for the slice of pydrake.solvers the reference's controllers call
(controllers/inverse_dynamics_controller.py:25-101,199-225; mptc_controller.py:30-57; pc_controller.py:14-40,229-237;
clf_controller.py:15-45,206): a MathematicalProgram that RECORDS numeric costs / constraints with Drake's documented
meaning (AddQuadraticCost(Q, b, x) = 1/2 x'Qx + b'x; AddLinearConstraint lb <= A x <= ub; ...), and a solver object.

The solver is NOT OSQP (absent from this image): `OsqpSolver.Solve` solves the recorded QP exactly (dense KKT active
set in extended precision, written here, sharing nothing with oracle/ or the kernels) under this repository's
documented selection rule for the reference's non-strictly-convex QPs -- + 1/2 eps2 |x_reg|^2 on the variables named
in `OsqpSolver.tiebreak`.  What a run through this module pins is therefore the arithmetic of the
reference's ControlLaw code (targets, gains, task-space terms, QP assembly, logging); OSQP's own selection among the
optimal set stays unpinned, and is stated so wherever these fixtures are used."""
import numpy as np


class Variable:
    def __init__(self, name, index):
        self.name, self.index = name, index

    def __repr__(self):
        return "Variable(%s)" % self.name

    def _expr(self):
        return Expression({(self.index,): 1.0})

    def __mul__(self, o):
        return self._expr() * o

    __rmul__ = __mul__

    def __add__(self, o):
        return self._expr() + o

    __radd__ = __add__


class Expression:
    """polynomial of degree <= 2: {(): c0, (i,): ci, (i, j): cij with i <= j}"""

    def __init__(self, terms=None):
        self.terms = dict(terms or {})

    @staticmethod
    def of(o):
        if isinstance(o, Expression):
            return o
        if isinstance(o, Variable):
            return o._expr()
        return Expression({(): float(o)})

    def __add__(self, o):
        o = Expression.of(o)
        t = dict(self.terms)
        for k, c in o.terms.items():
            t[k] = t.get(k, 0.0) + c
        return Expression(t)

    __radd__ = __add__

    def __mul__(self, o):
        o = Expression.of(o)
        t = {}
        for ka, ca in self.terms.items():
            for kb, cb in o.terms.items():
                k = tuple(sorted(ka + kb))
                assert len(k) <= 2, "only quadratic expressions"
                t[k] = t.get(k, 0.0) + ca * cb
        return Expression(t)

    __rmul__ = __mul__


def _idx(vars_):
    return [v.index for v in np.asarray(vars_, dtype=object).reshape(-1)]


class MathematicalProgram:
    def __init__(self):
        self.names = []                 # per scalar variable: the name given to NewContinuousVariables
        self.quad = []                  # (idx, Q, b):   1/2 x'Qx + b'x
        self.eq = []                    # (idx, A, b):   A x == b
        self.ineq = []                  # (idx, A, lb, ub)

    def num_vars(self):
        return len(self.names)

    def NewContinuousVariables(self, rows, *args):
        cols, name = (None, args[0]) if len(args) == 1 else (int(args[0]), args[1])
        n = int(rows) * (cols or 1)
        out = np.empty(n, dtype=object)
        for i in range(n):
            out[i] = Variable("%s(%d)" % (name, i), len(self.names))
            self.names.append(name)
        return out if cols is None else out.reshape(int(rows), cols)

    def AddQuadraticCost(self, Q, b, vars, is_convex=None):
        Q = np.asarray(Q, dtype=float); b = np.asarray(b, dtype=float).reshape(-1)
        idx = _idx(vars)
        assert Q.shape == (len(idx), len(idx)) and b.size == len(idx)
        self.quad.append((idx, Q, b))

    def AddLinearCost(self, a, b=0.0, vars=None):
        idx = _idx(vars)
        a = np.asarray(a, dtype=float).reshape(-1)
        assert a.size == len(idx)
        self.quad.append((idx, np.zeros((len(idx), len(idx))), a))

    def AddCost(self, e):
        e = np.asarray(e, dtype=object).reshape(-1)
        assert e.size == 1
        e = Expression.of(e[0])
        n = self.num_vars()
        Q = np.zeros((n, n)); b = np.zeros(n)
        for k, c in e.terms.items():
            if len(k) == 1:
                b[k[0]] += c
            elif len(k) == 2:               # c x_i x_j  ->  1/2 x'Qx with Q_ij = Q_ji = c (i != j), Q_ii = 2c
                Q[k[0], k[1]] += c; Q[k[1], k[0]] += c
        self.quad.append((list(range(n)), Q, b))

    def AddLinearEqualityConstraint(self, Aeq, beq, vars):
        idx = _idx(vars)
        A = np.asarray(Aeq, dtype=float).reshape(-1, len(idx)); b = np.asarray(beq, dtype=float).reshape(-1)
        assert A.shape[0] == b.size
        self.eq.append((idx, A, b))

    def AddLinearConstraint(self, A, lb, ub, vars):
        idx = _idx(vars)
        A = np.asarray(A, dtype=float).reshape(-1, len(idx))
        lb = np.asarray(lb, dtype=float).reshape(-1); ub = np.asarray(ub, dtype=float).reshape(-1)
        assert A.shape[0] == lb.size == ub.size
        self.ineq.append((idx, A, lb, ub))

    def assemble(self):
        """-> P, c, Aeq, beq, Ain, bin  (1/2 x'Px + c'x,  Aeq x = beq,  Ain x <= bin) over all variables."""
        n = self.num_vars()
        P = np.zeros((n, n)); c = np.zeros(n)
        for idx, Q, b in self.quad:
            P[np.ix_(idx, idx)] += 0.5 * (Q + Q.T)
            c[idx] += b
        Ae, be, Ai, bi = [], [], [], []
        for idx, A, b in self.eq:
            R = np.zeros((A.shape[0], n)); R[:, idx] = A
            Ae.append(R); be.append(b)
        for idx, A, lb, ub in self.ineq:
            R = np.zeros((A.shape[0], n)); R[:, idx] = A
            for r in range(A.shape[0]):
                if np.isfinite(ub[r]):
                    Ai.append(R[r]); bi.append(ub[r])
                if np.isfinite(lb[r]):
                    Ai.append(-R[r]); bi.append(-lb[r])
        z = lambda rows: np.array(rows).reshape(-1, n) if len(rows) else np.zeros((0, n))
        return (P, c, np.vstack(Ae) if Ae else np.zeros((0, n)), np.concatenate(be) if be else np.zeros(0),
                z(Ai), np.array(bi, dtype=float))


LD = np.longdouble


def _solve_ld(K, r):
    """Gaussian elimination with partial pivoting in extended precision (numpy's LAPACK path is double only)."""
    K = np.array(K, dtype=LD); r = np.array(r, dtype=LD)
    n = r.size
    for k in range(n):
        p = k + int(np.argmax(np.abs(K[k:, k])))
        if K[p, k] == 0:
            raise np.linalg.LinAlgError("singular KKT matrix")
        if p != k:
            K[[k, p]] = K[[p, k]]; r[[k, p]] = r[[p, k]]
        f = K[k + 1:, k] / K[k, k]
        K[k + 1:, k:] -= f[:, None] * K[k, k:][None, :]
        r[k + 1:] -= f * r[k]
    x = np.zeros(n, dtype=LD)
    for k in range(n - 1, -1, -1):
        x[k] = (r[k] - K[k, k + 1:] @ x[k + 1:]) / K[k, k]
    return x


def solve_qp(P, c, Aeq, beq, Ain, bin_, tol=1e-13, maxit=400):
    """Strictly convex (after the tie-break) QP by the dual active-set idea written directly on dense KKT systems
    (every step a fresh extended-precision solve; no factor updates, nothing shared with oracle/):
      state: working set W, x optimal on W (Aeq x = beq, Ain_W x = bin_W) with multipliers lam >= 0;
      pick the most violated row p; along  P dx + Aeq' dm + Ain_W' dl = -n_p,  Aeq dx = 0,  Ain_W dx = 0  the violation
      shrinks at rate -n_p'dx and the multipliers move by dl per unit of p's multiplier; step to the nearer of
      "p satisfied" (add p) and "a multiplier reaches zero" (drop that row, keep going with p).
    Returns (x, W, lam, iterations); the caller-visible guarantee is the KKT check at the end."""
    n, me, mi = c.size, beq.size, bin_.size
    P = np.array(P, dtype=LD); A = np.array(Ain, dtype=LD); b = np.array(bin_, dtype=LD)

    def kkt(W, rhs_top):
        m = me + len(W)
        K = np.zeros((n + m, n + m), dtype=LD); r = np.zeros(n + m, dtype=LD)
        K[:n, :n] = P; r[:n] = rhs_top
        K[n:n + me, :n] = Aeq; K[:n, n:n + me] = Aeq.T
        if W:
            K[n + me:, :n] = A[W]; K[:n, n + me:] = A[W].T
        return K, r

    K, r = kkt([], -np.array(c, dtype=LD)); r[n:n + me] = beq
    s = _solve_ld(K, r)
    x = s[:n]
    W, lam = [], np.zeros(0, dtype=LD)
    xs = lambda: 1.0 + float(np.abs(x).max())
    for it in range(maxit):
        viol = A @ x - b if mi else np.zeros(0, dtype=LD)
        if W:
            viol[W] = -np.inf
        if not mi or viol.max() <= tol * xs():
            break
        p = int(np.argmax(viol)); up = LD(0)
        while True:
            K, r = kkt(W, -A[p])
            d = _solve_ld(K, r)
            dx, dl = d[:n], d[n + me:]
            rate = -(A[p] @ dx)                              # >= 0; 0: n_p depends on the working rows
            vp = A[p] @ x - b[p]
            t2 = vp / rate if rate > 1e-22 * (A[p] @ A[p]) else np.inf
            t1, j1 = np.inf, -1
            for j in range(len(W)):
                if dl[j] < 0 and -lam[j] / dl[j] < t1:
                    t1, j1 = -lam[j] / dl[j], j
            if not np.isfinite(min(t1, t2)):
                raise RuntimeError("infeasible QP")
            if t2 <= t1:                                     # full step: p joins the working set
                x = x + t2 * dx; lam = np.append(lam + t2 * dl, up + t2); W.append(p)
                break
            if np.isfinite(t2):                              # partial step, then drop the blocking row
                x = x + t1 * dx
            lam = lam + t1 * dl; up = up + t1
            W.pop(j1); lam = np.delete(lam, j1)
    else:
        raise RuntimeError("active-set iteration did not settle")
    # KKT certificate (what makes the answer THE answer, whatever path led here)
    g = P @ x + c
    K, r = kkt(W, -g)
    mult = _solve_ld(K, r)                                   # [0; m; lam] of  g + Aeq' m + Ain_W' lam = 0
    assert float(np.abs(mult[:n]).max()) < 1e-9 * xs(), "stationarity"
    lam = mult[n + me:]
    assert not W or float(lam.min()) > -1e-9 * (1.0 + float(np.abs(lam).max())), "dual feasibility"
    assert not mi or float((A @ x - b).max()) < 1e-10 * xs(), "primal feasibility"
    assert float(np.abs(Aeq @ x - beq).max(initial=0.0)) < 1e-10 * xs(), "equalities"
    return np.array(x, dtype=float), list(W), np.array(lam, dtype=float), it + 1


class _Details:
    def __init__(self, primal_res):
        self.primal_res = primal_res


class MathematicalProgramResult:
    def __init__(self, ok, x, details):
        self._ok, self.x, self._details = ok, x, details

    def is_success(self):
        return self._ok

    def GetSolution(self, vars):
        v = np.asarray(vars, dtype=object)
        out = np.array([self.x[e.index] for e in v.reshape(-1)])
        # pydrake hands a column of variables back as a 1-D vector (the reference relies on it: mptc_controller.py:302-305)
        return out if (v.ndim == 1 or 1 in v.shape) else out.reshape(v.shape)

    def get_solver_details(self):
        return self._details


class OsqpSolver:
    """See the module docstring: exact solve + this repository's tie-break, NOT OSQP."""
    eps2 = 1e-8
    tiebreak = ("tau", "f_", "delta")      # names (prefixes) of the variables that get + 1/2 eps2 x_i^2
    last = None                            # the last assembled QP and its solution, for the fixture generator

    def Solve(self, prog, initial_guess=None, solver_options=None):
        P, c, Aeq, beq, Ain, bin_ = prog.assemble()
        reg = np.array([1.0 if any(nm.startswith(p) for p in self.tiebreak) else 0.0 for nm in prog.names])
        # the slack with its own quadratic cost (CLF's w_delta delta^2, clf_controller.py:206) is strictly convex already
        for i, nm in enumerate(prog.names):
            if nm.startswith("delta") and P[i, i] > 0:
                reg[i] = 0.0
        try:
            x, W, lam, it = solve_qp(P + self.eps2 * np.diag(reg), c, Aeq, beq, Ain, bin_)
            ok = True
        except (RuntimeError, np.linalg.LinAlgError, AssertionError):
            x, W, lam, it, ok = np.zeros(c.size), [], np.zeros(0), 0, False
        res = float(np.abs(Aeq @ x - beq).max(initial=0.0))
        OsqpSolver.last = dict(P=P, c=c, Aeq=Aeq, beq=beq, Ain=Ain, bin=bin_, reg=reg, x=x, active=W, lam=lam, iters=it,
                               names=list(prog.names))
        return MathematicalProgramResult(ok, x, _Details(res))


GurobiSolver = OsqpSolver
