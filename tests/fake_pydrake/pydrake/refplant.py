"""TEST-ONLY stand-in for the slice of pydrake's MultibodyPlant the reference's controllers call
(controllers/basic_controller.py:89-269).  The rigid-body quantities come from oracle/ (M, Cv, tau_g, frame positions,
Jacobians, bias accelerations) in Drake's documented conventions (world-frame base velocities [w; v], CalcBiasTerm without
gravity, CalcGravityGeneralizedForces = the force side).  The two autodiff recipes of the reference are served as
derivatives, NOT by handing over the oracle's own C / Jdot:
  * CalcBiasTerm on velocities seeded with derivatives (basic_controller.py:117-132): dCv/dv by central differences with
    step 1 -- exact for the quadratic form Cv;
  * CalcJacobianTranslationalVelocity on positions seeded with derivatives (basic_controller.py:198-220): dJ/dq by
    Richardson-extrapolated central differences of the oracle's foot Jacobian (error ~1e-11).
So a run of the reference's ControlLaw over this plant checks the oracle's Coriolis matrix and Jdot against the
reference's DEFINITIONS of them, and everything downstream against the reference's own arithmetic."""
import numpy as np

from oracle import oracle_py as orc
from .autodiffutils import join, split

FEET = {"LF_FOOT": 0, "RF_FOOT": 1, "LH_FOOT": 2, "RH_FOOT": 3}


class JacobianWrtVariable:
    kQDot, kV = 0, 1


class Frame:
    def __init__(self, name):
        self.name = name


class RotationMatrix:
    def __init__(self, R):
        self._R = np.array(R, dtype=float)

    def matrix(self):
        return self._R


class SpatialAcceleration:
    def __init__(self, c):
        self._c = np.array(c, dtype=float)

    def get_coeffs(self):
        return self._c


class _Pose:
    def __init__(self, R, p):
        self._R, self._p = RotationMatrix(R), np.array(p, dtype=float)

    def translation(self):
        return self._p

    def rotation(self):
        return self._R


class PlantContext:
    def __init__(self):
        self.q = np.zeros(19); self.q[0] = 1.0
        self.v = np.zeros(18)
        self.qD = self.vD = None        # derivative seeds (autodiff plant only)


class RefPlant:
    """Floating base + 12 joints.  Canonical joint j (leg-major) sits at position 7 + order[j] / velocity 6 + order[j]
    of the plant's own numbering and actuator k drives canonical joint act_joint[k] (defaults: identity) -- the freedom
    Drake has and the reference comments on (basic_controller.py:310-313).  The oracle is always asked in canonical
    order; every answer is re-indexed to the plant's numbering."""

    def __init__(self, model_name="mini_cheetah", body_frame="body", autodiff=False, order=None, act_joint=None,
                 backend="oracle"):
        self.model_name, self.body_frame_name, self.autodiff = model_name, body_frame, autodiff
        self.order = list(range(12)) if order is None else [int(x) for x in order]
        self.act_joint = list(range(12)) if act_joint is None else [int(x) for x in act_joint]
        self.pv = np.array(list(range(6)) + [6 + o for o in self.order])      # canonical i  <->  plant pv[i]
        self.pq = np.array(list(range(7)) + [7 + o for o in self.order])
        self.m = orc.model(model_name)
        self.backend = backend
        if backend == "energy":
            import energy_model as em           # tests/energy_model.py: the independent numpy derivation
            self._em, self._emodel = em, em.load(model_name)
        else:
            assert backend == "oracle"

    # ---- rigid-body terms in canonical order: oracle/ (default) or tests/energy_model.py ("energy": plain FK + Kane
    # projection, accelerations by Richardson-extrapolated differences along the exact flow; nothing shared with oracle/)
    def _dyn(self, q, v):
        if self.backend == "oracle":
            return orc.calc_dynamics(self.m, q, v)
        em, t = self._em, self._emodel
        rich = lambda f, h: (4.0 * f(0.5 * h) - f(h)) / 3.0
        return em.mass_matrix(t, q), rich(lambda h: em.bias_term(t, q, v, h), 2e-3), em.gravity_term(t, q)

    def _foot(self, q, v, foot):
        if self.backend == "oracle":
            return orc.foot_quantities(self.m, q, v, foot)
        em, t = self._em, self._emodel
        f = em.bodies(t, q)[1][foot]
        rich = lambda g, h: (4.0 * g(0.5 * h) - g(h)) / 3.0
        Jd = rich(lambda h: em.foot_jacobian_dot_fd(t, q, v, foot, h), 2e-3)
        return np.array(f["p"], dtype=float), np.array(f["J"], dtype=float), Jd @ v

    def _body(self, q, v):
        if self.backend == "oracle":
            return orc.body_quantities(self.m, q, v)
        # the task frame is the floating root itself: J = [I 0] in Drake's world-frame base velocities, so Jdot v = 0
        return self._em.quat_R(q[:4]), np.array(q[4:7], dtype=float), np.hstack([np.eye(6), np.zeros((6, 12))]), np.zeros(6)

    def _vec(self, x_c):                 # canonical 18-vector -> plant numbering
        out = np.empty_like(x_c); out[self.pv] = x_c
        return out

    def _cols(self, J_c):                # canonical [.., 18] -> plant numbering on the last-but-derivative axis 1
        out = np.empty_like(J_c); out[:, self.pv] = J_c
        return out

    # -- bookkeeping
    def CreateDefaultContext(self):
        return PlantContext()

    def ToAutoDiffXd(self):
        return RefPlant(self.model_name, self.body_frame_name, autodiff=True, order=self.order, act_joint=self.act_joint,
                        backend=self.backend)

    def num_positions(self):
        return 19

    def num_velocities(self):
        return 18

    def num_actuators(self):
        return 12

    def world_frame(self):
        return Frame("world")

    def GetFrameByName(self, name):
        if name != self.body_frame_name and name not in FEET:
            raise RuntimeError("no frame named %r" % name)
        return Frame(name)

    def SetPositions(self, ctx, q):      # the context keeps CANONICAL order
        val, D = split(q)
        ctx.q, ctx.qD = val[self.pq], (None if D is None else D[self.pq])

    def SetVelocities(self, ctx, v):
        val, D = split(v)
        ctx.v, ctx.vD = val[self.pv], (None if D is None else D[self.pv])

    def GetPositions(self, ctx):
        out = np.empty(19); out[self.pq] = ctx.q
        return out

    def GetVelocities(self, ctx):
        return self._vec(ctx.v)

    def GetJointByName(self, name):
        """The joint's place in the plant's own numbering (what make_leaf_system and tools/drake_crosscheck.py read)."""
        names = [l["joint"] for leg in orc.load_model_json(self.model_name)["legs"] for l in leg["links"]]
        j = names.index(name)

        class _J:
            def velocity_start(_s):
                return 6 + self.order[j]

            def position_start(_s):
                return 7 + self.order[j]
        return _J()

    def MakeActuationMatrix(self):
        B = np.zeros((18, 12))
        for k, j in enumerate(self.act_joint):
            B[6 + self.order[j], k] = 1.0
        return B

    def MapVelocityToQDot(self, ctx, v):
        """qdot = N(q) v for the quaternion floating base with world-frame angular velocity: quat' = 1/2 (0, w) * quat."""
        w, x, y, z = ctx.q[:4]
        v = np.asarray(v, dtype=float)
        om = v[:3]
        qd = np.zeros(19)
        qd[0] = -0.5 * (om[0] * x + om[1] * y + om[2] * z)
        qd[1:4] = 0.5 * (w * om + np.cross(om, [x, y, z]))
        qd[4:7] = v[3:6]
        qd[7:] = v[6:]          # joint rates keep their own (plant) numbering: position 7 + k <-> velocity 6 + k
        return qd

    def MapQDotToVelocity(self, ctx, qdot):
        """v = N+(q) qdot: (0, w) = 2 quat' * conj(quat) for the world-frame angular velocity (the inverse of the map above);
        position and joint rates pass through.  The PD law feeds it q - q_nom (basic_controller.py:343)."""
        w, x, y, z = ctx.q[:4]
        qd = np.asarray(qdot, dtype=float)
        a, b = qd[0], qd[1:4]
        vec = np.array([x, y, z])
        v = np.zeros(18)
        v[:3] = 2.0 * (-a * vec + w * b - np.cross(b, vec))
        v[3:6] = qd[4:7]
        v[6:] = qd[7:]
        return v

    # -- dynamics
    def CalcMassMatrixViaInverseDynamics(self, ctx):
        M = np.empty((18, 18)); M[np.ix_(self.pv, self.pv)] = self._dyn(ctx.q, ctx.v)[0]
        return M

    def CalcBiasTerm(self, ctx):
        Cv = self._dyn(ctx.q, ctx.v)[1]
        if ctx.vD is None:
            return self._vec(Cv)
        assert self.autodiff and ctx.qD is None
        dC = np.zeros((18, 18))
        for j in range(18):
            e = np.zeros(18); e[j] = 1.0
            dC[:, j] = 0.5 * (self._dyn(ctx.q, ctx.v + e)[1] - self._dyn(ctx.q, ctx.v - e)[1])
        return join(self._vec(Cv), self._vec(dC @ ctx.vD))

    def CalcGravityGeneralizedForces(self, ctx):
        return -self._vec(self._dyn(ctx.q, ctx.v)[2])   # the reference flips the sign (basic_controller.py:112)

    # -- frames
    def CalcRelativeTransform(self, ctx, frame_A, frame_B):
        assert frame_A.name == "world" and frame_B.name == self.body_frame_name
        R, p, _, _ = self._body(ctx.q, ctx.v)
        return _Pose(R, p)

    def CalcJacobianSpatialVelocity(self, ctx, wrt, frame, p_BoBp, frame_A, frame_E):
        assert wrt == JacobianWrtVariable.kV and frame.name == self.body_frame_name and not np.any(p_BoBp)
        return self._cols(self._body(ctx.q, ctx.v)[2])

    def CalcBiasSpatialAcceleration(self, ctx, wrt, frame, p_BoBp, frame_A, frame_E):
        assert wrt == JacobianWrtVariable.kV and frame.name == self.body_frame_name and not np.any(p_BoBp)
        return SpatialAcceleration(self._body(ctx.q, ctx.v)[3])

    def CalcPointsPositions(self, ctx, frame, p_BQ, frame_A):
        assert not np.any(p_BQ) and frame_A.name == "world"
        return self._foot(ctx.q, ctx.v, FEET[frame.name])[0].reshape(3, 1)

    def _J(self, q, v, foot):
        if self.backend == "energy":
            return np.array(self._em.bodies(self._emodel, q)[1][foot]["J"], dtype=float)
        return self._foot(q, v, foot)[1]

    def CalcJacobianTranslationalVelocity(self, ctx, wrt, frame, p_BoBp, frame_A, frame_E):
        assert wrt == JacobianWrtVariable.kV and not np.any(p_BoBp)
        foot = FEET[frame.name]
        J = self._J(ctx.q, ctx.v, foot)
        if ctx.qD is None:
            return self._cols(J)
        assert self.autodiff
        dJ = np.zeros((3, 18, 19))
        for k in range(19):
            e = np.zeros(19); e[k] = 1.0
            d = lambda h: (self._J(ctx.q + h * e, ctx.v, foot) - self._J(ctx.q - h * e, ctx.v, foot)) / (2.0 * h)
            h = 2e-3
            dJ[:, :, k] = (4.0 * d(0.5 * h) - d(h)) / 3.0          # Richardson: O(h^4)
        return join(self._cols(J), self._cols(dJ @ ctx.qD))

    def CalcBiasTranslationalAcceleration(self, ctx, wrt, frame, p_BoBp, frame_A, frame_E):
        assert wrt == JacobianWrtVariable.kV and not np.any(p_BoBp)
        return self._foot(ctx.q, ctx.v, FEET[frame.name])[2].reshape(3, 1)
