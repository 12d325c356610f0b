"""Minimal fake of `pydrake.all` for tests/test_leaf_system.py: LeafSystem / BasicVector / AbstractValue with the
call signatures the reference's controllers use (controllers/basic_controller.py:33-50,286-320), and a fake
MultibodyPlant whose joint numbering is BREADTH-FIRST (all abduction joints first, the older Drake order the
reference comments on at basic_controller.py:310-313) while the actuators follow the URDF transmission order."""
import numpy as np


class BasicVector:
    def __init__(self, n):
        self._v = np.zeros(int(n))

    def size(self):
        return self._v.size

    def SetFromVector(self, x):
        x = np.asarray(x, dtype=float).reshape(-1)
        assert x.size == self._v.size, "BasicVector size mismatch"
        self._v[:] = x

    def get_value(self):
        return self._v

    CopyToVector = get_value


class AbstractValue:
    def __init__(self, v):
        self._v = v

    @staticmethod
    def Make(v):
        return AbstractValue(v)

    def get_value(self):
        return self._v

    def set_value(self, v):
        self._v = v


class _InputPort:
    def __init__(self, system, index, name, kind, model):
        self.system, self.index, self.name, self.kind, self.model = system, index, name, kind, model

    def FixValue(self, context, value):
        if self.kind == "vector":
            bv = BasicVector(self.model.size())
            bv.SetFromVector(value)
            context.inputs[self.index] = bv
        else:
            context.inputs[self.index] = AbstractValue(value)

    def get_index(self):
        return self.index


class _OutputPort:
    def __init__(self, system, index, name, size, calc):
        self.system, self.index, self.name, self._size, self.calc = system, index, name, size, calc

    def size(self):
        return self._size

    def Eval(self, context):
        out = BasicVector(self._size)
        self.calc(context, out)
        return out.get_value().copy()


class Context:
    def __init__(self):
        self.inputs = {}
        self.time = 0.0

    def get_time(self):
        return self.time


class LeafSystem:
    def __init__(self):
        self._in, self._out = [], []

    def DeclareVectorInputPort(self, name, model_vector):
        p = _InputPort(self, len(self._in), name, "vector", model_vector)
        self._in.append(p)
        return p

    def DeclareAbstractInputPort(self, name, model_value):
        p = _InputPort(self, len(self._in), name, "abstract", model_value)
        self._in.append(p)
        return p

    def DeclareVectorOutputPort(self, name, model_vector, calc):
        p = _OutputPort(self, len(self._out), name, model_vector.size(), calc)
        self._out.append(p)
        return p

    def get_input_port(self, i):
        return self._in[i]

    def get_output_port(self, i):
        return self._out[i]

    def GetInputPort(self, name):
        return next(p for p in self._in if p.name == name)

    def GetOutputPort(self, name):
        return next(p for p in self._out if p.name == name)

    def CreateDefaultContext(self):
        return Context()

    def EvalVectorInput(self, context, i):
        return context.inputs[i]

    def EvalAbstractInput(self, context, i):
        return context.inputs[i]


class _AbstractOutputPort:
    def __init__(self, system, index, name, alloc, calc):
        self.system, self.index, self.name, self.alloc, self.calc = system, index, name, alloc, calc

    def Eval(self, context):
        out = self.alloc()
        self.calc(context, out)
        return out.get_value()


def _declare_abstract_output_port(self, name, alloc, calc):
    p = _AbstractOutputPort(self, len(self._out), name, alloc, calc)
    self._out.append(p)
    return p


LeafSystem.DeclareAbstractOutputPort = _declare_abstract_output_port
AbstractValue.get_mutable_value = AbstractValue.get_value


# geometry names the trunk planners touch for the visualiser output (planners/simple.py:24-33,126-139): containers only
class RigidTransform:
    def __init__(self):
        self._R, self._p = None, np.zeros(3)

    def set_rotation(self, r):
        self._R = r

    def set_translation(self, p):
        self._p = np.asarray(p, dtype=float).copy()

    def rotation(self):
        return self._R

    def translation(self):
        return self._p


class RollPitchYaw:
    """Drake's documented convention: R = Rz(yaw) Ry(pitch) Rx(roll); w_parent = E(rpy) rpyDt with
    E = [[cp cy, -sy, 0], [cp sy, cy, 0], [-sp, 0, 1]]  (written from the documentation, not from Drake sources)."""

    def __init__(self, x):
        if hasattr(x, "matrix"):
            R = x.matrix()
            self.rpy = np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0])),
                                 np.arctan2(R[1, 0], R[0, 0])])
        else:
            self.rpy = np.asarray(x, dtype=float).reshape(3).copy()

    def vector(self):
        return self.rpy

    def _E(self):
        r, p, y = self.rpy
        cp, sp, cy, sy = np.cos(p), np.sin(p), np.cos(y), np.sin(y)
        return np.array([[cp * cy, -sy, 0.0], [cp * sy, cy, 0.0], [-sp, 0.0, 1.0]])

    def CalcAngularVelocityInParentFromRpyDt(self, rpyDt):
        return self._E() @ np.asarray(rpyDt, dtype=float)

    def CalcRpyDtFromAngularVelocityInParent(self, w):
        return np.linalg.solve(self._E(), np.asarray(w, dtype=float))


class FramePoseVector:
    def __init__(self):
        self._d = {}

    def set_value(self, frame_id, X):
        self._d[frame_id] = X

    def value(self, frame_id):
        return self._d[frame_id]

    def clear(self):
        self._d.clear()


class _Joint:
    def __init__(self, vstart):
        self._vs = vstart

    def velocity_start(self):
        return self._vs


class FakePlant:
    """Floating base + 12 revolute joints.  `joint_names` is the canonical (leg-major) list; v-index of canonical
    joint j is 6 + order[j]; actuator k drives canonical joint act_joint[k]."""

    def __init__(self, joint_names, order, act_joint):
        self._names = list(joint_names)
        self._order = [int(x) for x in order]
        self._act = [int(x) for x in act_joint]

    def num_positions(self):
        return 19

    def num_velocities(self):
        return 18

    def num_actuators(self):
        return 12

    def GetJointByName(self, name):
        return _Joint(6 + self._order[self._names.index(name)])

    def MakeActuationMatrix(self):
        B = np.zeros((18, 12))
        for k, j in enumerate(self._act):
            B[6 + self._order[j], k] = 1.0
        return B


# ---- names the reference's controllers pull in through `from pydrake.all import *` (controllers/*.py)
from .autodiffutils import AutoDiffXd  # noqa: E402
from .mathprog import MathematicalProgram, OsqpSolver, GurobiSolver  # noqa: E402
from .refplant import JacobianWrtVariable, RefPlant, RotationMatrix  # noqa: E402


def jacobian(function, x):
    """pydrake.forwarddiff.jacobian as documented: seed x with unit derivatives, evaluate, stack the derivatives."""
    x = np.asarray(x, dtype=float)
    x_ad = np.empty(x.shape, dtype=object)
    for i in range(x.size):
        d = np.zeros(x.size); d[i] = 1.0
        x_ad.flat[i] = AutoDiffXd(x.flat[i], d)
    y_ad = np.asarray(function(x_ad), dtype=object)
    return np.vstack([y.derivatives() for y in y_ad.flat]).reshape(y_ad.shape + (-1,))


def ContinuousAlgebraicRiccatiEquation(A, B, Q, R):
    import scipy.linalg
    return scipy.linalg.solve_continuous_are(A, B, Q, R)
