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
        self.rotation, self.translation = None, np.zeros(3)

    def set_rotation(self, r):
        self.rotation = r

    def set_translation(self, p):
        self.translation = np.asarray(p, dtype=float).copy()


class RollPitchYaw:
    def __init__(self, rpy):
        self.rpy = np.asarray(rpy, dtype=float).copy()

    def vector(self):
        return self.rpy


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
