"""MI355X-native batched whole-body-QP controller (one hot path of vincekurtz/quadruped_drake).

Product code: csrc/ (HIP kernels + C ABI, include/wbc.h), controller.py (host-side mirror of the
reference's IDController / MPTCController interface), planners.py (mirror of the reference's trunk planners: the
callers that feed the path), trajectory.py (trunk_state_t decode, device-side target lookup), workloads.py (synthetic
batches of BASELINE.json's configs), stats.py (multi-GPU shard + RCCL statistics reduce)."""
from .controller import IDController, MPTCController, PCController, CLFController, BatchedController, SolverError, IllConditionedWarning, pack_trunk_input, load_model, make_leaf_system  # noqa
from . import workloads  # noqa
from . import lcm_io  # noqa
from .pd import BasicController  # noqa
from .planners import BasicTrunkPlanner, TowrTrunkPlanner, scenario_targets, scenario_trajectory, unpack_trunk_input  # noqa
