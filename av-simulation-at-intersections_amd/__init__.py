"""jsim-mpc: MI355X-native batched receding-horizon MPC (drop-in for the reference's lib.mpc path).

The directory name carries a hyphen (fixed by the build contract), so import it with
    importlib.import_module("av-simulation-at-intersections_amd")

Nothing here falls back to a CPU implementation: `BatchedMPC` / `MPC` raise if libjsim_mpc.so (HIP,
gfx950) is missing or no HIP device is present.  Importing the package itself needs neither.
"""
from . import synth, config, sharding, vehicle, build  # noqa: F401
from .config import MPCConfig  # noqa: F401
from .vehicle import State, BicycleModelDimensions  # noqa: F401
from . import _cabi  # noqa: F401
from .batched import BatchedMPC  # noqa: F401
from .closed_loop import ClosedLoop, PreTick, ScriptedObstacles, ScenarioLoop, car_circles  # noqa: F401
from . import mpc  # noqa: F401
from .mpc import MPC, MAX_ACCEL, MAX_DECEL, MPCSolutionNotFoundException  # noqa: F401
from . import mpc_with_speed  # noqa: F401
from . import mpc_sensitivity  # noqa: F401
from . import mpc_jerk  # noqa: F401
from . import planner  # noqa: F401
from . import workloads  # noqa: F401
