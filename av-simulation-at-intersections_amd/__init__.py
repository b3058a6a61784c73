"""jsim-mpc: MI355X-native batched receding-horizon MPC (drop-in for the reference's lib.mpc path).

The directory name carries a hyphen (fixed by the build contract), so import it with
    importlib.import_module("av-simulation-at-intersections_amd")
"""
from . import synth  # noqa: F401
