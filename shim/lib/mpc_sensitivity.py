"""`lib.mpc_sensitivity` -> the MI355X drop-in `av-simulation-at-intersections_amd.mpc_sensitivity` (same public names as the reference's
main/lib/mpc_sensitivity.py: the controller class, the module constants, the exception type)."""
import importlib as _importlib
import os as _os
import sys as _sys

_REPO = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _REPO not in _sys.path:
    _sys.path.insert(0, _REPO)
_m = _importlib.import_module("av-simulation-at-intersections_amd.mpc_sensitivity")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
