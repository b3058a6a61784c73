"""`lib.mp_search_ww_generic` -> the MI355X route planner's drop-in class (same constructor and run() as the reference's
main/lib/mp_search_ww_generic.py:26-58,136-140; the search runs on the GPU through jsim_plan_routes)."""
import importlib as _importlib
import os as _os
import sys as _sys

_REPO = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _REPO not in _sys.path:
    _sys.path.insert(0, _REPO)
_m = _importlib.import_module("av-simulation-at-intersections_amd.planner")
MotionPrimitiveSearch = _m.MotionPrimitiveSearch
NodeType = tuple
