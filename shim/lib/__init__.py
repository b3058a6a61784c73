"""`lib` shim: puts the MI355X controllers behind the names the reference's scenario scripts import.

The scripts run with CWD = main/scenarios, do `sys.path.append('..')` and then e.g.
`from lib.mpc import MPC, MAX_ACCEL` (main/scenarios/mpc_intersection.py:5-6,20).  Put THIS directory's parent
(`<repo>/shim`) on sys.path ahead of that (PYTHONPATH=<repo>/shim python mpc_intersection.py) and

  * `lib.mpc`, `lib.mpc_with_speed`, `lib.mpc_sensitivity`, `lib.mpc_jerk` and `lib.mp_search_ww_generic` (the planner class the
    scripts construct before their loop, main/scenarios/mpc_intersection.py:17,63-64) resolve to the modules in this directory, which
    re-export the drop-ins of the `av-simulation-at-intersections_amd` package (the hyphenated directory name cannot be
    imported with a plain `import` statement, hence these files);
  * every other submodule -- `lib.simulation`, `lib.car_dimensions`, `lib.trajectories`, `lib.motion_primitive`, `lib.scenario`, the plotting --
    still resolves to the reference's own `lib` package: `pkgutil.extend_path` appends every other `lib` directory found
    on sys.path (the script's '..' included, which it appends before its first `from lib...` import) to this package's
    search path, behind this directory.

Nothing of the reference is copied or imported here."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
