#!/bin/bash
# (dev) A/B of bench configurations across libraries built into build/dev/ (JSIM_LIB_OUT=... python .../build.py); run under gpurun from the repo root
# usage: ab3.sh "<configs>" lib1 lib2 ...  + T=13 at 256 and 4096 egos through horizon timing
cfgs="$1"; shift
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = ship ]; then unset JSIM_LIB_PATH; else export JSIM_LIB_PATH=$PWD/$lib; fi
  for c in $cfgs; do
    python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-respawn-start 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'config', $c, round(d['value']/1e6,3), 'M', d['config']['mean_active_set_iters'])"
  done
done
done
