#!/bin/bash
set -o pipefail
cd /root/repo
JSIM_HIPCC_EXTRA="-DJSIM_DEV_NO_REG -DJPL_STAMPS" JSIM_LIB_OUT=/root/repo/build/dev/libjsim_stamps.so python av-simulation-at-intersections_amd/build.py 2>&1 | grep -v "^/opt" | tail -5 &&
JSIM_HIPCC_EXTRA="-DJSIM_DEV_NO_REG" JSIM_LIB_OUT=/root/repo/build/dev/libjsim_noreg.so python av-simulation-at-intersections_amd/build.py 2>&1 | grep -v "^/opt" | tail -5
