#!/bin/bash
# planner dev loop on the GPU box: stamps build + plain build, timing and the planner tests
mkdir -p gpurun_out/r3p
JSIM_LIB_PATH=$PWD/build/dev/libjsim_stamps.so python tools/dev/plan48.py > gpurun_out/r3p/stamps.txt 2>&1 || { tail -20 gpurun_out/r3p/stamps.txt; exit 1; }
grep "jpl\|routes" gpurun_out/r3p/stamps.txt
JSIM_LIB_PATH=$PWD/build/dev/libjsim_noreg.so python tools/dev/plan48.py > gpurun_out/r3p/plan48.txt 2>&1 || { tail -20 gpurun_out/r3p/plan48.txt; exit 1; }
grep "routes" gpurun_out/r3p/plan48.txt
JSIM_LIB_PATH=$PWD/build/dev/libjsim_noreg.so timeout -k 10 240 python -m pytest tests/test_planner.py -m gpu -x -q -k "hip_planner or dropin" > gpurun_out/r3p/pytest.log 2>&1
tail -5 gpurun_out/r3p/pytest.log
