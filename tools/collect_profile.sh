#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box (run from the repo root under gpurun):
#   tools/collect_profile.sh TAG [bench.py args...]
# One kernel-trace/stats run and four separate PMC passes (never combined with other trace domains), outputs under
# gpurun_out/TAG/; tools/pmc_summary.py then condenses them into profiles/.
set -e -o pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="${JSIM_PROFILE_ARGS:---steps 100 --warmup 10} --no-cpu-baseline --no-respawn-start $*"   # (one run of the workload: the dominant kernel's launches are then all of one kind)
cd /tmp
run() { # name, rocprof args...
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o runc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
    echo "$name done"
}
run trace --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq1 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VALU
run sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
cd "$ROOT"
python3 tools/pmc_summary.py "$TAG"
