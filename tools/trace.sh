#!/bin/bash
# quick kernel trace of N steady steps (no PMC passes): bash tools/trace.sh <tag> [steps] [size]  -> gpurun_out/trace_<tag>.txt
set -e
TAG=${1:-t}; STEPS=${2:-16}; SIZE=${3:-512}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $ROOT/tools/steps.py $SIZE $STEPS > $OUT/run.log 2>&1
python3 $ROOT/profiles/summarize_trace.py $OUT/t_kernel_trace.csv $STEPS > $ROOT/gpurun_out/trace_$TAG.txt
rm -rf $OUT
head -${4:-34} $ROOT/gpurun_out/trace_$TAG.txt
