#!/bin/bash
# kernel trace + stats of one bench.py command line: tools/trace_cmd.sh <tag> <bench args...>  -> gpurun_out/trace_<tag>/
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/trace.log
python3 $ROOT/profiles/summarize_trace.py $OUT/t_kernel_trace.csv ${STEPS_IN_TRACE:-30} > $OUT/summary.txt || true
rm -f $OUT/t_kernel_trace.csv
