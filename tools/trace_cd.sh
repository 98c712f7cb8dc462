#!/bin/bash
# conv_diff! / BDIM! kernels of a short run, per kernel variant: tools/trace_cd.sh <tag> [opts] [dtype]   (opts: WL_OPTS of steps.py)
set -e
TAG=${1:-cd}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WL_OPTS=${2:-}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/tools/steps.py 512 12 ${3:-f32} > $OUT/run.log 2>&1
python3 - $OUT/t_kernel_stats.csv <<'PY' > $ROOT/gpurun_out/trace_$TAG.txt
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if re.search(r"convdiff|bdim", n):
        m = re.search(r"k_convdiff3s<[^>]*>|k_convdiff3<[^>]*>|k_convdiff_xghost<[^>]*>|k_bdim2_busy<[^>]*>|op_bdim2<[^>]*>", n)
        print(f'{(m.group(0) if m else n[:60]):50s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
rm -f $OUT/t_kernel_trace.csv
cat $ROOT/gpurun_out/trace_$TAG.txt
