#!/bin/bash
# rocprofv3 record of the bench command on the GPU box: kernel trace + stats, then HBM traffic from two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE).  usage: bash tools/profile.sh <tag e.g. r02a>   (outputs under gpurun_out/)
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.log
python3 $ROOT/profiles/summarize_trace.py $OUT/trace_kernel_trace.csv 27 > $OUT/${TAG}_kernel_trace_summary_512_f32.txt
cp $OUT/trace_kernel_stats.csv $OUT/${TAG}_kernel_stats_512_f32.csv
PMCB="python3 $ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- $PMCB > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- $PMCB > $OUT/bench_write.json 2> $OUT/write.log
cd $ROOT
python3 profiles/parse_pmc.py $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv "512^3/f32" > $OUT/${TAG}_pmc_traffic_512_f32.txt
cp profiles/traffic.json $OUT/traffic.json
head -30 $OUT/${TAG}_kernel_trace_summary_512_f32.txt
cat $OUT/${TAG}_pmc_traffic_512_f32.txt
# keep the merged output small: the raw traces stay on the box
rm -f $OUT/trace_kernel_trace.csv $OUT/fetch_kernel_trace.csv $OUT/write_kernel_trace.csv $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv
