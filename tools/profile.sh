#!/bin/bash
# rocprofv3 record of the bench command on the GPU box: kernel trace + stats, then HBM traffic from two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE).  Rewrites profiles/traffic.json (entries of this case + the tag they came from), which
# bench.py's `roofline.traffic` quotes -- copy gpurun_out/prof_<tag>/traffic.json back over profiles/traffic.json.
# usage: bash tools/profile.sh <tag e.g. r03a> [case: f32 (default: C3 512^3 f32 sphere) | f64 (C5: 512^3 f64 torus) | f64sphere]
set -e
TAG=${1:-r03}
CASE=${2:-f32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
case $CASE in
  f32) ARGS=""; LABEL="512_f32"; KEY="512^3/f32" ;;
  f64) ARGS="--dtype f64 --body donut"; LABEL="512_f64_donut"; KEY="512^3/f64/donut" ;;
  f64sphere) ARGS="--dtype f64"; LABEL="512_f64"; KEY="512^3/f64" ;;
  *) echo "unknown case $CASE"; exit 2 ;;
esac
OUT=$ROOT/gpurun_out/prof_${TAG}_$CASE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline $ARGS > $OUT/${TAG}_bench_$LABEL.json 2> $OUT/trace.log
python3 $ROOT/profiles/summarize_trace.py $OUT/trace_kernel_trace.csv 27 > $OUT/${TAG}_kernel_trace_summary_$LABEL.txt
cp $OUT/trace_kernel_stats.csv $OUT/${TAG}_kernel_stats_$LABEL.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- python3 $ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- python3 $ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline $ARGS > $OUT/bench_write.json 2> $OUT/write.log
cd $ROOT
python3 profiles/parse_pmc.py $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv "$KEY" $TAG > $OUT/${TAG}_pmc_traffic_$LABEL.txt
cp profiles/traffic.json $OUT/traffic.json
head -40 $OUT/${TAG}_kernel_trace_summary_$LABEL.txt
cat $OUT/${TAG}_pmc_traffic_$LABEL.txt
# keep the merged output small: the raw traces stay on the box
rm -f $OUT/trace_kernel_trace.csv $OUT/fetch_kernel_trace.csv $OUT/write_kernel_trace.csv $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv
