#!/bin/bash
# SQ activity counters of a short run, per kernel family: which unit a wavefront spends its cycles on / waiting for.
# (separate from every tracing run: --pmc with --kernel-trace only)   usage: tools/sq_counters.sh <tag> [dtype]
set -e
TAG=${1:-sq}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT -o a -- python3 $ROOT/tools/steps.py 512 6 ${2:-f32} > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES --kernel-trace --output-format csv -d $OUT -o b -- python3 $ROOT/tools/steps.py 512 6 ${2:-f32} > $OUT/b.log 2>&1
python3 - $OUT/a_counter_collection.csv $OUT/b_counter_collection.csv <<'PY' > $ROOT/gpurun_out/sq_$TAG.txt
import csv, re, sys, collections
FAM = [("conv_diff tile, predictor (FIN=1)", r"k_convdiff3s<.*, 1>"), ("conv_diff tile, corrector (FIN=2)", r"k_convdiff3s<.*, 2>"),
       ("pcg mult (7-point, R=2)", r"k_stencil7<\w+, 1, 2, wl::SrcArray.*op_pcg"), ("pcg update / direction / other rowvec", r"k_rowvec<.*op_pcg"),
       ("smoother (7-point, R=2)", r"k_stencil7<\w+, 0, 2, wl::SrcJacobi"), ("residual + div", r"ResidualDivEpi"),
       ("prolongate + increment", r"SrcProlong"), ("correct", r"k_correct3"), ("scale", r"k_scale_flat"), ("BDIM busy rows", r"k_bdim2_busy")]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f, tag in zip(sys.argv[1:], "ab"):
    rows = []
    for r in csv.DictReader(open(f)):
        for name, pat in FAM:
            if re.search(pat, r["Kernel_Name"]):
                rows.append((name, r["Counter_Name"], float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
                break
    longest = collections.defaultdict(int)
    for name, _, _, dur in rows:
        longest[name] = max(longest[name], dur)
    for name, cn, val, dur in rows:              # finest-level launches only: the coarser levels run the same kernels 8x shorter
        if dur * 4 > longest[name]:
            acc[name][cn + "@" + tag] += val
print("share of a wavefront's cycles (SQ_WAVE_CYCLES of the same pass = 1), finest-level launches of 6 steps at 512^3")
print(f"{'kernel family':42s} {'VALU':>6s} {'any inst':>8s} {'wait any':>8s} {'wait inst':>9s} {'LDS':>6s} {'scalar':>6s} {'VMEM':>6s}")
for name, _ in FAM:
    d = acc.get(name)
    if not d:
        continue
    def g(c, p):
        return d.get(c + "@" + p, 0.0)
    wa, wb = g("SQ_WAVE_CYCLES", "a"), g("SQ_WAVE_CYCLES", "b")
    if wa <= 0 or wb <= 0:
        continue
    print(f"{name:42s} {g('SQ_ACTIVE_INST_VALU','a')/wa:6.3f} {g('SQ_ACTIVE_INST_ANY','a')/wa:8.3f} {g('SQ_WAIT_ANY','a')/wa:8.3f} {g('SQ_WAIT_INST_ANY','a')/wa:9.3f} "
          f"{g('SQ_ACTIVE_INST_LDS','b')/wb:6.3f} {g('SQ_ACTIVE_INST_SCA','b')/wb:6.3f} {g('SQ_ACTIVE_INST_VMEM','b')/wb:6.3f}")
PY
rm -f $OUT/*_kernel_trace.csv $OUT/*_counter_collection.csv
cat $ROOT/gpurun_out/sq_$TAG.txt
