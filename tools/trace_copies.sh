#!/bin/bash
# which non-wl kernels (blit copies, fills, torch kernels) run inside steady time steps, and for how long:
# tools/trace_copies.sh <tag> [size]
set -e
TAG=${1:-cp}; SIZE=${2:-512}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $ROOT/tools/steps.py $SIZE 14 > $OUT/run.log 2>&1
python3 - $OUT/t_kernel_trace.csv <<'PY' > $ROOT/gpurun_out/trace_$TAG.txt
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps end with the CFL reduction kernel (op_cfl ... finalize / k_reduce): take the last 6 op_cfl rowvec launches as step marks
marks = [i for i, r in enumerate(rows) if "op_cfl" in r[2] and "k_rowvec" in r[2]]
lo, hi = marks[-7], marks[-1]
nsteps = 6
sel = rows[lo + 1:hi + 1]
span = (sel[-1][1] - sel[0][0]) / 1e6 / nsteps
busy = sum(e - s for s, e, _ in sel) / 1e6 / nsteps
print(f"{nsteps} steady steps: {span:.3f} ms per step wall, {busy:.3f} ms in kernels, {len(sel) / nsteps:.1f} kernels per step")
acc = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    if "wl::" in n:
        continue
    acc[n[:70]][0] += 1; acc[n[:70]][1] += e - s
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{c / nsteps:7.1f} per step  {t / 1e3 / nsteps:9.1f} us per step   {n}")
gaps = sorted(((sel[i + 1][0] - sel[i][1]) / 1e3, sel[i][2][:50], sel[i + 1][2][:50]) for i in range(len(sel) - 1))
print("largest gaps (us):")
for g in gaps[-8:]:
    print(f"  {g[0]:8.1f}  after {g[1]}  before {g[2]}")
PY
rm -rf $OUT
cat $ROOT/gpurun_out/trace_$TAG.txt
