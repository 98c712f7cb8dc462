#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel family, launches and time split into FINEST-LEVEL dispatches and
the rest (coarser multigrid levels, scalar epilogues).  Levels shrink by 8x in cells, so a dispatch of a family counts as
finest-level when it lasts longer than a quarter of the family's longest dispatch (variants on the finest level -- e.g.
the final, heavier pcg! update -- stay in the bucket; the 256^3 level, 8x shorter, does not).  Families whose longest
dispatch is below 60 us have no finest-level bucket (latency-bound helpers).
usage: summarize_trace.py <kernel_trace.csv> [steps]"""
import collections
import csv
import re
import sys


def short(n):
    if "k_convdiff3" in n:
        return "conv_diff(lds)"
    if "ResidualDivEpi" in n:   # residual! with z = div(u) formed on the fly (functor epilogue, not a lambda of op_residual)
        rr = re.search(r"k_stencil7<\w+, \d, (\d)", n)
        return f"op_residual+div[stencil7{',R=' + rr.group(1) if rr else ''}]"
    m = re.search(r"(k_stencil7|k_rowvec|k_range_red|k_range|k_finalize|k_reduce_only|k_apply)<.*?(op_\w+?|red_\w+?)<", n)
    if m:
        lam = re.search(r"#(\d)\}", n)
        rr = re.search(r"k_stencil7<\w+, \d, (\d)", n)
        return f"{m.group(2)}{'#' + lam.group(1) if lam else ''}[{m.group(1)[2:]}{',R=' + rr.group(1) if rr else ''}]"
    if "k_pforce" in n:
        return "pforce"
    return re.sub(r"<.*", "", n)[:40]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    durs = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        durs[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for k, v in durs.items():
        top = max(v)
        fin = [d for d in v if top >= 60 and d > 0.25 * top]
        rest = [d for d in v if not (top >= 60 and d > 0.25 * top)]
        rows.append((k, fin, rest))
    tot_f = sum(sum(f) for _, f, _ in rows)
    tot_r = sum(sum(r) for _, _, r in rows)
    print(f"{'kernel':52s} {'finest#':>8s} {'ms':>9s} {'avg us':>8s} {'min':>7s} {'max':>7s} | {'other#':>7s} {'ms':>8s} {'avg us':>7s}   (per {steps:g} step(s))")
    for k, f, r in sorted(rows, key=lambda x: -(sum(x[1]) + sum(x[2]))):
        fa = f"{sum(f) / len(f):8.1f} {min(f):7.1f} {max(f):7.1f}" if f else f"{'':8s} {'':7s} {'':7s}"
        ra = f"{sum(r) / len(r):7.1f}" if r else f"{'':7s}"
        print(f"{k:52s} {len(f) / steps:8.1f} {sum(f) / 1e3 / steps:9.3f} {fa} | {len(r) / steps:7.1f} {sum(r) / 1e3 / steps:8.3f} {ra}")
    print(f"{'TOTAL':52s} {'':8s} {tot_f / 1e3 / steps:9.3f} {'':24s} | {'':7s} {tot_r / 1e3 / steps:8.3f}")


if __name__ == "__main__":
    main()
