#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel family, launches and time split into 'large' dispatches
(>= 100 us: the finest multigrid level / full-field kernels) and 'small' ones (coarse levels, scalar epilogues).
usage: summarize_trace.py <kernel_trace.csv> [steps]"""
import collections
import csv
import re
import sys


def short(n):
    if "k_convdiff3" in n:
        return "conv_diff(lds)"
    m = re.search(r"(k_stencil7|k_rowvec|k_range_red|k_range|k_finalize|k_reduce_only|k_apply)<.*?(op_\w+?|red_\w+?)<", n)
    if m:
        lam = re.search(r"#(\d)\}", n)
        return f"{m.group(2)}{'#' + lam.group(1) if lam else ''}[{m.group(1)[2:]}]"
    if "k_pforce" in n:
        return "pforce"
    return re.sub(r"<.*", "", n)[:40]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for r in csv.DictReader(open(path)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[short(r["Kernel_Name"])]
        if d >= 100:
            a[0] += 1; a[1] += d
        else:
            a[2] += 1; a[3] += d
    tot_l = sum(a[1] for a in agg.values()); tot_s = sum(a[3] for a in agg.values())
    print(f"{'kernel':44s} {'large#':>7s} {'large ms':>9s} {'avg us':>8s} | {'small#':>7s} {'small ms':>9s} {'avg us':>7s}   (per {steps:g} step(s))")
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
        print(f"{k:44s} {a[0] / steps:7.1f} {a[1] / 1e3 / steps:9.3f} {a[1] / max(1, a[0]):8.1f} | {a[2] / steps:7.1f} {a[3] / 1e3 / steps:9.3f} {a[3] / max(1, a[2]):7.1f}")
    print(f"{'TOTAL':44s} {'':7s} {tot_l / 1e3 / steps:9.3f} {'':8s} | {'':7s} {tot_s / 1e3 / steps:9.3f}")


if __name__ == "__main__":
    main()
