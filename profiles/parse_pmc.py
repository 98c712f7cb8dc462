#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs) into per-launch HBM
traffic for the finest-level kernels, and update profiles/traffic.json (read by bench.py's roofline leg).

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md section "HBM" prescribes:
  * FETCH_SIZE / WRITE_SIZE are in KiB;
  * on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read -> doubled.  Calibrated here on
    kernels of this library whose byte counts are known exactly (pcg_direction: reads 2 arrays, writes 1;
    pcg_init: reads 2, writes 2): corrected reads land within +4..8 % of the known counts, writes are exact.

usage: parse_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <case key e.g. 512^3/f32> [profile tag e.g. r03a]
"""
import collections
import csv
import json
import os
import re
import sys

LAMBDA = {  # (operator, kernel kind, lambda ordinal) -> libwlhip kernel class
    ("op_pcg", "k_range_red", "#1"): "pcg_init", ("op_pcg", "k_range_red", "#2"): "pcg_mult_dot",
    ("op_pcg", "k_range_red", "#3"): "pcg_update", ("op_pcg", "k_range", "#1"): "pcg_direction",
    ("op_increment", "k_range", "#1"): "increment", ("op_jacobi", "k_range", "#1"): "jacobi",
    ("op_residual", "k_range_red", "#1"): "residual", ("op_bdim2", "k_range", "#1"): "bdim",
    ("op_correct", "k_range", "#1"): "correct", ("op_div", "k_range", "#1"): "div",
}


S7 = {"op_pcg": "pcg_mult_dot", "op_increment": "increment", "op_residual": "residual", "op_smooth_fused": "smooth",
      "op_prolong_increment_fused": "prolongate"}


ROWVEC = {("op_pcg", "#1"): "pcg_init", ("op_pcg", "#2"): "pcg_update", ("op_pcg", "#3"): "pcg_direction",
          ("op_bdim2", "#1"): "bdim", ("op_scale_all", "#1"): "scale"}


def classify(name):
    if "k_convdiff3" in name:
        return "conv_diff"
    if "k_correct3" in name:
        return "correct"
    if "k_measure_fill" in name:
        return "measure_fill"
    if "k_measure_rows" in name:
        return "measure_rows"
    if "k_scale_flat" in name:
        return "scale"
    if "k_rowvec" in name:
        m = re.search(r"(op_\w+?)<", name)
        o = re.search(r"#(\d)\}", name)
        return ROWVEC.get((m.group(1), "#" + o.group(1))) if m and o else None
    if "k_stencil7" in name:
        if "ResidualDivEpi" in name:
            return "residual"
        m = re.search(r"(op_\w+?)<", name)
        return S7.get(m.group(1)) if m else None
    m = re.search(r"(k_range_red|k_range)<.*?(op_\w+?)<", name)
    o = re.search(r"#(\d)\}", name)
    if not m or not o:
        return None
    return LAMBDA.get((m.group(2), m.group(1), "#" + o.group(1)))


def collect(path, counter):
    """mean counter value per FINEST-LEVEL dispatch of each kernel class.  The multigrid levels shrink by 8x in cells, so
    a dispatch belongs to the finest level when its value exceeds a quarter of the class maximum; variants of one class on
    that level (pcg_update: 5 non-final iterations at 3T and the final one at 6T per pcg! call) all pass, i.e. the mean is
    over the SAME launch mix bench.py times.  Also returns (count, min, max) per class."""
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = classify(r["Kernel_Name"])
        if k:
            per[k].append(float(r["Counter_Value"]) * 1024.0)
    out, info = {}, {}
    for k, v in per.items():
        top = max(v)
        fin = [x for x in v if x > 0.25 * top]
        out[k] = sum(fin) / len(fin)
        info[k] = (len(fin), min(fin), max(fin))
    return out, info


def csrc_digest():
    """sha1 over the kernel sources (waterlily_amd/csrc/*, include/wlhip.h): identifies the build a profile was taken on"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    d = os.path.join(root, "waterlily_amd", "csrc")
    for f in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "wlhip.h")]:
        p = os.path.join(d, f)
        if os.path.isfile(p):
            h.update(f.encode() + b"\0" + open(p, "rb").read())
    return h.hexdigest()


def main():
    fpath, wpath, tag = sys.argv[1:4]
    source = sys.argv[4] if len(sys.argv) > 4 else None
    rd, ri = collect(fpath, "FETCH_SIZE")
    rd = {k: 2.0 * v for k, v in rd.items()}
    wr, wi = collect(wpath, "WRITE_SIZE")
    here = os.path.dirname(os.path.abspath(__file__))
    tfile = os.path.join(here, "traffic.json")
    data = json.load(open(tfile)) if os.path.exists(tfile) else {}
    print(f"HBM traffic per finest-level launch, {tag} (FETCH_SIZE doubled per MI355X_MICROARCH.md; mean over the launch mix of the run)")
    for k in sorted(set(rd) | set(wr)):
        tot = rd.get(k, 0.0) + wr.get(k, 0.0)
        data[f"{k}@{tag}"] = tot
        n, lo, hi = ri.get(k, (0, 0, 0))
        spread = f"  [{n} launches, reads {2 * lo / 1e9:.2f}..{2 * hi / 1e9:.2f} GB]" if n else ""
        print(f"{k:14s} read(corrected) {rd.get(k, 0) / 1e9:7.3f} GB  write {wr.get(k, 0) / 1e9:7.3f} GB  total {tot / 1e9:7.3f} GB/launch{spread}")
    if source:   # which profile the entries of this case came from (bench.py prints it next to `traffic`)
        src = data.get("_source", {})
        src[tag] = source
        data["_source"] = src
        # ... and of which kernel sources: bench.py compares this digest with the tree it runs from and says when the
        # constants were measured on another build (`traffic_build_matches`)
        dig = data.get("_csrc_sha1", {})
        dig[tag] = csrc_digest()
        data["_csrc_sha1"] = dig
    json.dump(data, open(tfile, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
