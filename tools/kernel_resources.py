#!/usr/bin/env python3
"""Registers and occupancy of every gfx950 kernel of libwlhip.so, from the compiler's own report
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed): one line per kernel -- VGPRs, SGPRs, scratch, LDS,
waves per SIMD -- with the demangled template arguments shortened to what tells the variants apart.
usage: kernel_resources.py [filter substring ...]   (writes the table to stdout; profiles/r04a_kernel_resources.txt is its output)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "waterlily_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
       "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c", "wl_api.hip", "-o", "/tmp/wl_api_res.o"]
log = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in log.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()


def short(d):
    d = re.sub(r"wl::", "", d)
    k = re.match(r"(?:void )?(k_\w+)<(.*)>\(", d)
    if not k:
        return re.sub(r"\(.*", "", d)
    kern, args = k.group(1), k.group(2)
    T = "f32" if args.startswith("float") else ("f64" if args.startswith("double") else "")
    op = re.search(r"(op_\w+|mg_\w+|flow_\w+|red_\w+)<", args)
    lam = re.findall(r"#(\d+)\}", args)
    extra = []
    if kern == "k_stencil7":
        m = re.match(r"(?:float|double), (\d+), (\d+), (Src\w+)", args)
        if m:
            extra = [f"NRED={m.group(1)}", f"R={m.group(2)}", m.group(3)]
        if "ResidualDivEpi" in args:
            extra.append("ResidualDivEpi")
    elif kern == "k_rowvec":
        m = re.match(r"(?:float|double), (\d+), (true|false)", args)
        if m:
            extra = [f"NRED={m.group(1)}", "rowconst" if m.group(2) == "true" else ""]
    elif kern.startswith("k_convdiff3s"):
        m = re.match(r"(?:float|double), (true|false), (true|false), (\d+)", args)
        if m:
            extra = [f"fuse={m.group(1)}", f"copy={m.group(2)}", f"BY={m.group(3)}"]
    return " ".join(x for x in [kern, T, op.group(1) if op else "", ("#" + lam[0]) if lam else ""] + extra if x)


want = sys.argv[1:]
print(f"{'kernel':78s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'LDS B':>7s} {'waves/SIMD':>10s}")
out = []
for r, d in zip(rows, names):
    s = short(d)
    if want and not any(w in s for w in want):
        continue
    out.append((s, r))
for s, r in sorted(out, key=lambda t: t[0]):
    print(f"{s[:78]:78s} {r.get('VGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('ScratchSize', '?'):>7s} {r.get('LDS Size', '?'):>7s} {r.get('Occupancy', '?'):>10s}")
