#!/usr/bin/env python3
"""VGPRs / scratch / occupancy / LDS of the library's kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: kres.py <remarks.txt> [substring ...]     (make the remarks with:  hipcc <CXXFLAGS> -c wl_api.hip -o /dev/null
-Rpass-analysis=kernel-resource-usage 2> remarks.txt)"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:] or ["k_stencil7", "k_convdiff3", "k_rowvec", "k_correct3"]
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks, dem):
    if not any(w in d for w in want):
        continue
    g = lambda pat: int(re.search(pat, b).group(1))
    m = re.search(r"k_stencil7<(\w+), (\d), (\d), wl::(\w+)<", d)
    op = re.search(r"(op_\w+)<", d)
    lam = re.search(r"#(\d)\}", d)
    if m:
        tag = f"k_stencil7<{m.group(1)},NRED={m.group(2)},R={m.group(3)},{m.group(4)}> {op.group(1) if op else ''}#{lam.group(1) if lam else ''}" + (" ResidualDivEpi" if "ResidualDivEpi" in d else "")
    else:
        tag = re.sub(r"^void wl::", "", d)[:60] + (f" {op.group(1)}#{lam.group(1) if lam else ''}" if op and "k_rowvec" in d else "")
    v, a, sc, oc, lds = g(r"VGPRs: (\d+)"), g(r"AGPRs: (\d+)"), g(r"ScratchSize \[bytes/lane\]: (\d+)"), g(r"Occupancy \[waves/SIMD\]: (\d+)"), g(r"LDS Size \[bytes/block\]: (\d+)")
    print(f"{tag:86s} VGPR {v:4d} AGPR {a:3d} scratch {sc:4d} occ {oc} lds {lds}")
