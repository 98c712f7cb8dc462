"""Static consistency of the reference-side binding with the C ABI.

`julia/WaterLilyHIPNativeExt.jl` cannot be executed here (no Julia runtime), so what CAN be checked is checked: every
`ccall` in it names a function `include/wlhip.h` declares, passes as many arguments as the prototype takes, and passes the
same KIND of argument in every position (pointer / int / double / size_t / int64) -- a swapped or missing argument in a
binding nobody can run would otherwise go unnoticed until a maintainer tried it.  The structs the shim mirrors
(`WlGrid`, `WlLevel`, `WlFlow`, `WlBody`) are compared field by field with the header's, too."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "julia", "WaterLilyHIPNativeExt.jl")).read()
HDR = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "wlhip.h")).read(), flags=re.S)


def split_top(s):
    parts, d, cur = [], 0, ""
    for c in s:
        d += c in "({["
        d -= c in ")}]"
        if c == "," and d == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += c
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts if p.strip()]


def header_protos():
    out = {}
    for m in re.finditer(r"\b(int|const char \*)\s*(wl_\w+)\s*\(([^;]*?)\)\s*;", HDR, flags=re.S):
        args = m.group(3).strip()
        out[m.group(2)] = [] if args in ("void", "") else split_top(args)
    return out


def c_kind(decl):
    d = decl.strip()
    if "*" in d or "[" in d or re.search(r"\bwl_host_\w+_fn\b", d):
        return "ptr"
    if re.match(r"(const\s+)?double\b", d):
        return "double"
    if re.match(r"(const\s+)?size_t\b", d):
        return "size_t"
    if re.match(r"(const\s+)?int64_t\b", d):
        return "int64"
    if re.match(r"(const\s+)?(int|wl_dtype|int32_t)\b", d):
        return "int"
    raise AssertionError(f"unclassified C parameter: {d!r}")


def jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr"):
        return "ptr"
    return {"Cint": "int", "Int32": "int", "Cdouble": "double", "Csize_t": "size_t", "Int64": "int64"}[t]


def shim_ccalls():
    calls = []
    for m in re.finditer(r"ccall\(\(\s*(:\w+|\$\(QuoteNode\(sym\)\))\s*,\s*lib\)\s*,\s*(\w+)\s*,\s*\(", JL):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += JL[j] == "("
            depth -= JL[j] == ")"
            j += 1
        names = [m.group(1)[1:]] if m.group(1).startswith(":") else ["wl_sum", "wl_max"]   # (the one generated pair)
        for n in names:
            calls.append((n, m.group(2), split_top(JL[i:j - 1])))
    return calls


def test_every_ccall_matches_a_prototype_in_name_arity_and_argument_kinds():
    protos = header_protos()
    calls = shim_ccalls()
    assert len(calls) >= 60 and len({n for n, _, _ in calls}) >= 50
    for name, ret, types in calls:
        assert name in protos, f"the shim calls {name}, which include/wlhip.h does not declare"
        want = [c_kind(a) for a in protos[name]]
        got = [jl_kind(t) for t in types]
        assert got == want, f"{name}: header takes {want}, the shim passes {got}"
        assert ret == ("Cstring" if name == "wl_last_error" else "Cint"), (name, ret)


def test_the_hot_path_entry_points_are_bound():
    bound = {n for n, _, _ in shim_ccalls()}
    for need in ("wl_malloc", "wl_free", "wl_h2d_2d", "wl_d2h_2d", "wl_memset0", "wl_flow_create", "wl_mg_create", "wl_mg_update",
                 "wl_mg_update_changed", "wl_flow_update", "wl_mom_step", "wl_project", "wl_mg_solve", "wl_conv_diff", "wl_bdim",
                 "wl_bc_vec", "wl_bc_per", "wl_exit_bc", "wl_cfl", "wl_set_diag", "wl_mg_mult", "wl_mg_residual", "wl_mg_increment",
                 "wl_mg_jacobi", "wl_mg_pcg", "wl_mg_L2", "wl_mg_Linf", "wl_mg_vcycle", "wl_restrict", "wl_prolongate", "wl_restrictL",
                 "wl_pforce", "wl_vforce", "wl_pmoment", "wl_measure_rows", "wl_measure_fill", "wl_body_nds", "wl_dot", "wl_sum",
                 "wl_max", "wl_L2_inside", "wl_comm_unique_id", "wl_comm_init_rccl", "wl_comm_mailbox", "wl_comm_finalize"):
        assert need in bound, f"{need} is not bound by the shim"


def c_struct_fields(name):
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", HDR, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(.*?)([\w\s,\*\[\]]+)$", decl)
        base = re.match(r"(const\s+)?(\w+)", decl).group(2)
        for var in decl[len(re.match(r"(const\s+)?\w+", decl).group(0)):].split(","):
            var = var.strip()
            arr = re.search(r"\[(\d+)\]", var)
            kind = "ptr" if var.startswith("*") else base
            out.append((kind, int(arr.group(1)) if arr else 1))
    return out


def jl_struct_fields(name):
    body = re.search(r"struct " + name + r"\b(.*?)\nend", JL, flags=re.S).group(1)
    body = re.sub(r"#.*", "", body)
    out = []
    for f in re.split(r"[;\n]", body):
        f = f.strip()
        if "::" not in f:
            continue
        t = f.split("::")[1].strip()
        tup = re.match(r"NTuple\{(\d+),\s*(\w+)\}", t)
        out.append((tup.group(2), int(tup.group(1))) if tup else (t, 1))
    return out


def test_mirrored_structs_have_the_headers_fields():
    same = {"int32_t": "Int32", "int64_t": "Int64", "double": "Cdouble", "ptr": "Ptr{Cvoid}", "wl_grid": "WlGrid"}
    for cname, jname in (("wl_grid", "WlGrid"), ("wl_level_desc", "WlLevel"), ("wl_flow_desc", "WlFlow"), ("wl_body_desc", "WlBody")):
        c, j = c_struct_fields(cname), jl_struct_fields(jname)
        assert [(same[k], n) for k, n in c] == j, (cname, c, j)
