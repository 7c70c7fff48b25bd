"""Static ABI check of the Julia host package against include/pfmi.h.

No Julia exists in this image, so `pathfinder.jl_amd/julia/PathfinderMI355X.jl` has never been parsed or run by one; what CAN be
verified here is that the package still speaks the header's ABI: every `ccall((:pfmi_x, libpfmi), Ret, (Args...), ...)` is compared --
arity, integer widths, float-ness, pointer-ness, return type -- with the prototype of `pfmi_x` in include/pfmi.h, the `CTarget` struct
with `pfmi_target` field by field, and the `@cfunction` trampolines with the callback typedefs.  (Call ORDER is replayed on the GPU
by examples/julia_sequence.c; this file pins the signatures.)
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pfmi.h")
JULIA = os.path.join(ROOT, "pathfinder.jl_amd", "julia", "PathfinderMI355X.jl")


def _c_kind(ctype):
    """C parameter / return type -> ABI class"""
    t = re.sub(r"\bconst\b", "", ctype).strip()
    if "*" in t or t in ("pfmi_logp_fn", "pfmi_logp_dev_fn"):
        return "ptr"
    t = t.split()[0] if t else t
    return {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "double": "f64",
            "void": "void"}[t]


def _split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({":
            depth += 1
        elif ch in ")}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _header():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_0-9]+\s*\**)\s*(pfmi_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.M | re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ps = [] if params.strip() in ("", "void") else _split_params(params)
        kinds = []
        for p in ps:
            p = re.sub(r"\s+", " ", p).strip()
            mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", p)          # strip the parameter name
            kinds.append(_c_kind(mm.group(1) if mm and mm.group(1).strip() else p))
        protos[name] = (_c_kind(ret), kinds)
    st = re.search(r"typedef struct \{(.*?)\}\s*pfmi_target\s*;", txt, flags=re.S).group(1)
    fields = []
    for line in st.split(";"):
        line = re.sub(r"\s+", " ", line).strip()
        if not line:
            continue
        mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", line)
        fields.append((mm.group(2), _c_kind(mm.group(1))))
    cbs = {}
    for m in re.finditer(r"typedef void \(\*(pfmi_logp[a-z_]*fn)\)\((.*?)\);", txt, flags=re.S):
        kinds = []
        for p in _split_params(m.group(2)):
            mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", re.sub(r"\s+", " ", p).strip())
            kinds.append(_c_kind(mm.group(1)))
        cbs[m.group(1)] = kinds
    return protos, fields, cbs


def _jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr"):
        return "ptr"
    return {"Int32": "i32", "Cint": "i32", "Int64": "i64", "UInt64": "u64", "UInt32": "u32", "UInt8": "u8", "Float64": "f64",
            "Cdouble": "f64", "Cvoid": "void", "Nothing": "void"}[t]


def _julia():
    src = open(JULIA).read()
    src_nc = re.sub(r"#=.*?=#", "", src, flags=re.S)
    src_nc = "\n".join(line.split("#")[0] if "\"" not in line else line for line in src_nc.split("\n"))   # (no `#` inside the ccall lines' strings)
    calls = []
    for m in re.finditer(r"ccall\(\(\s*:(pfmi_[a-z0-9_]+)\s*,\s*libpfmi\s*\)\s*,", src_nc):
        i = m.end()
        # return type up to the next top-level comma, then the parenthesised argument-type tuple
        j = src_nc.index(",", i)
        ret = src_nc[i:j].strip()
        k = src_nc.index("(", j)
        depth, e = 0, k
        while True:
            ch = src_nc[e]
            depth += ch in "({"
            depth -= ch in ")}"
            if depth == 0:
                break
            e += 1
        tup = src_nc[k + 1:e]
        args = [a for a in _split_params(tup) if a]
        line = src_nc.count("\n", 0, m.start()) + 1
        calls.append((m.group(1), ret, args, line))
    st = re.search(r"struct CTarget\n(.*?)\nend", src_nc, flags=re.S).group(1)
    fields = []
    for part in re.split(r"[;\n]", st):
        part = part.strip()
        if part:
            n, t = part.split("::")
            fields.append((n.strip(), _jl_kind(t)))
    cfs = re.findall(r"@cfunction\(\s*([A-Za-z_0-9]+)\s*,\s*([A-Za-z0-9]+)\s*,\s*\((.*?)\)\)", src_nc, flags=re.S)
    return calls, fields, cfs


def test_every_julia_ccall_matches_the_header_prototype():
    protos, _, _ = _header()
    calls, _, _ = _julia()
    assert len(protos) >= 50 and len(calls) >= 35, (len(protos), len(calls))
    bad = []
    for name, ret, args, line in calls:
        assert name in protos, f"PathfinderMI355X.jl:{line}: {name} is not declared in include/pfmi.h"
        cret, cargs = protos[name]
        jret, jargs = _jl_kind(ret), [_jl_kind(a) for a in args]
        if jret != cret or jargs != cargs:
            bad.append(f"PathfinderMI355X.jl:{line}: {name}: Julia {jret} {jargs} != header {cret} {cargs}")
    assert not bad, "\n".join(bad)
    # the package reaches every stage of the path: the four call sites' entry points are among its ccalls
    used = {c[0] for c in calls}
    for need in ("pfmi_fit_batch", "pfmi_elbo_batch_enqueue", "pfmi_elbo_batch_wait", "pfmi_pool_build_best", "pfmi_comm_psis_resample",
                 "pfmi_optimize_batch_enqueue", "pfmi_set_traces", "pfmi_get_fit", "pfmi_draws", "pfmi_woodbury_apply"):
        assert need in used, need


def test_julia_ctarget_layout_is_pfmi_target():
    _, cfields, _ = _header()
    _, jfields, _ = _julia()
    assert [k for _, k in jfields] == [k for _, k in cfields], (jfields, cfields)
    # same field order by name too (the Julia struct spells the header's names)
    assert [n for n, _ in jfields] == [n for n, _ in cfields], (jfields, cfields)
    # all-8-byte-aligned layout: four int32 then pointers / double: no implicit padding on either side
    assert [k for _, k in cfields[:4]] == ["i32"] * 4 and all(k in ("ptr", "f64") for _, k in cfields[4:])


def test_julia_cfunction_trampolines_match_the_callback_typedefs():
    _, _, cbs = _header()
    _, _, cfs = _julia()
    assert cfs, "no @cfunction in the Julia package"
    sigs = {tuple(_jl_kind(a) for a in _split_params(args)) for _, ret, args in cfs if _jl_kind(ret) == "void"}
    assert tuple(cbs["pfmi_logp_fn"]) in sigs, (sigs, cbs["pfmi_logp_fn"])
