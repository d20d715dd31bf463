"""The Rust shim (rust/phnsw-sys, rust/parallel-hnsw-gpu) cannot be compiled in this image
(no cargo / rustc), so its FFI surface is checked textually: every prototype of include/phnsw.h must be
declared in rust/phnsw-sys/src/lib.rs with the same name, argument count, pointer depth, const-ness and
integer / float widths; the #[repr(C)] structs must list the header's fields in order; and every
sys::phnsw_* call of the wrapper crate must exist with that many arguments."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "phnsw.h")
SYS = os.path.join(ROOT, "rust", "phnsw-sys", "src", "lib.rs")
WRAP = os.path.join(ROOT, "rust", "parallel-hnsw-gpu", "src", "lib.rs")

C_SCALAR = {"int": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "uint16_t": "u16", "float": "f32", "char": "char",
            "void": "void", "size_t": "usize", "double": "f64"}
R_SCALAR = {"c_int": "i32", "u32": "u32", "u64": "u64", "u8": "u8", "u16": "u16", "c_float": "f32", "f32": "f32", "c_char": "char",
            "c_void": "void", "usize": "usize", "f64": "f64", "c_double": "f64"}


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_type(decl, with_name=True):
    """'const uint64_t *const *nodes' -> ('u64', depth 2, outer pointee const)"""
    d = decl.strip()
    if with_name:
        d = re.sub(r"\b[A-Za-z_]\w*\s*$", "", d).strip() if not d.endswith("*") else d
    depth = d.count("*")
    toks = [t for t in re.split(r"[\s\*]+", d) if t]
    const_first = d.lstrip().startswith("const") or (toks and toks[0] == "const")
    base = [t for t in toks if t != "const"]
    assert len(base) == 1, decl
    b = base[0]
    if b == "phnsw_progress_cb":
        return ("cb", 0, False)
    b = C_SCALAR.get(b, b)  # struct names stay as they are
    return (b, depth, bool(const_first) and depth > 0)


def rust_type(t):
    t = t.strip()
    depth, const_inner = 0, False
    first = True
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        depth += 1
        t = t[m.end():]
        const_inner = m.group(1) == "const"  # the innermost pointer's const == C's leading const
        first = False
    if t == "phnsw_progress_cb":
        return ("cb", 0, False)
    return (R_SCALAR.get(t, t), depth, const_inner and depth > 0)


def header_prototypes():
    src = strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_]\w*(?:\s*\*)?)\s*(phnsw_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret:
            continue
        params = [] if args in ("void", "") else [c_type(a) for a in split_args(args)]
        protos[name] = (c_type(ret + " x")[:2] if ret.strip() != "void" else ("void", 0), params)
    return protos


def rust_declarations():
    src = re.sub(r"//.*", "", open(SYS).read())
    block = src[src.index('extern "C" {'):]
    decls = {}
    for m in re.finditer(r"pub fn (phnsw_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()), m.group(3)
        params = [rust_type(a.split(":", 1)[1]) for a in split_args(args)] if args else []
        decls[name] = (rust_type(ret)[:2] if ret else ("void", 0), params)
    return decls


def test_every_header_prototype_is_declared_identically():
    h, r = header_prototypes(), rust_declarations()
    assert len(h) > 60, "header parser lost prototypes"
    assert sorted(h) == sorted(r), {"missing in rust": sorted(set(h) - set(r)), "not in header": sorted(set(r) - set(h))}
    for name in sorted(h):
        assert h[name][0] == r[name][0], (name, "return", h[name][0], r[name][0])
        assert len(h[name][1]) == len(r[name][1]), (name, "arity", len(h[name][1]), len(r[name][1]))
        for i, (a, b) in enumerate(zip(h[name][1], r[name][1])):
            assert a == b, (name, "argument %d" % i, a, b)


def header_struct(name):
    src = strip_c_comments(open(HEADER).read())
    m = re.search(r"typedef struct \{([^}]*)\}\s*%s\s*;" % name, src)
    return [(f.split()[-1], C_SCALAR.get(" ".join(f.split()[:-1]), " ".join(f.split()[:-1])))
            for f in (x.strip() for x in m.group(1).split(";")) if f]


def rust_struct(name):
    src = re.sub(r"//.*", "", open(SYS).read())
    m = re.search(r"pub struct %s \{([^}]*)\}" % name, src)
    return [(f.split(":")[0].replace("pub", "").strip(), R_SCALAR.get(f.split(":")[1].strip(), f.split(":")[1].strip()))
            for f in (x.strip() for x in m.group(1).split(",")) if f]


def test_repr_c_structs_match_the_header_field_for_field():
    for s in ("phnsw_search_params", "phnsw_optimization_params", "phnsw_build_params"):
        assert header_struct(s) == rust_struct(s), s
        assert re.search(r"#\[repr\(C\)\]\s*#\[derive[^\]]*\]\s*pub struct %s " % s, open(SYS).read()), s + " is not #[repr(C)]"


def test_wrapper_calls_exist_with_the_declared_arity():
    decls = rust_declarations()
    src = re.sub(r"//.*", "", open(WRAP).read())
    calls = 0
    for m in re.finditer(r"sys::(phnsw_\w+)\s*\(", src):
        name = m.group(1)
        assert name in decls, name
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = split_args(src[m.end():i - 1])
        assert len(args) == len(decls[name][1]), (name, len(args), len(decls[name][1]))
        calls += 1
    assert calls >= 15


def test_fixed_bench_calls_generate_with_four_arguments():
    """benches/bench.rs:54-63 of the reference still passes (c, vs, 24, 48, 2); lib.rs:825-830 takes
    (c, vs, bp, progress)"""
    src = open(os.path.join(ROOT, "rust", "parallel-hnsw-gpu", "benches", "bench.rs")).read()
    for m in re.finditer(r"(?:Hnsw|GpuHnsw)::generate\((.*?)\);", src, flags=re.S):
        assert len(split_args(m.group(1))) == 4, m.group(0)
    assert len(re.findall(r"::generate\(", src)) >= 2
