"""The Rust `-sys` declarations (integration/rust/src/raytracer/hip/sys.rs) cannot be compiled here (no rustc/cargo), so
they are kept honest by text: every export of include/ptx_core.h must be declared with the same arity and the same
argument / return types, every `#[repr(C)]` struct must list the header's fields in the header's order with matching
types, and the sizes that follow from those fields must equal the C compiler's.  The seam these bindings serve is
`RenderLoop` (/root/reference/src/raytracer/renderloop/mod.rs:9-22)."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptx_core.h")
SYS_RS = os.path.join(ROOT, "integration", "rust", "src", "raytracer", "hip", "sys.rs")

C_TO_RUST = {
    "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "double": "f64", "uint8_t": "u8",
    "int32_t*": "*mut i32", "uint32_t*": "*mut u32", "uint64_t*": "*mut u64", "float*": "*mut f32", "uint8_t*": "*mut u8",
    "const uint32_t*": "*const u32", "const float*": "*const f32", "const char*": "*const c_char",
    "void*": "*mut core::ffi::c_void", "const void*": "*const core::ffi::c_void",
}
SIZES = {"i32": (4, 4), "u32": (4, 4), "u64": (8, 8), "f32": (4, 4), "f64": (8, 8), "u8": (1, 1)}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def c_type_to_rust(t):
    t = re.sub(r"\s+", " ", t.strip()).replace(" *", "*")
    if t in C_TO_RUST:
        return C_TO_RUST[t]
    m = re.fullmatch(r"(const )?(Ptx\w+)(\*{1,2})", t)
    assert m, f"unmapped C type {t!r}"
    inner = ("*const " if m.group(1) else "*mut ") + m.group(2)
    return inner if m.group(3) == "*" else "*mut " + inner


def header_functions():
    text = strip_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"^(const char\*|int32_t|uint64_t)\s+(ptx_\w+)\(([^;]*?)\);", text, flags=re.M | re.S):
        ret, name, args = m.groups()
        params = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",") if x.strip()]:
            mm = re.fullmatch(r"(.*?)(\w+)(\[\d*\])?", a)
            params.append(c_type_to_rust(mm.group(1)))
        out[name] = (C_TO_RUST.get(ret, ret), params)
    return out


def rust_functions():
    text = strip_comments(re.sub(r"///[^\n]*", "", open(SYS_RS).read()))
    block = re.search(r'extern "C" \{(.*)\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (ptx_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.groups()
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args.replace("\n", " ").split(",") if a.strip()]
        out[name] = ((ret or "()").strip(), params)
    return out


def header_structs():
    text = strip_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef struct(?: __attribute__\(\(aligned\(64\)\)\))? (Ptx\w+) \{(.*?)\} \1;", text, flags=re.S):
        name, body = m.groups()
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            mm = re.fullmatch(r"(.*?)\s*((?:\*?\w+(?:\[\d+\])?\s*,\s*)*\*?\w+(?:\[\d+\])?)", decl.replace("\n", " "), flags=re.S)
            ctype = mm.group(1).strip()
            for item in [x.strip() for x in mm.group(2).split(",")]:
                t = ctype
                if item.startswith("*"):
                    t, item = ctype + "*", item[1:]
                am = re.fullmatch(r"(\w+)(?:\[(\d+)\])?", item)
                fields.append((am.group(1), t.replace(" *", "*"), int(am.group(2)) if am.group(2) else None))
        out[name] = fields
    return out


def rust_structs():
    text = strip_comments(re.sub(r"///[^\n]*", "", open(SYS_RS).read()))
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (Ptx\w+) \{(.*?)\}", text, flags=re.S):
        name, body = m.groups()
        out[name] = [(f.group(1), re.sub(r"\s+", " ", f.group(2).strip())) for f in re.finditer(r"pub (\w+):\s*([^,]+),", body)]
    return out


def rust_field_type(ctype, n):
    if ctype in C_TO_RUST and not ctype.endswith("*"):
        base = C_TO_RUST[ctype]
    elif ctype.endswith("*"):
        inner = ctype[:-1].strip()
        const = inner.startswith("const ")
        inner = inner[6:] if const else inner
        inner = {"PtxBvh8Node": "Bvh8Node"}.get(inner, C_TO_RUST.get(inner, inner))
        base = ("*const " if const else "*mut ") + inner
    else:
        base = ctype          # nested Ptx struct by value
    return f"[{base}; {n}]" if n else base


def test_every_export_is_declared_with_the_same_signature():
    c, r = header_functions(), rust_functions()
    assert len(c) >= 35
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, (ret, params) in c.items():
        r_ret, r_params = r[name]
        assert r_ret == ret, (name, r_ret, ret)
        assert r_params == params, (name, r_params, params)


def test_repr_c_structs_match_the_header_field_for_field():
    c, r = header_structs(), rust_structs()
    c.pop("PtxBvh8Node")            # the reference's own Bvh8Node (accel/bvh.rs:86-98) passes as-is
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, fields in c.items():
        want = [(n, rust_field_type(t, k)) for n, t, k in fields]
        assert r[name] == want, (name, r[name], want)


def test_struct_sizes_follow_from_the_fields():
    """repr(C) layout of the scalar structs computed from the Rust field list == sizeof in C."""
    r = rust_structs()
    names = ["PtxCounters", "PtxTimings", "PtxDebugStats", "PtxFrameStats", "PtxAdaptiveConfig", "PtxRay", "PtxHitRecord", "PtxCamera", "PtxSun", "PtxMaterial", "PtxInstance"]

    def layout(fields):
        off, align = 0, 1
        for _, t in fields:
            m = re.fullmatch(r"\[(\w+); (\d+)\]", t)
            base, n = (m.group(1), int(m.group(2))) if m else (t, 1)
            sz, al = SIZES[base]
            off = (off + al - 1) // al * al + sz * n
            align = max(align, al)
        return (off + align - 1) // align * align

    src = '#include "ptx_core.h"\n#include <stdio.h>\nint main(){' + "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = dict(line.split() for line in subprocess.check_output([os.path.join(d, "s")], text=True).strip().splitlines())
    for n in names:
        assert layout(r[n]) == int(sizes[n]), (n, layout(r[n]), sizes[n])
