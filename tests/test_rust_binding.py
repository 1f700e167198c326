"""bindings/rust/raytracer-amd-sys/src/lib.rs against include/rt_amd.h, field by field.

There is no Rust toolchain in this image, so the hand-written `-sys` crate cannot be compiled here.  What CAN be
checked without one: that every function, struct field (name, order, type), constant and enum value the header
declares is declared the same way in lib.rs, that every struct is #[repr(C)], and that the reference-side patch
(bindings/rust/reference-patch/src/amd.rs) only names items lib.rs has and fills every field of the structs it
builds.  A header change that is not carried into the Rust side fails here.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_amd.h")
LIB_RS = os.path.join(ROOT, "bindings", "rust", "raytracer-amd-sys", "src", "lib.rs")
AMD_RS = os.path.join(ROOT, "bindings", "rust", "reference-patch", "src", "amd.rs")

C_TO_RUST = {"uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "double": "f64", "int": "c_int",
             "size_t": "usize", "void": "c_void", "char": "c_char", "unsigned char": "u8"}


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _strip_rs(text):
    return re.sub(r"//[^\n]*", " ", text)


def rust_type_of(ctype: str) -> str:
    """'const float*' -> '*const f32', 'rt_scene**' -> '*mut *mut rt_scene', 'uint32_t' -> 'u32'."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").strip()
    const = t.startswith("const ")
    base = t[6:].strip() if const else t
    base = C_TO_RUST.get(base, base)
    out = base
    for k in range(stars):
        out = ("*const " if (const and k == 0) else "*mut ") + out
    return out


def parse_header():
    src = _strip_c(open(HEADER).read())
    consts = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(RT_\w+)\s+(0x[0-9a-fA-F]+|\d+)u?\b", src)}
    for m in re.finditer(r"enum(?:\s+\w+)?\s*\{([^}]*)\}", src):
        for item in m.group(1).split(","):
            if "=" in item:
                k, v = item.split("=")
                consts[k.strip()] = int(v.strip(), 0)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{([^}]*)\}\s*\1\s*;", src):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "const float* prim_inv" | "uint32_t hsize, vsize" | "float inv[12]" | "double a, b, c"
            mm = re.match(r"(.*?[\s\*])(\w+(?:\[\d+\])?(?:\s*,\s*\w+(?:\[\d+\])?)*)$", decl)
            ctype, names = mm.group(1).strip(), mm.group(2)
            for name in names.split(","):
                name = name.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", name)
                if arr:
                    fields.append((arr.group(1), f"[{rust_type_of(ctype)}; {arr.group(2)}]"))
                else:
                    fields.append((name, rust_type_of(ctype)))
        structs[m.group(1)] = fields
    funcs = {}
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{[^}]*\}\s*\w+\s*;", " ", src)
    for m in re.finditer(r"\b((?:const\s+)?\w+\s*\**)\s*(rt_\w+)\s*\(([^)]*)\)\s*;", body):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        alist = []
        if args.strip() != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?[\s\*])(\w+)$", a)
                alist.append((mm.group(2), rust_type_of(mm.group(1))))
        funcs[name] = (None if ret.strip() == "void" else rust_type_of(ret), alist)
    return consts, structs, funcs


def parse_lib_rs():
    src = _strip_rs(open(LIB_RS).read())
    consts = {m.group(1): int(m.group(3).replace("_", ""), 0) for m in
              re.finditer(r"pub const (\w+): (\w+) = (-?(?:0x[0-9a-fA-F_]+|\d+));", src)}
    structs, repr_c = {}, {}
    for m in re.finditer(r"((?:#\[[^\]]*\]\s*)*)pub struct (\w+) \{([^}]*)\}", src):
        repr_c[m.group(2)] = "#[repr(C)]" in m.group(1)
        structs[m.group(2)] = [(f.group(1), " ".join(f.group(2).split())) for f in re.finditer(r"pub (\w+): ([^,]+),", m.group(3))]
    funcs = {}
    ext = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
        args = [(a.split(":")[0].strip(), " ".join(a.split(":", 1)[1].split())) for a in m.group(2).split(",") if a.strip()]
        funcs[m.group(1)] = (m.group(3).strip() if m.group(3) else None, args)
    opaque = set(re.findall(r"pub enum (\w+) \{\}", src))
    return consts, structs, repr_c, funcs, opaque


def test_type_mapping_of_the_checker_itself():
    assert rust_type_of("const float*") == "*const f32"
    assert rust_type_of("rt_scene**") == "*mut *mut rt_scene"
    assert rust_type_of("void*") == "*mut c_void"
    assert rust_type_of("const char*") == "*const c_char"
    assert rust_type_of("uint32_t") == "u32"


def test_every_function_of_the_header_is_declared_identically_in_lib_rs():
    _, _, c_funcs = parse_header()
    _, _, _, r_funcs, opaque = parse_lib_rs()
    assert "rt_scene" in opaque
    assert set(c_funcs) == set(r_funcs), (sorted(set(c_funcs) - set(r_funcs)), sorted(set(r_funcs) - set(c_funcs)))
    for name, (ret, args) in c_funcs.items():
        r_ret, r_args = r_funcs[name]
        assert ret == r_ret, (name, ret, r_ret)
        assert args == r_args, (name, args, r_args)  # names, order and types


def test_every_struct_matches_field_by_field_and_is_repr_c():
    _, c_structs, _ = parse_header()
    _, r_structs, repr_c, _, _ = parse_lib_rs()
    assert set(c_structs) == {"rt_scene_desc", "rt_camera_desc", "rt_options", "rt_stats", "rt_ipc_handle"}
    for name, fields in c_structs.items():
        assert name in r_structs, name
        assert repr_c[name], f"{name} is not #[repr(C)]"
        assert fields == r_structs[name], (name, [a for a in zip(fields, r_structs[name]) if a[0] != a[1]],
                                           len(fields), len(r_structs[name]))


def test_every_constant_and_enum_value_matches():
    c_consts, _, _ = parse_header()
    r_consts, _, _, _, _ = parse_lib_rs()
    missing = sorted(set(c_consts) - set(r_consts))
    assert not missing, missing
    for k, v in c_consts.items():
        assert r_consts[k] == v, (k, v, r_consts[k])
    assert sorted(set(r_consts) - set(c_consts)) == []


def test_header_constants_agree_with_the_python_binding_too():
    from raytracer_amd import capi
    c_consts, _, _ = parse_header()
    for k in ("RT_ABI_VERSION", "RT_FLAG_GLOBAL_ROWS", "RT_FLAG_F32_ONLY", "RT_FLAG_DEPTH_ZERO", "RT_NOALLOC_MAX_TILES"):
        assert getattr(capi, k) == c_consts[k], k


def test_the_reference_side_patch_only_uses_what_lib_rs_declares_and_fills_every_field():
    src = _strip_rs(open(AMD_RS).read())
    r_consts, r_structs, _, r_funcs, _ = parse_lib_rs()
    known = set(r_consts) | set(r_structs) | set(r_funcs)
    used = set(re.findall(r"\bsys::(\w+)", src))
    assert used <= known, sorted(used - known)
    assert {"rt_scene_create", "rt_render", "rt_scene_destroy", "rt_last_error"} <= used
    for name in ("rt_scene_desc", "rt_camera_desc"):
        m = re.search(r"^\s+(?:let \w+ = )?sys::%s \{(.*?)\n\s*\}" % name, src, flags=re.S | re.M)  # the struct literal
        given = re.findall(r"^\s*(\w+):", m.group(1), flags=re.M)
        assert given == [f for f, _ in r_structs[name]], (name, given)
    m = re.search(r"sys::rt_options \{(.*?)\}", src, flags=re.S)
    assert "..Default::default()" in m.group(1)
    for f in re.findall(r"^\s*(\w+):", m.group(1), flags=re.M):
        assert f in [x for x, _ in r_structs["rt_options"]], f
