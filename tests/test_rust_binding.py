"""bindings/rust/wr_sys.rs against include/wr_abi.h.

The image has no Rust toolchain, so the binding cannot be compiled here.  What can be checked is
that it says the same as the header: every function in the same order with the same argument
count and the same types, every struct with the same fields in the same order, every constant
with the same value; and that the header is valid C99 on its own, which is what bindgen or a
cgo-style tool would need of it.
"""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "wr_abi.h"
RUST = ROOT / "bindings" / "rust" / "wr_sys.rs"

SCALARS = {
    "float": "f32", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "uint8_t": "u8",
    "int": "c_int", "size_t": "usize", "char": "c_char", "void": "c_void",
}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def c_type_to_rust(ctype, dims):
    """Rust spelling of a C declarator's type: base type with `const` and stars, then the array
    dimensions of the declarator (outermost first)."""
    tokens = ctype.replace("*", " * ").split()
    const = "const" in tokens
    tokens = [t for t in tokens if t not in ("const", "struct")]
    base, stars = tokens[0], tokens.count("*")
    rust = SCALARS.get(base, base)
    for level in range(stars):
        # `const` binds to the pointee next to the base type; outer pointer levels are `mut`
        rust = ("*const " if (const and level == 0) else "*mut ") + rust
    for d in reversed(dims):
        rust = f"[{rust}; {d}]"
    return rust


def c_param_to_rust(ctype, dims):
    """A C array parameter is a pointer to its element."""
    if not dims:
        return c_type_to_rust(ctype, [])
    inner = c_type_to_rust(ctype, dims[1:])
    const = "const" in ctype.split()
    return ("*const " if const else "*mut ") + inner


def split_declarator(decl):
    """'const float position[3]' -> ('const float', 'position', ['3'])"""
    dims = re.findall(r"\[(\d+)\]", decl)
    decl = re.sub(r"\[\d+\]", "", decl).strip()
    m = re.match(r"(.*?)(\w+)$", decl, flags=re.S)
    return m.group(1).strip(), m.group(2), dims


def parse_header():
    text = strip_c_comments(HEADER.read_text())
    structs, funcs, consts = {}, [], {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for member in m.group(2).split(";"):
            member = " ".join(member.split())
            if not member:
                continue
            fp = re.match(r"(\w[\w\s\*]*?)\(\*(\w+)\)\((.*)\)$", member)
            if fp:
                args = [split_declarator(a) for a in fp.group(3).split(",")]
                sig = ", ".join(c_param_to_rust(t, d) for t, _, d in args)
                fields.append((fp.group(2), f"fn({sig}) -> {c_type_to_rust(fp.group(1), [])}"))
            else:
                t, name, dims = split_declarator(member)
                fields.append((name, c_type_to_rust(t, dims)))
        structs[m.group(1)] = fields
    opaque = [n for n in re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text)]
    body = re.sub(r"typedef\s+(struct|enum)\s+\w*\s*\{.*?\}\s*\w*\s*;", "", text, flags=re.S)
    body = re.sub(r"enum\s*\{.*?\}\s*;", "", body, flags=re.S)
    for m in re.finditer(r"([\w][\w\s\*]*?)\b(wr_\w+)\s*\(([^;{]*)\)\s*;", body, flags=re.S):
        ret = " ".join(m.group(1).split())
        args = [" ".join(a.split()) for a in m.group(3).split(",")]
        params = []
        for a in args:
            t, name, dims = split_declarator(a)
            params.append((name, c_param_to_rust(t, dims)))
        funcs.append((m.group(2), params, None if ret == "void" else c_type_to_rust(ret, [])))
    for m in re.finditer(r"\b(WR_[A-Z0-9_]+)\s*=\s*(-?\d+)u?", text):
        consts[m.group(1)] = int(m.group(2))
    return structs, opaque, funcs, consts


def parse_rust():
    text = strip_rust_comments(RUST.read_text())
    structs, funcs, consts = {}, [], {}
    for m in re.finditer(r"#\[repr\(C\)\][^{;]*?pub struct (\w+)\s*\{(.*?)\n\}", text, flags=re.S):
        fields = []
        for fm in re.finditer(r"(?:pub\s+)?(\w+)\s*:\s*(.+?),\s*\n", m.group(2) + "\n"):
            ftype = " ".join(fm.group(2).split())
            opt = re.match(r"Option<unsafe extern \"C\" fn\((.*)\) -> (\w+)>$", ftype)
            if opt:
                sig = ", ".join(a.split(":", 1)[1].strip() for a in opt.group(1).split(","))
                ftype = f"fn({sig}) -> {opt.group(2)}"
            fields.append((fm.group(1), ftype))
        structs[m.group(1)] = fields
    block = re.search(r"extern \"C\" \{(.*?)\n\}", text, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        params = []
        for a in filter(None, (a.strip() for a in m.group(2).split(","))):
            name, t = a.split(":", 1)
            params.append((name.strip(), " ".join(t.split())))
        funcs.append((m.group(1), params, m.group(3).strip() if m.group(3) else None))
    for m in re.finditer(r"pub const (WR_\w+): \w+ = (-?\d+);", text):
        consts[m.group(1)] = int(m.group(2))
    return structs, funcs, consts


def test_header_is_valid_c99_on_its_own():
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", str(HEADER)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_function_of_the_header_is_bound_with_the_same_signature():
    _, _, c_funcs, _ = parse_header()
    _, r_funcs, _ = parse_rust()
    assert len(c_funcs) >= 42
    assert [f[0] for f in r_funcs] == [f[0] for f in c_funcs]  # all of them, in the header's order
    for (name, c_params, c_ret), (_, r_params, r_ret) in zip(c_funcs, r_funcs):
        assert [t for _, t in r_params] == [t for _, t in c_params], name
        assert [n for n, _ in r_params] == [n for n, _ in c_params], name
        assert r_ret == c_ret, name


def test_every_struct_of_the_header_has_the_same_fields_in_the_same_order():
    c_structs, opaque, _, _ = parse_header()
    r_structs, _, _ = parse_rust()
    assert set(c_structs) == {"WrModelVertex", "WrInstanceRaw", "WrCameraUniform", "WrLightUniform", "WrGlobalUniforms",
                              "WrMaterialUniform", "WrConfig", "WrFrameStats", "WrTransport"}
    for name, fields in c_structs.items():
        assert r_structs[name] == fields, name
    assert set(opaque) == {"WrContext", "WrGlb"}
    for name in opaque:
        assert r_structs[name] == [("_private", "[u8; 0]")]


def test_every_constant_of_the_header_has_the_same_value():
    _, _, _, c_consts = parse_header()
    _, _, r_consts = parse_rust()
    assert len(c_consts) >= 16 and r_consts == c_consts


def test_the_ctypes_host_binds_the_same_symbols():
    from wgpu_renderer_amd import abi
    _, _, c_funcs, _ = parse_header()
    assert sorted(abi.SYMBOLS) == sorted(f[0] for f in c_funcs)


def test_the_parser_tells_a_wrong_binding_from_a_right_one(tmp_path, monkeypatch):
    """The comparison is only worth something if it fails on a drifted binding."""
    good = RUST.read_text()
    for old, new in (("pub fn wr_sync(ctx: *mut WrContext) -> c_int;", ""),
                     ("passes: u32, device_color: *mut c_void", "passes: u64, device_color: *mut c_void"),
                     ("    pub ms_total: f32,\n    pub profiled_frames: u32,", "    pub profiled_frames: u32,\n    pub ms_total: f32,")):
        assert old in good
        bad = tmp_path / "wr_sys.rs"
        bad.write_text(good.replace(old, new))
        monkeypatch.setattr("test_rust_binding.RUST", bad)
        with pytest.raises(AssertionError):
            test_every_function_of_the_header_is_bound_with_the_same_signature()
            test_every_struct_of_the_header_has_the_same_fields_in_the_same_order()
