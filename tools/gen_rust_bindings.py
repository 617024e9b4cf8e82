#!/usr/bin/env python3
"""Generates bindings/art_sys.rs -- the raw Rust binding of include/art.h a maintainer of the reference would add --
from the header itself: every #define, struct, opaque handle, callback type and function, plus a compile-time layout block whose sizes
come from gcc (sizeof / offsetof of the C structs).  There is no Rust toolchain in this image, so the file is generated and checked for
completeness (tests/test_host.py), not compiled.

    python tools/gen_rust_bindings.py            # writes bindings/art_sys.rs
    python tools/gen_rust_bindings.py --check    # exit 1 if the committed file is not what the header generates"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "art.h")
OUT = os.path.join(ROOT, "bindings", "art_sys.rs")

SCALAR = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "uint16_t": "u16", "float": "f32", "double": "f64", "size_t": "usize",
          "char": "c_char", "void": "c_void", "int": "i32"}
RUST_KEYWORDS = {"type": "light_type", "in": "in_", "ref": "ref_", "box": "box_", "move": "move_", "fn": "fn_"}


def strip_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def rust_type(ctype, opaque):
    """C declarator type (without the name) -> Rust"""
    t = ctype.strip()
    m = re.match(r"^(const\s+)?(\w+)\s*((?:\*\s*(?:const\s*)?)*)$", t)
    if not m:
        raise ValueError(f"cannot map C type {ctype!r}")
    const, base, stars = bool(m.group(1)), m.group(2), m.group(3)
    r = SCALAR.get(base, base)
    levels = re.findall(r"\*\s*(const)?", stars)
    for i, lvl_const in enumerate(levels):
        # pointer level i (innermost first): constness of what it points to
        pointee_const = const if i == 0 else bool(levels[i - 1])
        r = ("*const " if pointee_const else "*mut ") + r
    return r


def field_name(n):
    return RUST_KEYWORDS.get(n, n)


def parse(header):
    src = strip_comments(header)
    consts = re.findall(r"^#define\s+(ART_\w+)\s+\(?(-?\d+)u?\)?\s*$", src, flags=re.M)
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src)
    structs = []
    for m in re.finditer(r"(#pragma pack\(push, 1\)\s*)?typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\2\s*;", src, flags=re.S):
        packed, name, body = bool(m.group(1)), m.group(2), m.group(3)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(.*?)([\w\[\]\s,]+)$", decl)
            # one declaration may name several fields of one type: "uint32_t a, b;" / "float pos[3];"
            tm = re.match(r"^((?:const\s+)?\w+(?:\s*\*)*)\s*(.*)$", decl)
            ctype, names = tm.group(1), tm.group(2)
            for n in names.split(","):
                n = n.strip()
                am = re.match(r"^(\**)\s*(\w+)(?:\[(\d+)\])?$", n)
                ptr, fname, arr = am.group(1), am.group(2), am.group(3)
                rt = rust_type(ctype + ptr, opaque)
                if rt in ("ArtMgpuExchangeFn",):
                    pass
                fields.append((fname, f"[{rt}; {arr}]" if arr else rt))
        structs.append((name, packed, fields))
    fnptr = re.findall(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", src, flags=re.S)
    funcs = []
    for m in re.finditer(r"^(const char \*|int32_t )\s*(art_\w+)\s*\((.*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        funcs.append((ret, name, params))
    return consts, opaque, structs, fnptr, funcs


def rust_params(params, opaque):
    if params in ("void", ""):
        return ""
    out = []
    for i, p in enumerate(params.split(",")):
        p = p.strip()
        m = re.match(r"^(.*?)(\w+)\s*(\[\w*\])?$", p)
        ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
        if arr:                      # an array parameter is a pointer
            ctype = ctype + " *"
        out.append(f"{field_name(name)}: {rust_type(ctype, opaque)}")
    return ", ".join(out)


def c_layout(structs):
    """sizeof of every struct and offsetof of every field, from gcc"""
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HDR}"', "int main(void) {"]
    for name, _, fields in structs:
        prog.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in fields:
            prog.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    prog.append("  return 0; }")
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(c, "w").write("\n".join(prog))
        subprocess.check_call(["gcc", "-o", exe, c])
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    return dict((k, int(v)) for k, v in (line.split() for line in out.splitlines()))


def generate():
    header = open(HDR).read()
    consts, opaque, structs, fnptr, funcs = parse(header)
    lay = c_layout(structs)
    L = ["// art_sys.rs -- raw binding of include/art.h (libart: the MI355X ray-tracing core that stands in for vk_renderer's Vulkan RT path).",
         "// GENERATED by tools/gen_rust_bindings.py from the header; do not edit.  Where it would live in the reference: src/vk_renderer/art_sys.rs,",
         "// linked with `cargo:rustc-link-lib=art` (libart.so is built by araytracingjourney_amd/csrc/Makefile).  Every function returns 0 or a negative",
         "// ART_E_* code (art_last_error() has the message); the reference panics instead (unwrap / expect): a wrapper would panic on non-zero.",
         "#![allow(non_camel_case_types, dead_code)]", "use core::ffi::{c_char, c_void};", ""]
    for name, val in consts:
        ty = "i32" if name.startswith("ART_E_") or name == "ART_OK" else ("usize" if name.endswith("_BYTES") else "u32")
        L.append(f"pub const {name}: {ty} = {val};")
    L.append("")
    for name in opaque:
        L.append(f"#[repr(C)] pub struct {name} {{ _private: [u8; 0] }}")
    L.append("")
    for ret, name, params in fnptr:
        L.append(f"pub type {name} = Option<unsafe extern \"C\" fn({rust_params(' '.join(params.split()), opaque)}) -> {SCALAR[ret]}>;")
    L.append("")
    for name, packed, fields in structs:
        L.append(f"#[repr(C{', packed' if packed else ''})] #[derive(Clone, Copy)]")
        L.append(f"pub struct {name} {{")
        for fname, rt in fields:
            L.append(f"    pub {field_name(fname)}: {rt},")
        L.append("}")
    L.append("")
    L.append("// layout: sizeof / offsetof of the C structs (gcc, x86-64), checked at compile time")
    for name, packed, fields in structs:
        L.append(f"const _: () = assert!(core::mem::size_of::<{name}>() == {lay[name]});")
        if not packed:   # (offset_of! on a packed struct needs no reference either, but keep to the stable subset)
            for fname, _ in fields:
                L.append(f"const _: () = assert!(core::mem::offset_of!({name}, {field_name(fname)}) == {lay[name + '.' + fname]});")
    L.append("")
    L.append('#[link(name = "art")]')
    L.append('extern "C" {')
    for ret, name, params in funcs:
        r = "*const c_char" if ret.startswith("const char") else "i32"
        L.append(f"    pub fn {name}({rust_params(params, opaque)}) -> {r};")
    L.append("}")
    return "\n".join(L) + "\n", [f[1] for f in funcs]


def main():
    text, _ = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("bindings/art_sys.rs is stale: run python tools/gen_rust_bindings.py")
            sys.exit(1)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count(chr(10))} lines")


if __name__ == "__main__":
    main()
