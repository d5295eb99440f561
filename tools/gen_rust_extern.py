#!/usr/bin/env python3
"""Prints the Rust `extern "C"` declaration of every entry point include/bitnet_hip.h declares (one line each), or with --missing only those
INTEGRATION.md section 2 does not spell out yet.  A mechanical map of the C prototypes (size_t -> usize, const T * -> *const T, ...); the two
by-value / by-pointer structs are referred to by the names INTEGRATION.md gives them.   python tools/gen_rust_extern.py [--missing]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = {"size_t": "usize", "int": "c_int", "float": "c_float", "double": "f64", "void": "c_void", "char": "c_char", "uint8_t": "u8", "int8_t": "i8",
        "uint16_t": "u16", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "unsigned long long": "u64", "unsigned": "u32",
        "bitnet_hip_weights_t": "Weights", "bitnet_hip_device_info": "DeviceInfo", "bitnet_hip_gemv_item": "GemvItem", "_Float16": "u16"}


def rust_type(c: str) -> str:
    c = c.strip()
    stars = c.count("*")
    c = c.replace("*", " ").replace("struct ", " ")
    const = bool(re.search(r"\bconst\b", c))
    base = re.sub(r"\bconst\b", " ", c).strip()
    base = re.sub(r"\s+", " ", base)
    t = BASE[base]
    for i in range(stars):
        t = ("*const " if const and i == 0 else "*mut ") + t
    return t


def prototypes(header: str):
    text = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(bitnet_hip_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                params.append((mm.group(2), rust_type(mm.group(1))))
        r = "" if ret == "void" else " -> " + rust_type(ret)
        yield name, "pub fn %s(%s)%s;" % (name, ", ".join(f"{n}: {t}" for n, t in params), r)


def main():
    header = open(os.path.join(ROOT, "include", "bitnet_hip.h")).read()
    have = set(re.findall(r"pub fn (bitnet_hip_[a-z0-9_]+)", open(os.path.join(ROOT, "INTEGRATION.md")).read())) if "--missing" in sys.argv else set()
    for name, line in prototypes(header):
        if name not in have:
            print("    " + line)


if __name__ == "__main__":
    main()
