"""The `extern "C"` block of nisshi-hip/src/ffi.rs from include/nisshi_hip.h: one
declaration a prototype, C types mapped one to one.  tests/test_rust_ffi.py runs
`declarations()` and compares it with what ffi.rs holds.

    python tools/gen_ffi.py            # prints the block
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TYPES = {"int": "c_int", "void": "c_void", "char": "c_char", "double": "f64",
         "uint8_t": "u8", "int8_t": "i8", "uint16_t": "u16", "int16_t": "i16",
         "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64",
         "nisshi_hip_ctx": "Ctx", "nisshi_hip_pool": "Pool", "nisshi_pool_decode_fn": "PoolDecodeFn", "nisshi_columns": "Columns", "nisshi_totals": "Totals",
         "nisshi_rows": "Rows", "nisshi_fetch_rows": "FetchRows", "nisshi_header_span": "HeaderSpan",
         "nisshi_record_key": "RecordKey", "nisshi_record_value": "RecordValue"}


def header_text():
    text = open(os.path.join(ROOT, "include", "nisshi_hip.h")).read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def rust_type(c: str) -> str:
    """`const T *`, `T **`, `T *const *` ... as Rust raw pointers: a `const` in front of
    what a `*` points at makes that level `*const`, anything else `*mut`."""
    tokens = re.findall(r"\w+|\*", c)
    const_pointee = tokens[0] == "const"
    if const_pointee:
        tokens = tokens[1:]
    t = TYPES[tokens[0]]
    i = 1
    while i < len(tokens):
        assert tokens[i] == "*", c
        t = ("*const " if const_pointee else "*mut ") + t
        const_pointee = i + 1 < len(tokens) and tokens[i + 1] == "const"
        i += 2 if const_pointee else 1
    return t


def prototypes():
    """[(name, return type, [(parameter, type)])] in header order, as Rust types."""
    out = []
    for ret, name, args in re.findall(r"\n\s*((?:const\s+)?\w+\s*\**)\s*(nisshi_\w+)\s*\(([^)]*)\)\s*;", header_text()):
        params = []
        for a in args.split(","):
            a = " ".join(a.split())
            if a == "void":     # (no parameters)
                continue
            m = re.match(r"(.*?)(\w+)$", a)
            params.append((m.group(2), rust_type(m.group(1))))
        out.append((name, rust_type(ret), params))
    return out


def declarations():
    lines = []
    for name, ret, params in prototypes():
        args = ", ".join(f"{p}: {t}" for p, t in params)
        lines.append(f"    pub fn {name}({args})" + ("" if ret == "c_void" else f" -> {ret}") + ";")
    return lines


def struct_fields(c_name: str):
    """[(field, Rust type)] of a struct of the header."""
    text = header_text()
    body = text[text.index(f"typedef struct {c_name} {{"):text.index(f"}} {c_name};")]
    return [(f, rust_type(t)) for t, f in re.findall(r"\n\s*((?:const\s+)?\w+\s*\**)\s*(\w+);", body)]


if __name__ == "__main__":
    print("\n".join(declarations()))
