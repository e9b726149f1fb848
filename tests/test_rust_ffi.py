"""The Rust side (nisshi-hip/, patches/nisshi-sans-io-hip.diff) cannot be compiled
here: no Rust toolchain.  What can be checked without one is that its `extern "C"`
block and its #[repr(C)] structs say what include/nisshi_hip.h says: the same
symbols, the same parameters in the same order with the matching types, the same
fields in the same order; that the status constants agree; and that the patch
applies to the files it names when the reference tree is present."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_ffi  # noqa: E402

FFI = open(os.path.join(ROOT, "nisshi-hip", "src", "ffi.rs")).read()


def rust_functions():
    block = FFI[FFI.index('extern "C" {'):]
    out = []
    for name, args, ret in re.findall(r"pub fn (\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block):
        params = []
        for a in filter(None, (a.strip() for a in args.split(","))):
            p, t = a.split(":", 1)
            params.append((p.strip(), " ".join(t.split())))
        out.append((name, (ret or "c_void").strip(), params))
    return out


def rust_struct(name):
    m = re.search(r"#\[repr\(C\)\]\n(?:#\[derive[^\n]*\n)?pub struct %s \{(.*?)\n\}" % name, FFI, flags=re.S)
    assert m, name
    return [(f, " ".join(t.split())) for f, t in re.findall(r"pub (\w+): ([^,]+),", m.group(1))]


def test_every_export_is_declared_with_the_header_s_signature():
    want = gen_ffi.prototypes()
    got = rust_functions()
    assert [n for n, _, _ in got] == [n for n, _, _ in want]
    for (name, ret, params), (_, wret, wparams) in zip(got, want):
        assert ret == wret, name
        assert params == wparams, name


def test_exports_are_those_of_the_python_binding_too():
    from tansu_amd import hip
    assert sorted(n for n, _, _ in rust_functions()) == sorted(hip.EXPORTS)


@pytest.mark.parametrize("rust,c", [("RecordKey", "nisshi_record_key"), ("RecordValue", "nisshi_record_value"),
                                    ("HeaderSpan", "nisshi_header_span"), ("Columns", "nisshi_columns"),
                                    ("Totals", "nisshi_totals"), ("Rows", "nisshi_rows"),
                                    ("FetchRows", "nisshi_fetch_rows")])
def test_repr_c_structs_have_the_header_s_fields_in_order(rust, c):
    assert rust_struct(rust) == gen_ffi.struct_fields(c)


def test_status_constants_and_limits():
    text = gen_ffi.header_text()
    enum = dict(re.findall(r"(NISSHI_[A-Z_]+) = (\d+)", text))
    assert len(enum) == 12
    for name, value in enum.items():
        assert re.search(rf"pub const {name}: c_int = {value};", FFI), name
    assert re.search(r"#define NISSHI_STATIC_MAX_BATCHES 16384u", text) and "NISSHI_STATIC_MAX_BATCHES: u32 = 16384;" in FFI
    assert re.search(r"#define NISSHI_STATIC_MAX_BYTES \(512ull << 20\)", text) and "NISSHI_STATIC_MAX_BYTES: u64 = 512 << 20;" in FFI


def test_safe_layer_uses_only_declared_symbols():
    lib = open(os.path.join(ROOT, "nisshi-hip", "src", "lib.rs")).read()
    declared = {n for n, _, _ in rust_functions()}
    used = set(re.findall(r"ffi::(nisshi_\w+)", lib))
    assert used and used <= declared
    # HostColumns::as_ffi fills every field of Columns
    fields = [f for f, _ in rust_struct("Columns")]
    body = lib[lib.index("fn as_ffi"):lib.index("pub fn status_of")]
    assert re.findall(r"\n\s+(\w+): self\.\w+\.as_mut_ptr\(\),", body) == fields


def test_patch_names_the_reference_s_hook_points():
    patch = open(os.path.join(ROOT, "patches", "nisshi-sans-io-hip.diff")).read()
    for path in ("Cargo.toml", "nisshi-sans-io/Cargo.toml", "nisshi-sans-io/src/record.rs",
                 "nisshi-sans-io/src/record/inflated.rs", "nisshi-sans-io/src/record/deflated.rs",
                 "nisshi-sans-io/src/record/hip.rs"):
        assert f"+++ b/{path}" in patch, path
    assert 'hip = ["dep:nisshi-hip"]' in patch and '#[cfg(feature = "hip")]' in patch
    ref = "/root/reference"
    if not (os.path.isdir(ref) and shutil.which("patch")):
        pytest.skip("the reference tree is not here")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for path in re.findall(r"^--- a/(\S+)", patch, flags=re.M):
            src = os.path.join(ref, path)
            if os.path.exists(src):
                os.makedirs(os.path.dirname(os.path.join(tmp, path)), exist_ok=True)
                shutil.copy(src, os.path.join(tmp, path))
        out = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "patches", "nisshi-sans-io-hip.diff")],
                             cwd=tmp, capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
