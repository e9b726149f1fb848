"""The C-ABI library loads and exports every symbol include/nisshi_hip.h declares
(no compute: runs without a GPU); plus the host-side framing entry points."""
import ctypes
import os
import re

import numpy as np
import pytest

from tansu_amd import build, columns as C, hip
from util import load_vectors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return hip.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nisshi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nisshi_\w+)\s*\(", text)))


def test_exports_match_header(lib):
    names = declared_symbols()
    assert sorted(hip.EXPORTS) == names
    for name in names:
        assert getattr(lib, name) is not None


def test_columns_struct_matches_header():
    text = open(os.path.join(ROOT, "include", "nisshi_hip.h")).read()
    body = text[text.index("typedef struct nisshi_columns {"):text.index("} nisshi_columns;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\*(\w+);", body)
    assert fields == [n for n, _, _, _ in C.DEVICE_ARRAYS]
    ctypes_by_np = {np.int64: "int64_t", np.int32: "int32_t", np.int16: "int16_t", np.int8: "int8_t",
                    np.uint64: "uint64_t", np.uint32: "uint32_t", np.uint8: "uint8_t"}
    groups = dict(C.RECORD_GROUPS + C.HEADER_GROUPS)
    struct_of = {"rkey": "nisshi_record_key", "rval": "nisshi_record_value", "hdr": "nisshi_header_span"}
    field_dt = dict(C.ALL_COLUMNS)
    for name, dt, words, _ in C.DEVICE_ARRAYS:
        if words == 1:
            assert re.search(rf"{ctypes_by_np[dt]} \*{name};", body), name
            continue
        # a group: four 32-bit fields, in the order the Python side packs them
        assert re.search(rf"{struct_of[name]} \*{name};", body), name
        sbody = text[text.index(f"typedef struct {struct_of[name]} {{"):text.index(f"}} {struct_of[name]};")]
        sbody = re.sub(r"/\*.*?\*/", "", sbody, flags=re.S)
        got = re.findall(r"(u?int32_t) (\w+);", sbody)
        want = groups[name]
        assert len(got) == 4
        for (ctype, cname), fname in zip(got, want):
            assert ctype == ctypes_by_np[field_dt[fname]], (name, fname)
            # C names drop the prefix that tells records' and headers' fields apart
            assert fname.endswith(cname.replace("hdr_", "rhdr_")) or fname == cname or fname.endswith(cname), (cname, fname)


def test_rows_struct_matches_header():
    text = open(os.path.join(ROOT, "include", "nisshi_hip.h")).read()
    body = text[text.index("typedef struct nisshi_rows {"):text.index("} nisshi_rows;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    got = re.findall(r"(u?int\d+_t) \*(\w+);", body)
    want = [(np.dtype(dt).name + "_t", name)
            for name, dt, _ in hip.ROW_ARRAYS]
    assert got == want


def test_status_strings(lib):
    for st in range(12):
        assert lib.nisshi_hip_status_str(st) not in (None, b"unknown status")
    assert lib.nisshi_hip_status_str(12) == b"unknown status"
    assert b"TryGet" in lib.nisshi_hip_status_str(C.TRY_GET)


def test_ctx_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = ctypes.c_void_p()
    assert lib.nisshi_hip_ctx_create(0, ctypes.byref(ctx)) == 8  # NISSHI_DEVICE_ERROR
    with pytest.raises(RuntimeError):
        hip.Codec(0)


def test_frame_split_matches_oracle(lib):
    import oracle
    raw = bytes(load_vectors()["batch_71"]["bytes"])
    for blob in (b"", raw, raw * 3, raw + raw[:30], raw + raw[:10]):
        st, want = oracle.frame_split(blob)
        if st == C.OK:
            assert list(hip.frame_split(blob)) == list(want)
        else:
            with pytest.raises(hip.NisshiError) as e:
                hip.frame_split(blob)
            assert e.value.status == st
    neg = bytearray(raw)
    neg[8] = 0x80
    with pytest.raises(hip.NisshiError) as e:
        hip.frame_split(bytes(neg))
    assert e.value.status == C.MALFORMED == oracle.frame_split(bytes(neg))[0]


def test_records_prefix(lib):
    import oracle
    buf = ctypes.create_string_buffer(8)
    obuf = ctypes.create_string_buffer(8)
    for n in (0, 1, 71, 127, 128, 16383, 16384, 2 ** 31):
        for flex in (0, 1):
            k = lib.nisshi_records_prefix_encode(n, flex, buf)
            ko = oracle.LIB.no_records_prefix_encode(n, flex, obuf)
            assert buf.raw[:k] == obuf.raw[:ko]
            pos = ctypes.c_uint64(0); ln = ctypes.c_uint32(); null = ctypes.c_int()
            data = np.frombuffer(buf.raw[:k], dtype=np.uint8)
            assert lib.nisshi_records_prefix_decode(data.ctypes.data, k, flex, ctypes.byref(pos),
                                                    ctypes.byref(ln), ctypes.byref(null)) == C.OK
            assert (ln.value, null.value, pos.value) == (n, int(n == 0), k)
    zero = np.zeros(1, dtype=np.uint8)
    pos = ctypes.c_uint64(0); ln = ctypes.c_uint32(); null = ctypes.c_int()
    assert lib.nisshi_records_prefix_decode(zero.ctypes.data, 1, 1, ctypes.byref(pos),
                                            ctypes.byref(ln), ctypes.byref(null)) == C.OVERFLOW
