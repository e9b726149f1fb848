#!/usr/bin/env python3
"""Transcribe the golden vectors held by the reference's own tests into JSON.

The reference (tansu / nisshi-sans-io) is Rust and cannot be built here, so the
parity oracle is pinned by the data its tests hold: wire bytes of frames and
batches, and the fully expanded `inflated::Batch { .. records: [Record {..}] }`
values they are asserted equal to.  This script reads those test sources AS
TEXT, pulls out the byte arrays and parses the struct literals into plain data,
and writes tests/golden/vectors.json.  Only data is copied: numbers and byte
strings.  Run it in the build container (the reference tree is not present on
the GPU box); the tests read only the JSON.

    python tests/golden/extract_vectors.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SANS = os.path.join(REF, "nisshi-sans-io")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")

LOREM = (
    b"Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do "
    b"eiusmod tempor incididunt ut labore et dolore magna aliqua. Ut enim ad "
    b"minim veniam, quis nostrud exercitation ullamco laboris nisi ut "
    b"aliquip ex ea commodo consequat. Duis aute irure dolor in "
    b"reprehenderit in voluptate velit esse cillum dolore eu fugiat nulla "
    b"pariatur. Excepteur sint occaecat cupidatat non proident, sunt in "
    b"culpa qui officia deserunt mollit anim id est laborum."
)


def read(path):
    with open(os.path.join(SANS, path), encoding="utf-8") as f:
        return f.read()


def functions(src):
    """Yield (name, first_line, body) for every `fn name(` in a source file."""
    heads = [(m.start(), m.group(1)) for m in re.finditer(r"^\s*(?:pub )?fn (\w+)\s*\(", src, re.M)]
    for i, (pos, name) in enumerate(heads):
        end = heads[i + 1][0] if i + 1 < len(heads) else len(src)
        yield name, src.count("\n", 0, pos) + 1, src[pos:end]


def int_list(text):
    return [int(t.replace("_", "")) for t in re.findall(r"-?\d[\d_]*", text)]


def byte_arrays(body):
    """All `vec![..]` / `&[..]` / `[..]` integer arrays of >= 8 elements, in order."""
    out = []
    for m in re.finditer(r"\[([\s\d,_]+)\]", body):
        vals = int_list(m.group(1))
        if len(vals) >= 8 and all(0 <= v <= 255 for v in vals):
            out.append(vals)
    return out


# ---- a tiny parser for the Rust struct literals the tests use -------------


class P:
    def __init__(self, s):
        self.s, self.i = s, 0

    def ws(self):
        while self.i < len(self.s) and self.s[self.i] in " \t\r\n":
            self.i += 1

    def eat(self, tok):
        self.ws()
        if self.s.startswith(tok, self.i):
            self.i += len(tok)
            return True
        return False

    def expect(self, tok):
        if not self.eat(tok):
            raise ValueError(f"expected {tok!r} at {self.s[self.i:self.i+60]!r}")

    def ident(self):
        self.ws()
        m = re.compile(r"[A-Za-z_]\w*(?:::\w+)*").match(self.s, self.i)
        if not m:
            return None
        self.i = m.end()
        return m.group(0)

    def suffix(self):
        """Swallow `.into()`, `.try_into()?`, `.clone()` and friends."""
        while True:
            self.ws()
            m = re.compile(r"\.\s*\w+\s*\(\s*\)\s*\??").match(self.s, self.i)
            if not m:
                return
            self.i = m.end()

    def value(self):
        self.ws()
        c = self.s[self.i]
        if c == "-" or c.isdigit():
            m = re.compile(r"-?\d[\d_]*(?:[iu](?:8|16|32|64|size))?").match(self.s, self.i)
            self.i = m.end()
            v = int(re.match(r"-?\d[\d_]*", m.group(0)).group(0).replace("_", ""))
            self.suffix()
            return v
        if c == "&":
            self.i += 1
            return self.value()
        if c == "[":
            self.i += 1
            items = []
            while not self.eat("]"):
                items.append(self.value())
                self.eat(",")
            self.suffix()
            return items
        if c == "b" and self.s[self.i + 1] == '"':
            m = re.compile(r'b"((?:[^"\\]|\\.)*)"').match(self.s, self.i)
            self.i = m.end()
            raw = m.group(1)
            raw = re.sub(r"\\\n\s*", "", raw)  # line continuation
            data = raw.encode("utf-8").decode("unicode_escape").encode("latin-1")
            self.suffix()
            return {"bytes": list(data)}
        name = self.ident()
        if name is None:
            raise ValueError(f"unexpected {self.s[self.i:self.i+60]!r}")
        if name == "None":
            return None
        if name == "LOREM":
            self.suffix()
            return {"bytes": list(LOREM)}
        if name == "vec":
            self.expect("!")
            return self.value()
        if self.eat("("):
            args = []
            while not self.eat(")"):
                args.append(self.value())
                self.eat(",")
            self.suffix()
            if name == "Some":
                return args[0]
            if name.startswith("Bytes::"):
                a = args[0]
                if isinstance(a, dict):
                    return a
                return {"bytes": a}
            raise ValueError(f"unknown call {name}")
        if self.eat("{"):
            fields = {}
            while not self.eat("}"):
                k = self.ident()
                self.expect(":")
                fields[k] = self.value()
                self.eat(",")
            self.suffix()
            fields["_type"] = name.split("::")[-1]
            return fields
        raise ValueError(f"unknown identifier {name}")


def octets(v):
    if v is None:
        return None
    if isinstance(v, dict) and "bytes" in v:
        return v["bytes"]
    if isinstance(v, list):
        return v
    raise ValueError(v)


def norm_record(r):
    return {
        "length": r["length"],
        "attributes": r["attributes"],
        "timestamp_delta": r["timestamp_delta"],
        "offset_delta": r["offset_delta"],
        "key": octets(r["key"]),
        "value": octets(r["value"]),
        "headers": [{"key": octets(h["key"]), "value": octets(h["value"])} for h in r["headers"]],
    }


BATCH_FIELDS = [
    "base_offset", "batch_length", "partition_leader_epoch", "magic", "crc", "attributes",
    "last_offset_delta", "base_timestamp", "max_timestamp", "producer_id", "producer_epoch",
    "base_sequence",
]


def inflated_batches(body):
    out = []
    for m in re.finditer(r"inflated::Batch\s*\{", body):
        p = P(body)
        p.i = m.start()
        try:
            v = p.value()
        except Exception as e:  # a literal using syntax this parser does not know
            print("  skip literal:", e, file=sys.stderr)
            continue
        b = {k: v[k] for k in BATCH_FIELDS}
        b["records"] = [norm_record(r) for r in v["records"]]
        out.append(b)
    return out


def frame_cases(path, names):
    src = read(path)
    cases = []
    for name, line, body in functions(src):
        if name not in names:
            continue
        arrays = byte_arrays(body)
        frame = arrays[0]
        cases.append({
            "name": name,
            "source": f"nisshi-sans-io/{path}:{line}",
            "frame": frame,
            "batches": inflated_batches(body),
        })
    return cases


def main():
    vec = {}

    # -- whole frames with fully expanded inflated batches (tests/decode.rs)
    vec["decode_frames"] = frame_cases("tests/decode.rs", {
        "fetch_response_v12_001", "fetch_response_v12_002", "fetch_response_v16_001",
        # produce_request_v0_000 (decode.rs:4818) is left out: it is #[ignore]d, its
        # bytes are a magic-0 message set and do not match the value it is compared to
        "fetch_response_v16_002", "produce_request_v3_000",
        "produce_request_v7_000", "produce_request_v9_000", "produce_request_v9_001",
        "produce_request_v10_000", "produce_request_v10_001",
    })
    # -- typed value -> wire bytes (tests/encode.rs); the byte array is the expected frame
    enc = []
    src = read("tests/encode.rs")
    for name, line, body in functions(src):
        if name in {"produce_request_v9_000", "fetch_response_v12_001", "fetch_response_v16_001",
                    "fetch_response_v16_002"}:
            arrays = byte_arrays(body)
            enc.append({
                "name": name,
                "source": f"nisshi-sans-io/tests/encode.rs:{line}",
                "frame": max(arrays, key=len),
                "batches": inflated_batches(body),
            })
    vec["encode_frames"] = enc

    # -- compressed batches: whole frames with their inflated batches (tests/decode.rs), the three
    #    producer frames of tests/snappy.rs with what is asserted of their one record, and the bare
    #    batches of the codec tests in src/record/deflated.rs (one record whose value is LOREM)
    vec["compressed_frames"] = frame_cases("tests/decode.rs", {"produce_request_v10_002", "produce_request_v10_003"})
    src = read("tests/snappy.rs")
    snappy = []
    for name, line, body in functions(src):
        arrays = byte_arrays(body)
        if not arrays:
            continue
        asserted = {}
        for want, what in re.findall(r"assert_eq!\(\s*(\d+)\s*,\s*([\w.()]+)\s*\)", body):
            asserted[what] = int(want)
        # an #[ignore]d test is one the reference does not pass: its frame is kept as a case that must fail
        ignored = re.search(r"#\[ignore\]\s*(?:#\[test\]\s*)?fn\s+" + name + r"\b", src) is not None
        snappy.append({"name": name, "source": f"nisshi-sans-io/tests/snappy.rs:{line}", "frame": arrays[0],
                       "ignored": ignored,
                       "record_count": asserted["deflated.record_count"], "records": asserted["inflated.records.len()"],
                       "first_record_length": asserted["record.length"]})
    vec["snappy_frames"] = snappy
    src = read("src/record/deflated.rs")
    codecs = []
    for name, line, body in functions(src):
        if name in {"decode_gzip", "decode_zstd", "decode_lz4", "decode_snappy"}:
            arrays = byte_arrays(body)
            m = re.search(r"vec!\[\s*Record\s*\{", body)
            q = P(body)
            q.i = m.start()
            records = [norm_record(r) for r in q.value()]
            codecs.append({"name": name, "source": f"nisshi-sans-io/src/record/deflated.rs:{line}",
                           "batch": arrays[0], "records": records})
    vec["compressed_batches"] = codecs

    # -- criterion bench inputs (benches/decode_bench.rs): bytes only
    bench = []
    src = read("benches/decode_bench.rs")
    for name, line, body in functions(src):
        arrays = byte_arrays(body)
        if arrays and name in {"produce_request_v10_001", "produce_request_v11_001"}:
            bench.append({"name": name, "source": f"nisshi-sans-io/benches/decode_bench.rs:{line}",
                          "frame": max(arrays, key=len)})
    vec["bench_frames"] = bench

    # -- unit vectors, transcribed by hand from the cited lines
    vec["record_def"] = {
        "source": "nisshi-sans-io/src/record.rs:478-523",
        "bytes": [18, 0, 0, 0, 1, 6, 100, 101, 102, 0],
        "record": {"length": 9, "attributes": 0, "timestamp_delta": 0, "offset_delta": 0,
                   "key": None, "value": [100, 101, 102], "headers": []},
        "body_size": 9,           # record.rs:465-475, 526-533
        "sequence_size": 14,      # record.rs:536-553
        "sequence_bytes": [0, 0, 0, 1, 18, 0, 0, 0, 1, 6, 100, 101, 102, 0],
    }
    vec["crc_known_answer"] = {
        "source": "nisshi-sans-io/src/record.rs:556-570",
        "bytes": [0, 0, 0, 0, 0, 0, 0, 0, 1, 141, 116, 152, 137, 53, 0, 0, 1, 141, 116, 152, 137,
                  53, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 18, 0, 0, 0, 1, 6, 100,
                  101, 102, 0],
        "crc": 1126819645,
    }
    vec["batch_71"] = {
        "source": "nisshi-sans-io/src/record/inflated.rs:533-618",
        "bytes": [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 59, 255, 255, 255, 255, 2, 67, 41, 231, 61, 0,
                  0, 0, 0, 0, 0, 0, 0, 1, 141, 116, 152, 137, 53, 0, 0, 1, 141, 116, 152, 137, 53,
                  0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 18, 0, 0, 0, 1, 6, 100,
                  101, 102, 0],
        "batch": {"base_offset": 0, "batch_length": 59, "partition_leader_epoch": -1, "magic": 2,
                  "crc": 1126819645, "attributes": 0, "last_offset_delta": 0,
                  "base_timestamp": 1707058170165, "max_timestamp": 1707058170165,
                  "producer_id": 1, "producer_epoch": 0, "base_sequence": 1,
                  "records": [{"length": 9, "attributes": 0, "timestamp_delta": 0,
                               "offset_delta": 0, "key": None, "value": [100, 101, 102],
                               "headers": []}]},
    }
    vec["octets_sizes"] = {
        "source": "nisshi-sans-io/src/record/codec.rs:577-623, src/record.rs:457-462",
        "cases": [
            {"octets": None, "size": 1},                       # codec.rs:578-584
            {"octets": list(b"34543"), "size": 6},             # codec.rs:587-593
            {"octets": [100, 101, 102], "size": 4},            # record.rs:457-462
        ],
        # VarIntSequence<Octets>: zigzag count, then the items
        "sequences": [
            {"items": [], "size": 1},                                   # codec.rs:596-602
            {"items": [list(b"34543")], "size": 7},                     # codec.rs:605-611
            {"items": [list(b"34543"), list(b"32123")], "size": 13},    # codec.rs:614-623
        ],
    }
    vec["control_batch"] = {
        "source": "nisshi-sans-io/src/record/deflated.rs:935-985",
        "batch": {"base_offset": 1, "batch_length": 66, "partition_leader_epoch": 0, "magic": 2,
                  "crc": 820655041, "attributes": 48, "last_offset_delta": 0,
                  "base_timestamp": 1729509916024, "max_timestamp": 1729509916024,
                  "producer_id": 5, "producer_epoch": 0, "base_sequence": -1, "record_count": 1},
        "record_data": list(b" \0\0\0\x08\0\0\0\x01\x0c\0\0\0\0\0\0\0"),
        "is_transactional": True,
        "is_control": True,
        "key": [0, 0, 0, 1],
        "value": [0, 0, 0, 0, 0, 0],
    }
    # saved proptest failures: inputs of the inflate <-> deflate round trip
    # (tests/proptest.rs:225-295); only the uncompressed case is on this path
    reg = read("tests/proptest.proptest-regressions")
    cases = []
    for line in reg.splitlines():
        if not line.startswith("cc "):
            continue
        def num(name):
            return int(re.search(name + r" = (-?\d+)", line).group(1))
        def blobs(name):
            body = re.search(name + r" = \[(.*?)\](?:, rec_|$)", line).group(1)
            out = []
            for m in re.finditer(r'None|Some\(b"((?:[^"\\]|\\.)*)"\)', body):
                if m.group(0) == "None":
                    out.append(None)
                else:
                    raw = re.sub(r"\\0(?![0-7])", r"\\x00", m.group(1))
                    out.append(list(raw.encode("utf-8").decode("unicode_escape").encode("latin-1")))
            return out
        cases.append({
            "base_offset": num("base_offset"), "partition_leader_epoch": num("partition_leader_epoch"),
            "base_timestamp": num("base_timestamp"), "max_timestamp": num("max_timestamp"),
            "producer_id": num("producer_id"), "producer_epoch": num("producer_epoch"),
            "base_sequence": num("base_sequence"),
            "compression": re.search(r"compression = (\w+)", line).group(1),
            "rec_keys": blobs("rec_keys"), "rec_values": blobs("rec_values"),
        })
    vec["proptest_regressions"] = {
        "source": "nisshi-sans-io/tests/proptest.proptest-regressions:7-8",
        "cases": cases,
    }
    vec["varints"] = {
        "source": "nisshi-sans-io/src/primitive/varint.rs:639-686",
        "signed": [{"value": 1, "bytes": [2]}, {"value": 0, "bytes": [0]}],
        "unsigned": [{"value": 1, "bytes": [1]}],
    }
    vec["attributes"] = {
        "source": "nisshi-sans-io/src/lib.rs:2316-2348",
        "bits": {"none": 0, "gzip": 1, "snappy": 2, "lz4": 3, "zstd": 4, "log_append_time": 8,
                 "transaction": 16, "control": 32, "delete_horizon": 64},
    }
    vec["bad_compression"] = {
        "source": "nisshi-sans-io/tests/gap6.rs:224-238",
        "codes": [5, 6, 7],
    }

    with open(OUT, "w") as f:
        json.dump(vec, f, separators=(",", ":"))
    n = sum(len(c["batches"]) for c in vec["decode_frames"] + vec["encode_frames"])
    print(f"wrote {OUT}: {len(vec['decode_frames'])} decode frames, "
          f"{len(vec['encode_frames'])} encode frames, {n} expanded batches, "
          f"{len(vec['bench_frames'])} bench frames")


if __name__ == "__main__":
    main()
