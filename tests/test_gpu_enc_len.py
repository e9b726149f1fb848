"""The optional `enc_len` column: the decode notes a batch's record bytes (plus one)
where every varint of its records is written the shortest way, so that the encoder
need not size those records again; anything else leaves zero and is sized from the
columns.  Either way the bytes that come out are the CPU checker's."""
import struct

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip, synth
from util import offsets_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def wide_codec():
    """A context whose calls all go to the kernel of large batches, the one that notes sizes
    (a call whose batches average less than 64 KiB otherwise goes to the other)."""
    import os
    os.environ["NISSHI_WIDE_RING_FROM"] = "0"
    try:
        c = hip.Codec(0)
    finally:
        del os.environ["NISSHI_WIDE_RING_FROM"]
    yield c
    c.close()


def decode(codec, blob, off):
    blob = np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    n = len(off) - 1
    d_blob, d_off = codec.to_device(blob, off)
    rec_cap = codec.count_dev(d_blob, len(blob), d_off, n)
    cols = hip.DeviceColumns(n, rec_cap, len(blob) // 2 + 1, codec.device)
    codec.decode_dev(d_blob, len(blob), d_off, n, cols, True)
    return blob, cols, d_blob


def encode(codec, cols, n, d_blob, cap):
    import torch
    out = torch.zeros(cap, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n, d_blob, cols.t["data_off"], out, cap, out_off)
    return out[:t.out_bytes].cpu().numpy().tobytes(), out_off.cpu().numpy()


def checker_bytes(blob, off):
    st, c, nr, nh = oracle.decode_columns(blob, off, True)
    assert st == C.OK
    st, want, want_off = oracle.encode_columns(c, len(off) - 1, blob, c["data_off"])
    assert st == C.OK
    return c, want


def record(key: bytes, value: bytes, ts: bytes, od: bytes, klen: bytes = None, hc: bytes = b"\x00") -> bytes:
    """One record from the bytes of its varints (so that they can be padded)."""
    klen = bytes([2 * len(key)]) if klen is None else klen
    body = b"\x00" + ts + od + klen + key + bytes([2 * len(value)]) + value + hc
    assert len(body) < 64
    return bytes([2 * len(body)]) + body


def batch(data: bytes, count: int) -> bytes:
    return struct.pack(">qiibIhiqqqhiI", 0, 49 + len(data), -1, 2, 0, 0, count - 1, 1, 2, -1, -1, -1, count) + data


@pytest.mark.parametrize("name,scale", [("c1", 0.02), ("c2", 0.004), ("c3", 0.02), ("c5", 0.02)])
def test_sizes_carried_from_the_decode(codec, name, scale):
    shape, counts, seed = synth.config(name, scale)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    blob, cols, d_blob = decode(codec, blob, off)
    n = len(off) - 1
    c, want = checker_bytes(blob, off)
    enc_len = cols.t["enc_len"].cpu().numpy().view(np.uint32)[:n]
    known = enc_len != 0
    # what is noted is right, and the generator's batches (all shortest varints) are noted
    # wherever waves walked them by their shapes (not a lane a batch, nor records longer than a wave's window)
    assert (enc_len[known] - 1 == c["data_len"][:n][known]).all()
    if name == "c2":
        assert known.all()
    if name == "c5":
        assert known.any()
    if name == "c1":    # (batches of 12 KiB: the kernel of small batches notes none)
        assert not known.any()
    with_sizes, off_a = encode(codec, cols, n, d_blob, len(blob) + 64)
    cols.t["enc_len"].zero_()
    without, off_b = encode(codec, cols, n, d_blob, len(blob) + 64)
    assert with_sizes == want and without == want and (off_a == off_b).all()


def test_small_batches_through_the_kernel_that_notes_sizes(wide_codec):
    shape, counts, seed = synth.config("c1", 0.02)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    blob, cols, d_blob = decode(wide_codec, blob, off)
    n = len(off) - 1
    c, want = checker_bytes(blob, off)
    enc_len = cols.t["enc_len"].cpu().numpy().view(np.uint32)[:n]
    assert (enc_len == c["data_len"][:n] + 1).all()
    got, _ = encode(wide_codec, cols, n, d_blob, len(blob) + 64)
    assert got == want


def test_records_of_changing_sizes_carry_their_size_too(codec):
    """Records whose sizes change from one to the next are walked along their lengths (the chained walk),
    which looks at how every varint is written as well; one padded varint in one record and the batch is sized
    by the encoder again."""
    blob, off, nr, nh = synth.make_var_batches(12, 1000, 31)
    blob = bytearray(blob.tobytes())
    st, c0, _, _ = oracle.decode_columns(bytes(blob), off, True)
    # batch 5, record 700: its key length (16: one byte, 0x20) becomes 0xa0 0x00 -- two bytes for the same
    # value -- and its length varint one more; what follows moves up a byte
    at = int(off[5]) + 61 + int(c0["key_off"][5 * 1000 + 700]) - 1
    assert blob[at] == 0x20
    rec = at - 1 - 2 - 2 - 2      # (the record's start: length of two bytes, attributes, two deltas of two bytes)
    length = (blob[rec] & 0x7f) | (blob[rec + 1] << 7)
    assert blob[rec] & 0x80 and not blob[rec + 1] & 0x80 and length // 2 > 100
    length += 2                   # (zigzag: the body is one byte longer)
    patched = blob[:rec] + bytes([0x80 | (length & 0x7f), length >> 7]) + blob[rec + 2:at] + b"\xa0\x00" + blob[at + 1:]
    head = bytearray(patched[int(off[5]):int(off[5]) + 61])
    struct.pack_into(">i", head, 8, struct.unpack_from(">i", head, 8)[0] + 1)
    patched[int(off[5]):int(off[5]) + 61] = head
    off = off.copy()
    off[6:] += 1
    blob, cols, d_blob = decode(codec, bytes(patched), off)
    n = len(off) - 1
    assert (cols.t["status"].cpu().numpy()[:n] == C.OK).all()
    c, want = checker_bytes(blob, off)
    enc_len = cols.t["enc_len"].cpu().numpy().view(np.uint32)[:n]
    expect = c["data_len"][:n] + 1
    expect[5] = 0
    assert (enc_len == expect).all()
    got, _ = encode(codec, cols, n, d_blob, len(blob) + 64)
    assert got == want and len(want) == len(blob) - 1


def uv(n: int) -> bytes:
    """The zigzag varint of n, written the shortest way."""
    z = (n << 1) ^ (n >> 63)
    out = bytearray()
    while z >= 0x80:
        out.append(0x80 | (z & 0x7f))
        z >>= 7
    out.append(z)
    return bytes(out)


def test_padded_varints_are_not_carried(wide_codec):
    codec = wide_codec
    # (batches of more than 1 KiB: smaller ones are walked a lane each, which notes no size)
    n = 100
    plain = [record(b"k%02d" % i, b"value of twenty bytes", uv(i), uv(i)) for i in range(n)]
    cases = {"plain": list(plain)}   # (the deltas grow to two bytes at record 64)
    padded_ts = list(plain)
    padded_ts[30] = record(b"k30", b"value of twenty bytes", b"\xbc\x00", uv(30))
    cases["timestamp delta"] = padded_ts
    padded_od = list(plain)
    padded_od[n - 1] = record(b"k99", b"value of twenty bytes", uv(n - 1), b"\xc6\x81\x00")
    cases["offset delta"] = padded_od
    padded_key = list(plain)
    padded_key[17] = record(b"k17", b"value of twenty bytes", uv(17), uv(17), klen=b"\x86\x00")
    cases["key length"] = padded_key
    padded_hc = list(plain)
    padded_hc[5] = record(b"k05", b"value of twenty bytes", uv(5), uv(5), hc=b"\x80\x00")
    cases["header count"] = padded_hc
    padded_len = list(plain)
    r = padded_len[0]
    padded_len[0] = bytes([r[0] | 0x80, 0x00]) + r[1:]
    cases["length"] = padded_len
    padded_len_later = list(plain)
    r = padded_len_later[70]
    padded_len_later[70] = bytes([r[0] | 0x80, 0x00]) + r[1:]
    cases["length of a later record"] = padded_len_later
    cases["wide deltas"] = [record(b"k%02d" % i, b"value of twenty bytes", uv(1000 + i), uv(i)) for i in range(n)]
    names = list(cases)
    batches = [batch(b"".join(cases[k]), n) for k in names]
    assert min(map(len, batches)) > 1024
    blob = b"".join(batches)
    off = offsets_of(batches)
    blob, cols, d_blob = decode(codec, blob, off)
    assert (cols.t["status"].cpu().numpy()[:len(names)] == C.OK).all()
    enc_len = dict(zip(names, cols.t["enc_len"].cpu().numpy().view(np.uint32)))
    for k in names:
        data = len(b"".join(cases[k]))
        assert enc_len[k] == (data + 1 if k in ("plain", "wide deltas") else 0), k
    c, want = checker_bytes(blob, off)
    got, _ = encode(codec, cols, len(names), d_blob, len(blob) + 64)
    assert got == want and len(want) < len(blob)


@pytest.mark.parametrize("key_len,val_len,headers,placed", [(8, 40, 0, True), (-1, 40, 1, True), (5, -1, 0, True),
                                                             (-1, -1, 2, False)])
def test_large_batches_of_known_size_are_written_in_pieces(wide_codec, key_len, val_len, headers, placed):
    """A batch of 2048 records and more is written in pieces of 1024: with its size carried
    over, a piece starts where its first record's key (or value) says; a record with neither
    leaves the batch to be sized as before (`placed` false).  The bytes are the checker's either way."""
    codec = wide_codec
    shape = synth.Shape(key_len=key_len, val_len=val_len, n_headers=headers, hkey_len=3, hval_len=5)
    counts = np.array([5000, 2048, 3000, 7001, 100, 2047, 4096])
    blob, off, nr, nh = synth.make_batches(shape, counts, 77)
    blob, cols, d_blob = decode(codec, blob, off)
    n = len(off) - 1
    c, want = checker_bytes(blob, off)
    enc_len = cols.t["enc_len"].cpu().numpy().view(np.uint32)[:n]
    assert (enc_len == c["data_len"][:n] + 1).all()
    got, got_off = encode(codec, cols, n, d_blob, len(blob) + 64)
    assert got == want


def test_host_buffers_carry_the_sizes_too(codec):
    """nisshi_hip_decode hands the sizes back with the other columns and nisshi_hip_encode takes them (or none:
    a caller's struct may leave the pointer NULL, which hip.py does when the column is missing)."""
    shape, counts, seed = synth.config("c2", 0.002)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    cols, totals = codec.decode(blob.tobytes(), off)
    n = len(off) - 1
    assert (cols["enc_len"][:n] == cols["data_len"][:n] + 1).all()
    c, want = checker_bytes(blob, off)
    got, got_off, t = codec.encode(cols, blob.tobytes(), cols["data_off"])
    assert got == want
    without = {k: v for k, v in cols.items() if k != "enc_len"}
    got2, _, _ = codec.encode(without, blob.tobytes(), cols["data_off"])
    assert got2 == want


def test_operators_forget_the_sizes(codec):
    """Batches written by combine carry no size: their records are not those that were sized."""
    shape, counts, seed = synth.config("c5", 0.01)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    blob, cols, d_blob = decode(codec, blob, off)
    n = len(off) - 1
    assert (cols.t["enc_len"].cpu().numpy() != 0).any()
    import torch
    groups = n // 2
    first = torch.arange(0, 2 * groups + 1, 2, dtype=torch.int32, device=codec.device)
    sink = codec.combine_dev(cols, n, cols.rec_cap, first, groups)
    assert (sink.t["enc_len"].cpu().numpy()[:groups] == 0).all()
    # and the sizes of the members whose deltas were rebased go with them (the first of a group keeps its deltas)
    assert (cols.t["enc_len"].cpu().numpy()[1:2 * groups:2] == 0).all()
