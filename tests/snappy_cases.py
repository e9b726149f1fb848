"""Snappy streams for the tests: a small compressor (greedy, hash of four bytes: valid streams of
literals and of all three kinds of copies, nothing more is asked of it) and streams that must not inflate."""
import struct


def uvarint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def literal(data: bytes) -> bytes:
    n = len(data) - 1
    if n < 60:
        return bytes([n << 2]) + data
    for extra in (1, 2, 3, 4):
        if n < 1 << (8 * extra):
            return bytes([(59 + extra) << 2]) + n.to_bytes(extra, "little") + data
    raise ValueError(n)


def copy(offset: int, length: int) -> bytes:
    out = bytearray()
    while length:
        n = min(length, 64)
        if length - n in (1, 2, 3):          # never leave a copy too short to encode well
            n -= 4
        if 4 <= n <= 11 and offset < 2048:
            out += bytes([1 | (n - 4) << 2 | (offset >> 8) << 5, offset & 0xFF])
        elif offset < 65536:
            out += bytes([2 | (n - 1) << 2]) + struct.pack("<H", offset)
        else:
            out += bytes([3 | (n - 1) << 2]) + struct.pack("<I", offset)
        length -= n
    return bytes(out)


def compress(data: bytes, xerial: bool = False, window: int = 1 << 20) -> bytes:
    """One raw block (behind the 20 bytes the reference skips when xerial): greedy matches of four bytes and more."""
    out = bytearray(uvarint(len(data)))
    table = {}
    i, lit = 0, 0
    n = len(data)
    while i + 4 <= n:
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j <= window:
            m = 4
            while i + m < n and data[j + m] == data[i + m] and m < 1000:
                m += 1
            if lit < i:
                out += literal(data[lit:i])
            out += copy(i - j, m)
            i += m
            lit = i
        else:
            i += 1
    if lit < n:
        out += literal(data[lit:])
    body = bytes(out)
    if xerial:
        return b"\x82SNAPPY\x00" + struct.pack(">iii", 1, 1, len(body)) + body
    return body


def corrupt_cases():
    good = compress(b"abcdabcdabcdabcd" * 8 + b"tail")
    return {
        "empty": b"",
        "xerial magic and nothing more": b"\x82SNAPPY\x00" + bytes(5),
        "xerial header and an empty block": b"\x82SNAPPY\x00" + bytes(12),
        "length varint without an end": b"\xff\xff\xff",
        "length beyond u32": b"\xff\xff\xff\xff\x7f",
        "length varint of eleven bytes": b"\x80" * 10 + b"\x01",
        "announced longer than the data": uvarint(len(good) + 50) + good[1:],
        "announced shorter than the data": uvarint(10) + good[1:],
        "literal past the input": uvarint(10) + bytes([9 << 2]) + b"abc",
        "long literal without its length bytes": uvarint(100) + bytes([61 << 2, 99]),
        "copy before anything is written": uvarint(4) + bytes([1 | 0 << 2, 1]),
        "copy with offset zero": uvarint(8) + literal(b"abcd") + bytes([2 | 3 << 2, 0, 0]),
        "copy from before the start": uvarint(8) + literal(b"abcd") + bytes([2 | 3 << 2, 5, 0]),
        "copy past the announced length": uvarint(6) + literal(b"abcd") + bytes([2 | 3 << 2, 4, 0]),
        "copy cut short": uvarint(8) + literal(b"abcd") + bytes([3 | 3 << 2, 4, 0]),
        "trailing byte": good + b"\x00",
    }
