"""LZ4 frames for the tests: a small compressor (greedy, hash of four bytes; blocks that keep the format's end
rules: the last five bytes literals, no match within the last twelve) with every option of the frame
descriptor, and frames that must not inflate."""
import struct

import oracle

MAGIC = struct.pack("<I", 0x184D2204)


def _length(n: int) -> bytes:
    """The bytes that follow a nibble of 15: n - 15 in steps of 255."""
    out = bytearray()
    n -= 15
    while n >= 255:
        out.append(255)
        n -= 255
    out.append(n)
    return bytes(out)


def sequence(literals: bytes, offset: int = 0, match: int = 0) -> bytes:
    ll, ml = len(literals), match - 4 if match else 0
    token = (min(ll, 15) << 4) | (min(ml, 15) if match else 0)
    out = bytearray([token])
    if ll >= 15:
        out += _length(ll)
    out += literals
    if match:
        out += struct.pack("<H", offset)
        if ml >= 15:
            out += _length(ml)
    return bytes(out)


def block(data: bytes, prefix: bytes = b"") -> bytes:
    """One compressed block of `data`; matches may reach into `prefix` (a linked frame's earlier output)."""
    buf = prefix[-65535:] + data
    base = len(buf) - len(data)
    table = {}
    for i in range(max(0, base - 65535), max(0, base - 3)):
        table[buf[i:i + 4]] = i
    out = bytearray()
    i, lit, n = base, base, len(buf)
    while i + 12 < n:                       # a match starts twelve bytes and more before the end
        key = buf[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and 0 < i - j <= 65535:
            m = 4
            while i + m < n - 5 and buf[j + m] == buf[i + m] and m < 70000:
                m += 1
            out += sequence(buf[lit:i], i - j, m)
            i += m
            lit = i
        else:
            i += 1
    out += sequence(buf[lit:n])
    return bytes(out)


def frame(data: bytes, size_id: int = 4, linked: bool = False, content_size: bool = False, content_sum: bool = False,
          block_sums: bool = False, store: bool = False, block_bytes: int = 0) -> bytes:
    flg = 0x40 | (0 if linked else 0x20) | (0x10 if block_sums else 0) | (8 if content_size else 0) | (4 if content_sum else 0)
    desc = bytes([flg, size_id << 4]) + (struct.pack("<Q", len(data)) if content_size else b"")
    out = bytearray(MAGIC + desc + bytes([(oracle.xxh32(desc) >> 8) & 0xFF]))
    limit = 1 << (8 + 2 * size_id)
    step = block_bytes or limit
    for at in range(0, len(data), step):
        piece = data[at:at + step]
        body = None if store else block(piece, data[:at] if linked else b"")
        if body is None or len(body) >= len(piece):
            body, word = piece, len(piece) | 0x80000000
        else:
            word = len(body)
        out += struct.pack("<I", word) + body
        if block_sums:
            out += struct.pack("<I", oracle.xxh32(body))
    out += struct.pack("<I", 0)
    if content_sum:
        out += struct.pack("<I", oracle.xxh32(data))
    return bytes(out)


def corrupt_cases():
    text = b"abcdabcdabcdabcdabcd" * 20 + b"the tail of it"
    good = frame(text)
    head = 7
    def with_desc(flg, bd, extra=b""):
        desc = bytes([flg, bd]) + extra
        return MAGIC + desc + bytes([(oracle.xxh32(desc) >> 8) & 0xFF]) + good[head:]
    body = good[head + 4:-4]
    def blocks(*bodies):
        out = bytearray(good[:head])
        for b in bodies:
            out += struct.pack("<I", len(b)) + b
        return bytes(out + struct.pack("<I", 0))
    return {
        "empty": b"",
        "magic only": MAGIC,
        "another magic": struct.pack("<I", 0x184C2102) + good[4:],
        "a skippable frame first": struct.pack("<II", 0x184D2A50, 0) + good,
        "version 0": with_desc(0x20, 0x40),
        "reserved bit of FLG": with_desc(0x62, 0x40),
        "reserved bits of BD": with_desc(0x60, 0x41),
        "block size id 3": with_desc(0x60, 0x30),
        "a dictionary id": with_desc(0x61, 0x40, struct.pack("<I", 7)),
        "header checksum": good[:6] + bytes([good[6] ^ 1]) + good[7:],
        "no end mark": good[:-4],
        "block longer than the data": good[:head] + struct.pack("<I", len(body) + 40) + body,
        "block beyond the maximal size": good[:head] + struct.pack("<I", 70000) + bytes(70000) + struct.pack("<I", 0),
        "content size that is not met": frame(text, content_size=True)[:6] + struct.pack("<Q", len(text) + 1) + b"\0" + good[head:],
        "content checksum": frame(text, content_sum=True)[:-1] + b"\x7f",
        "block checksum": (lambda f: f[:-5] + bytes([f[-5] ^ 1]) + f[-4:])(frame(text, block_sums=True)),
        "offset zero": blocks(sequence(b"abcdefgh", 0, 8) + sequence(b"12345678")),
        "offset before the start": blocks(sequence(b"abcdefgh", 9, 8) + sequence(b"12345678")),
        "offset into the block before, blocks independent": blocks(sequence(b"abcdefghijklmnop"), sequence(b"abcd", 12, 8) + sequence(b"12345678")),
        "literals past the block": blocks(bytes([0xF0, 200]) + b"abc"),
        "match with no room for the last literals": blocks(sequence(b"abcdefgh", 4, 8) + b"\x10a"),
        "ends with a match": blocks(sequence(b"abcdefghabcdefgh", 8, 8)),
        "trailing literals do not end the block": blocks(sequence(b"abcd") + b"zz"),
        "match length bytes at the very end": blocks(sequence(b"abcdefgh", 4, 19 + 255)[:-1] + sequence(b"123")),
    }
