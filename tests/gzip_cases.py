"""gzip members for the tests, from Python's zlib (a real DEFLATE compressor: stored, fixed and dynamic blocks),
with the header's optional fields, and members that must not inflate."""
import struct
import zlib


def member(data: bytes, level: int = 6, strategy: int = zlib.Z_DEFAULT_STRATEGY, extra: bytes = None, name: bytes = None,
           comment: bytes = None, chunk: int = 0) -> bytes:
    """One gzip member.  chunk: the data is compressed so many bytes at a time with a full flush between
    (several blocks, stored ones of no bytes among them)."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if chunk:
        raw = b"".join(co.compress(data[i:i + chunk]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), chunk)) + co.flush()
    else:
        raw = co.compress(data) + co.flush()
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0)
    head = bytes([0x1f, 0x8b, 8, flg, 0, 0, 0, 0, 0, 255])
    if extra is not None:
        head += struct.pack("<H", len(extra)) + extra
    if name is not None:
        head += name + b"\0"
    if comment is not None:
        head += comment + b"\0"
    return head + raw + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


def bits_to_bytes(bits: str) -> bytes:
    """A string of 0/1 in the order DEFLATE reads them, as bytes (lowest bit of a byte first)."""
    bits += "0" * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8][::-1], 2) for i in range(0, len(bits), 8))


def raw_member(deflate: bytes, size: int) -> bytes:
    return bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255]) + deflate + struct.pack("<II", 0, size)


def corrupt_cases():
    text = b"abcdabcdabcdabcdabcd" * 20 + b"the tail of it"
    good = member(text)
    fixed = member(text, strategy=zlib.Z_FIXED)
    stored = member(text, level=0)
    def lsb(v, n):
        return format(v, "0%db" % n)[::-1]
    return {
        "empty": b"",
        "magic only": b"\x1f\x8b",
        "another method": good[:2] + b"\x07" + good[3:],
        "a reserved flag": good[:3] + b"\x20" + good[4:],
        "a header checksum": good[:3] + b"\x02" + good[4:10] + b"\0\0" + good[10:],
        "a name without an end": good[:3] + b"\x08" + good[4:10] + b"name",
        "an extra field longer than the data": good[:3] + b"\x04" + good[4:10] + b"\xff\xff" + good[10:],
        "block type 3": raw_member(bits_to_bytes("1" + "11"), 0),
        "stored: the complement does not match": raw_member(bits_to_bytes("1" + "00") + b"\x03\x00\xfc\xfe" + b"abc", 3),
        "stored: longer than the data": raw_member(bits_to_bytes("1" + "00") + b"\x09\x00\xf6\xff" + b"abc", 9),
        "no last block": stored[:10] + bytes([stored[10] & 0xFE]) + stored[11:],
        "the length field one more": good[:-4] + struct.pack("<I", len(text) + 1),
        "the length field one less": good[:-4] + struct.pack("<I", len(text) - 1),
        "bytes behind the trailer": good + b"\0",
        "cut in the blocks": good[:len(good) // 2] + good[-8:],
        "fixed: a distance before the start": raw_member(bits_to_bytes("1" + "10" + "00110001" + "0000001" + "00001" + "0000000"), 4),
        "fixed: length symbol 286": raw_member(bits_to_bytes("1" + "10" + "00110001" + "11000110" + "00000" + "0000000"), 4),
        "fixed: distance symbol 30": raw_member(bits_to_bytes("1" + "10" + "00110001" + "0000001" + "11110" + "0000000"), 4),
        "dynamic: too many length codes": raw_member(bits_to_bytes("1" + "01" + lsb(30, 5) + lsb(0, 5) + lsb(0, 4)), 0),
        "dynamic: code of code lengths incomplete": raw_member(bits_to_bytes("1" + "01" + lsb(0, 5) + lsb(0, 5) + lsb(0, 4) + lsb(1, 3) + lsb(0, 3) * 3), 0),
        "dynamic: repeat with nothing before it": raw_member(bits_to_bytes("1" + "01" + lsb(0, 5) + lsb(0, 5) + lsb(0, 4) + lsb(1, 3) + lsb(1, 3) + lsb(0, 3) * 2 + "0" + "00"), 0),
        "a fixed block damaged": fixed[:14] + bytes([fixed[14] ^ 0x55]) + fixed[15:],
    }
