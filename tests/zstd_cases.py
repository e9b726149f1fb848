"""zstd frames for the tests of the zstd inflate: from a real compressor (pyarrow's
bundled libzstd: one-shot frames at several levels, which between them use raw,
repeated and Huffman-coded literals in one and four streams, predefined, described,
repeated and one-symbol code tables, repeat offsets, and several blocks a frame),
variations the compressor does not produce put together by hand (a content
checksum, a frame without its content size, stored and repeated blocks, skippable
frames, several frames one after the other), and damaged ones."""
import struct

import numpy as np
import pyarrow as pa

import oracle

MAGIC = 0xFD2FB528


def compress(data: bytes, level: int = 3) -> bytes:
    return pa.Codec("zstd", compression_level=level).compress(bytes(data), asbytes=True)


def data_of(kind: str, n: int, rng) -> bytes:
    if kind == "text":
        words = [b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"consectetur", b"adipiscing", b"elit", b"sed", b"do"]
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, len(words)))] + b" "
        return bytes(out[:n])
    if kind == "random":
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    if kind == "zeros":
        return bytes(n)
    if kind == "skewed":      # few symbols, very unequal: long and short Huffman codes
        return bytes(rng.choice(np.array([65, 66, 67, 68, 69, 70, 200, 201], dtype=np.uint8), n,
                                p=[0.5, 0.25, 0.12, 0.06, 0.03, 0.02, 0.01, 0.01]))
    unit = bytes(rng.integers(0, 256, min(n, 40000), dtype=np.uint8))     # repeats 40 000 bytes apart
    return (unit * (n // max(len(unit), 1) + 1))[:n] if n else b""


KINDS = ["text", "random", "zeros", "skewed", "far"]
SIZES = (0, 1, 7, 100, 1000, 5000, 70000, 140000, 300000)
LEVELS = (1, 3, 9, 19)


def header_len(frame: bytes) -> int:
    """Bytes of the frame header of a frame the compressor made (no dictionary)."""
    fhd = frame[4]
    single, fcs = (fhd >> 5) & 1, fhd >> 6
    return 5 + (0 if single else 1) + ([single, 2, 4, 8][fcs])


def with_checksum(frame: bytes, content: bytes, good: bool = True) -> bytes:
    """The frame with the content checksum flag set and the checksum behind its last block."""
    assert struct.unpack_from("<I", frame, 0)[0] == MAGIC and not frame[4] & 4
    out = bytearray(frame)
    out[4] |= 4
    digest = oracle.xxh64(content) & 0xFFFFFFFF
    return bytes(out) + struct.pack("<I", digest if good else digest ^ 1)


def without_size(frame: bytes) -> bytes:
    """A one-shot frame (single segment, content size) as a streaming compressor writes it:
    a window descriptor in place of the content size."""
    h = header_len(frame)
    assert (frame[4] >> 5) & 1
    # window: the smallest of 1 KiB * 2^e that holds 128 KiB and the content is not needed: take 8 MiB
    return struct.pack("<IBB", MAGIC, 0, (13 << 3)) + frame[h:]


def stored_frame(data: bytes, block: int = 1000, rle_byte=None) -> bytes:
    """A frame of raw blocks (or, with rle_byte, of repeated-byte blocks), no content size."""
    out = bytearray(struct.pack("<IBB", MAGIC, 0, (10 << 3)))
    pieces = [data[i:i + block] for i in range(0, len(data), block)] or [b""]
    for i, piece in enumerate(pieces):
        last = 1 if i + 1 == len(pieces) else 0
        if rle_byte is None:
            out += struct.pack("<I", last | 0 << 1 | len(piece) << 3)[:3] + piece
        else:
            out += struct.pack("<I", last | 1 << 1 | len(piece) << 3)[:3] + bytes([rle_byte])
    return bytes(out)


def skippable(payload: bytes, nibble: int = 0) -> bytes:
    return struct.pack("<II", 0x184D2A50 | nibble, len(payload)) + payload


def corrupt_cases(rng):
    """Record data that must not inflate."""
    text = data_of("text", 20000, rng)
    good = compress(text, 3)
    h = header_len(good)
    cases = [("empty magic", b"\x00\x00\x00\x00"), ("short", good[:3]), ("cut in the header", good[:5]),
             ("cut in a block", good[:len(good) // 2]), ("cut by a byte", good[:-1]),
             ("a byte behind the frame", good + b"\x00"),
             ("reserved bit", good[:4] + bytes([good[4] | 8]) + good[5:]),
             ("dictionary id", struct.pack("<IBBB", MAGIC, 1, 10 << 3, 7) + good[h:]),
             ("window too large", struct.pack("<IBB", MAGIC, 0, (18 << 3)) + good[h:]),
             ("reserved block type", good[:h] + bytes([good[h] | 6]) + good[h + 1:]),
             ("wrong content size", good[:5] + struct.pack("<H", struct.unpack_from("<H", good, 5)[0] + 1) + good[7:]
              if (good[4] >> 6) == 1 else good[:h - 1] + bytes([good[h - 1] ^ 1]) + good[h:]),
             ("wrong checksum", with_checksum(good, text, good=False)),
             ("skippable frame cut", good + skippable(b"abcdef")[:-2])]
    for k in range(40):     # a byte changed inside the compressed block
        at = int(rng.integers(h + 3, len(good)))
        bad = bytearray(good)
        bad[at] ^= 1 << int(rng.integers(0, 8))
        cases.append((f"bit flipped at {at}", bytes(bad)))
    return cases
