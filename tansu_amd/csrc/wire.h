// Device-side readers for the Kafka record wire format: byte windows over
// global memory, zigzag varints and the per-record field walk.  Semantics follow
// nisshi-sans-io (paths in the reference tree):
//   VarInt::decode / LongVarInt::decode   src/primitive/varint.rs:49-78, 266-295
//   Octets::decode                         src/record/codec.rs:65-76
//   VarIntSequence<Header>::decode         src/record/codec.rs:283-302
//   Header::decode                         src/record/header.rs:55-65
//   Record::decode                         src/record.rs:254-277
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf2_crc32c.h"
#include "nisshi_hip.h"

namespace nisshi {

// Eight bytes starting at data[pos], assembled from aligned dword loads so that
// no load touches a 4-byte word that holds no byte of [data, data + len): an
// aligned word that contains a valid byte lies in that byte's page.  Bytes past
// `len` read as whatever follows or zero and must be ignored by the caller.
__device__ __forceinline__ uint64_t window64(const uint8_t *data, uint32_t pos, uint32_t len) {
    uintptr_t p = reinterpret_cast<uintptr_t>(data) + pos;
    uintptr_t last = (reinterpret_cast<uintptr_t>(data) + len - 1) & ~uintptr_t(3);
    uintptr_t a0 = p & ~uintptr_t(3);
    a0 = a0 <= last ? a0 : last;
    uintptr_t a1 = a0 + 4 <= last ? a0 + 4 : last;
    uintptr_t a2 = a0 + 8 <= last ? a0 + 8 : last;
    typedef __attribute__((address_space(1))) const uint32_t gbl_u32;  // global, not flat
    uint32_t d0 = *(gbl_u32 *)a0;
    uint32_t d1 = *(gbl_u32 *)a1;
    uint32_t d2 = *(gbl_u32 *)a2;
    uint32_t sh = (uint32_t)(p & 3);
    uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return ((uint64_t)hi << 32) | lo;
}

// VarInt::decode on a byte window.  `avail` = bytes left in the buffer at pos.
// At most five bytes; the fifth byte's bits above 32 are dropped (checked_shl
// only rejects a shift >= 32); a sixth byte is Overflow, unless the buffer ends
// first (the byte is fetched before the shift is checked), then TryGet.
__device__ __forceinline__ int varint32(uint64_t w, uint32_t avail, uint32_t &used, int32_t &out) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if ((uint32_t)i >= avail) return NISSHI_TRY_GET;
        uint32_t b = (uint32_t)(w >> (8 * i)) & 0xffu;
        acc |= (b & 0x7fu) << (7 * i);
        if (!(b & 0x80u)) {
            used = i + 1;
            out = (int32_t)(acc >> 1) ^ -(int32_t)(acc & 1u);
            return NISSHI_OK;
        }
    }
    return avail <= 5 ? NISSHI_TRY_GET : NISSHI_OVERFLOW;
}

// Branch-free form of VarInt::decode for the common case: a terminating byte
// within the first five.  Returns false otherwise (a sixth byte, or a buffer
// that ends inside the varint, which the caller detects through its own bounds);
// the caller then leaves the batch to the sequential walk, which reproduces the
// reference's outcome exactly.  As in the reference, the fifth byte's bits above
// bit 31 are dropped.
__device__ __forceinline__ bool fast_varint(uint64_t w, uint32_t &used, int32_t &out) {
    uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
    uint32_t stop = ~lo & 0x80808080u;  // bit 7 of every byte that ends a varint
    uint32_t idx = stop ? (uint32_t)__builtin_ctz(stop) >> 3 : 4u;
    uint32_t v = (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) | ((lo >> 2) & 0x1fc000u) |
                 ((lo >> 3) & 0xfe00000u) | (hi << 28);
    v &= idx == 4 ? 0xffffffffu : 0xffffffffu >> (25 - 7 * idx);
    used = idx + 1;
    out = (int32_t)(v >> 1) ^ -(int32_t)(v & 1u);
    return stop != 0 || !(hi & 0x80u);
}

// fast_varint that also notes whether the varint was longer than its value needs (its last byte
// carries no bits: what the encoder would write is shorter).  A fifth byte with only bits that
// are dropped counts as such too.  `slack` ends up zero if any varint noted into it was: the
// least, over them, of the last byte's bits, a varint of one byte counting as one.  (Kept in a
// vector register on purpose: a bool a varint would be scalar instructions, of which the
// chained walk has none to spare.)
__device__ __forceinline__ bool fast_varint_noting(uint64_t w, uint32_t &used, int32_t &out, uint32_t &slack) {
    uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
    uint32_t stop = ~lo & 0x80808080u;
    uint32_t idx = stop ? (uint32_t)__builtin_ctz(stop) >> 3 : 4u;
    uint32_t v = (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) | ((lo >> 2) & 0x1fc000u) |
                 ((lo >> 3) & 0xfe00000u) | (hi << 28);
    v &= idx == 4 ? 0xffffffffu : 0xffffffffu >> (25 - 7 * idx);
    uint32_t last = (v >> (7 * idx)) | ((idx - 1u) >> 31);
    slack = last < slack ? last : slack;
    used = idx + 1;
    out = (int32_t)(v >> 1) ^ -(int32_t)(v & 1u);
    return stop != 0 || !(hi & 0x80u);
}

// The same for LongVarInt::decode when it ends within four bytes (|value| < 2^27).
__device__ __forceinline__ bool fast_varlong(uint32_t lo, uint32_t &used, int64_t &out) {
    uint32_t stop = ~lo & 0x80808080u;
    uint32_t idx = (uint32_t)__builtin_ctz(stop | 0x80000000u) >> 3;
    uint32_t v = (lo & 0x7fu) | ((lo >> 1) & 0x3f80u) | ((lo >> 2) & 0x1fc000u) |
                 ((lo >> 3) & 0xfe00000u);
    v &= 0xffffffffu >> (25 - 7 * idx);
    used = idx + 1;
    out = (int64_t)((int32_t)(v >> 1) ^ -(int32_t)(v & 1u));
    return stop != 0;
}

// Reads eight bytes at any byte position of a batch's record data.
//
// When the wave has staged a stretch of the batch in LDS (BatchStream), positions
// inside [win_lo, win_hi) are read from there with one unaligned ds_read_b64 (the
// ring has a guard, so a read may run past its end); anything else comes from
// global memory through aligned dword loads.  `ring_bias` is the distance of the
// record data from the ring's grid origin.
struct Reader {
    const uint8_t *data;
    uint32_t len;
    // The 24 bytes last loaded, from a 4-byte boundary on: the fields of a record's
    // head are a few bytes apart, and each would otherwise cost a memory round
    // trip of its own.  Every lane keeps its own window.
    uint64_t q0, q1, q2;
    uintptr_t win;  // address of the window's first byte; 0: none

    __device__ __forceinline__ Reader(const uint8_t *data_, uint32_t len_)
        : data(data_), len(len_), q0(0), q1(0), q2(0), win(0) {}

    // Six aligned words from the one that holds `p`.  As in window64, no load
    // touches a word that holds no byte of the data.
    __device__ __forceinline__ void refill(uintptr_t p) {
        uintptr_t last = (reinterpret_cast<uintptr_t>(data) + len - 1) & ~uintptr_t(3);
        uintptr_t a0 = p & ~uintptr_t(3);
        a0 = a0 <= last ? a0 : last;
        typedef __attribute__((address_space(1))) const uint32_t gbl_u32;  // global, not flat
        uint32_t d[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            uintptr_t a = a0 + 4 * i <= last ? a0 + 4 * i : last;
            d[i] = *(gbl_u32 *)a;
        }
        q0 = ((uint64_t)d[1] << 32) | d[0];
        q1 = ((uint64_t)d[3] << 32) | d[2];
        q2 = ((uint64_t)d[5] << 32) | d[4];
        win = a0;
    }

    // eight bytes starting at pos (pos < len)
    __device__ __forceinline__ uint64_t get64(uint32_t pos) {
        uintptr_t p = reinterpret_cast<uintptr_t>(data) + pos;
        if (win == 0 || p < win || p + 8 > win + 24) refill(p);
        uint32_t o = (uint32_t)(p - win);  // 0 .. 16
        uint64_t lo = o < 8 ? q0 : o < 16 ? q1 : q2, hi = o < 8 ? q1 : q2;  // o == 16: the last eight, no shift
        uint32_t sh = (o & 7) * 8;
        return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    }

    __device__ __forceinline__ int varint(uint32_t &pos, int32_t &out) {
        if (pos >= len) return NISSHI_TRY_GET;
        uint32_t used = 0;
        int st = varint32(get64(pos), len - pos, used, out);
        pos += used;
        return st;
    }

    __device__ __forceinline__ int varlong(uint32_t &pos, int64_t &out) {
        if (pos >= len) return NISSHI_TRY_GET;
        uint32_t avail = len - pos;
        uint64_t acc = 0;
        uint64_t w = get64(pos);
        for (int i = 0; i < 11; i++) {
            if ((uint32_t)i >= avail) return NISSHI_TRY_GET;
            if (i == 10) return NISSHI_OVERFLOW;
            if (i == 8) w = get64(pos + 8);
            uint32_t b = (uint32_t)(w >> (8 * (i & 7))) & 0xffu;
            acc |= (uint64_t)(b & 0x7fu) << (7 * i);
            if (!(b & 0x80u)) {
                pos += i + 1;
                out = (int64_t)(acc >> 1) ^ -(int64_t)(acc & 1u);
                return NISSHI_OK;
            }
        }
        return NISSHI_OVERFLOW;
    }

    __device__ __forceinline__ int octets(uint32_t &pos, uint32_t &off, int32_t &n) {
        int st = varint(pos, n);
        if (st) return st;
        off = pos;
        if (n == -1) return NISSHI_OK;
        if (n < 0 || (uint32_t)n > len - pos) return NISSHI_MALFORMED;
        pos += (uint32_t)n;
        return NISSHI_OK;
    }
};

struct RecordFields {
    int32_t length;
    uint32_t attributes;
    int64_t timestamp_delta;
    int32_t offset_delta;
    uint32_t key_off, val_off;
    int32_t key_len, val_len;
    int32_t hdr_count;
    uint32_t hdr_pos;  // position of the first header, after the count
};

// Record::decode up to and including the header count; pos ends on the first header.
__device__ __forceinline__ int read_record_head(Reader &rd, uint32_t &pos, RecordFields &r) {
    int st;
    if ((st = rd.varint(pos, r.length))) return st;
    if (pos >= rd.len) return NISSHI_MALFORMED;  // get_u8 on an empty buffer panics
    r.attributes = (uint32_t)rd.get64(pos) & 0xffu;
    pos += 1;
    if ((st = rd.varlong(pos, r.timestamp_delta))) return st;
    if ((st = rd.varint(pos, r.offset_delta))) return st;
    if ((st = rd.octets(pos, r.key_off, r.key_len))) return st;
    if ((st = rd.octets(pos, r.val_off, r.val_len))) return st;
    if ((st = rd.varint(pos, r.hdr_count))) return st;
    if (r.hdr_count < 0) return NISSHI_MALFORMED;  // Vec::with_capacity(huge) panics
    r.hdr_pos = pos;
    return NISSHI_OK;
}

__device__ __forceinline__ uint32_t be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
__device__ __forceinline__ uint64_t be64(const uint8_t *p) {
    return ((uint64_t)be32(p) << 32) | be32(p + 4);
}
__device__ __forceinline__ uint16_t be16(const uint8_t *p) {
    return (uint16_t)(((uint32_t)p[0] << 8) | p[1]);
}

}  // namespace nisshi
