// CRC-32C arithmetic for gfx950 without lookup tables in the streaming loop.
//
// CRC-32C ("Crc32Iscsi" in the reference, nisshi-sans-io/src/record/deflated.rs:158-161)
// is the remainder of the message polynomial modulo
//   P(x) = x^32 + x^28 + x^27 + x^26 + x^25 + x^23 + x^22 + x^20 + x^19 + x^18
//        + x^14 + x^13 + x^11 + x^10 + x^9 + x^8 + x^6 + 1
// with bit-reflected I/O, init and final xor 0xFFFFFFFF.
//
// Representation used throughout: a 32-bit residue has bit i = coefficient of
// x^(31-i) (the reflected register), and message bytes loaded as little-endian
// words put stream bit s at word bit s, so a LOWER bit index is a HIGHER degree
// and multiplying by x^k is a logical right shift by k.
//
// The streaming loop keeps a 8192-bit accumulator row per wavefront (64 lanes x
// 16 bytes, exactly one coalesced dwordx4 load) and folds it onto the next row
// with a sparse congruence found by exhaustive search
// (tools/crc32c_sparse_search.c):
//   x^8192 == x^62 + x^59 + x^48 + x^43 + x^32 + x^27 + x^21 + x^5 + 1  (mod P)
// so one fold is nine shifted copies of the row XORed together: funnel shifts
// (v_alignbit_b32) and v_xor3, about 3.4 vector ops per byte and no LDS traffic,
// where a table-driven CRC needs one LDS lookup per byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nisshi {

constexpr uint32_t kPolyReflected = 0x82F63B78u;
constexpr uint32_t kOne = 0x80000000u;  // x^0
constexpr uint32_t kX = 0x40000000u;    // x^1
constexpr int kRowBytes = 1024;         // one accumulator row: 64 lanes x 16 bytes
constexpr int kChunkRows = 64;          // rows per CRC work item
constexpr int kChunkBytes = kRowBytes * kChunkRows;

__host__ __device__ inline uint32_t gf_mulx(uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // three instructions: the low bit spread over the word, the shift, a ^ (spread & P)
    uint32_t low = (uint32_t)__builtin_amdgcn_sbfe((int)a, 0, 1);
    return __builtin_amdgcn_bitop3_b32(a >> 1, low, kPolyReflected, 0x78);
#else
    return (a >> 1) ^ (kPolyReflected & (0u - (a & 1u)));
#endif
}

// a * b mod P
__host__ __device__ inline uint32_t gf_mulmod(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
        r = __builtin_amdgcn_bitop3_b32(r, a, (uint32_t)((int32_t)b >> 31), 0x78);  // r ^ (a & top bit of b, spread)
#else
        r ^= a & (0u - (b >> 31));
#endif
        b <<= 1;
        a = gf_mulx(a);
    }
    return r;
}

// v * x^32 mod P: one word step of the bitwise CRC
__host__ __device__ inline uint32_t gf_shift32(uint32_t v) {
#pragma unroll 8
    for (int i = 0; i < 32; i++) v = gf_mulx(v);
    return v;
}

// Constants in device memory, built once per context on the host.
struct CrcTables {
    uint32_t inv8[kRowBytes];   // x^(-8k): undoes k trailing zero bytes
    uint32_t lane_weight[64];   // x^(128 * (63 - lane)): position of a lane's 16 bytes in the row
    uint32_t pow8[40];          // x^(8 * 2^k)
    uint32_t chunk_weight;      // x^(8 * kChunkBytes) = W
    uint32_t chunk_pow[7];      // W^(2^k), k = 0 .. 6
    // For the residue of an accumulator row (crc_row_residue_staged), 16-byte aligned:
    // slice[k][v] = v * x^(8 * (k + 1)) for a byte v in the low bits of a word (the
    // byte-wise CRC tables, "slicing by four": (b0 | b1 << 8 | b2 << 16 | b3 << 24) * x^32
    // = slice[3][b0] ^ slice[2][b1] ^ slice[1][b2] ^ slice[0][b3]), and
    // lane_pad_weight[p][l] = lane_weight[l] * inv8[p], a lane's place in the row with p
    // bytes of padding at the row's end divided out.
    alignas(16) uint32_t slice[4][256];
    uint32_t chunk_wpow[64];      // W^k, k = 0 .. 63
    uint32_t xpow8_byte[4][256];  // xpow8_byte[k][v] = x^(8 * (v << 8k)): x^(8n) is the product over the bytes of n
    uint32_t lane_pad_weight[kRowBytes][64];
};
static_assert(offsetof(CrcTables, slice) % 16 == 0, "the slice tables are loaded sixteen bytes a lane");

inline uint32_t host_xpow(uint64_t n) {  // x^n mod P
    uint32_t r = kOne, base = kX;
    while (n) {
        if (n & 1) r = gf_mulmod(r, base);
        base = gf_mulmod(base, base);
        n >>= 1;
    }
    return r;
}

inline void build_crc_tables(CrcTables *t) {
    // P = x * Q + 1, so x^-1 = Q: drop the constant term of P and divide by x.
    uint32_t xinv = ((kOne ^ kPolyReflected) << 1) | 1u;
    uint32_t xinv8 = kOne;
    for (int i = 0; i < 8; i++) xinv8 = gf_mulmod(xinv8, xinv);
    t->inv8[0] = kOne;
    for (int k = 1; k < kRowBytes; k++) t->inv8[k] = gf_mulmod(t->inv8[k - 1], xinv8);
    for (int l = 0; l < 64; l++) t->lane_weight[l] = host_xpow(128ull * (63 - l));
    uint32_t p = host_xpow(8);
    for (int k = 0; k < 40; k++) {
        t->pow8[k] = p;
        p = gf_mulmod(p, p);
    }
    for (uint32_t v = 0; v < 256; v++) {
        uint32_t r = v;
        for (int k = 0; k < 4; k++) {
            for (int i = 0; i < 8; i++) r = gf_mulx(r);
            t->slice[k][v] = r;
        }
    }
    for (int p = 0; p < kRowBytes; p++)
        for (int l = 0; l < 64; l++) t->lane_pad_weight[p][l] = gf_mulmod(t->lane_weight[l], t->inv8[p]);
    for (int k = 0; k < 4; k++) {
        // x^(8 * v * 256^k) = (x^(8 * 256^k))^v, by repeated multiplication
        uint32_t step = host_xpow(8ull << (8 * k)), r = kOne;
        for (uint32_t v = 0; v < 256; v++) {
            t->xpow8_byte[k][v] = r;
            r = gf_mulmod(r, step);
        }
    }
    t->chunk_weight = host_xpow(8ull * kChunkBytes);
    t->chunk_wpow[0] = kOne;
    for (int k = 1; k < 64; k++) t->chunk_wpow[k] = gf_mulmod(t->chunk_wpow[k - 1], t->chunk_weight);
    t->chunk_pow[0] = t->chunk_weight;
    for (int k = 1; k < 7; k++) t->chunk_pow[k] = gf_mulmod(t->chunk_pow[k - 1], t->chunk_pow[k - 1]);
}

#if defined(__HIPCC__)

// x^(8n) mod P from the table of x^(8 * 2^k)
__device__ inline uint32_t gf_xpow8(const CrcTables *t, uint64_t n) {
    // a factor a byte of n from the tables (three multiplications below 4 GiB); what
    // lies above bit 31 bit by bit
    uint32_t lo = (uint32_t)n;
    uint32_t r = t->xpow8_byte[0][lo & 0xffu];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        uint32_t v = (lo >> (8 * k)) & 0xffu;
        if (v) r = gf_mulmod(r, t->xpow8_byte[k][v]);
    }
    n >>= 32;
    for (int k = 32; n; k++, n >>= 1)
        if (n & 1) r = gf_mulmod(r, t->pow8[k]);
    return r;
}

// Tell the compiler a value is the same in every lane, so that it lives in a
// scalar register and branches on it are scalar branches.
__device__ __forceinline__ uint32_t uni(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}

__device__ __forceinline__ uint32_t fsh(uint32_t hi, uint32_t lo, int r) {
    return __builtin_amdgcn_alignbit(hi, lo, r);  // ({hi,lo} >> r)[31:0]
}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // v_bitop3_b32, truth table of a ^ b ^ c
}

// The value of the next lane up, zero in lane 63 (v_mov_b32_dpp wave_shl:1, bound_ctrl:0).
__device__ __forceinline__ uint32_t lane_above(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

// One fold: acc = next ^ (acc * x^8192 mod Q), Q = x^8192 + T.  The row is a
// little-endian 8192-bit number spread over the wave; multiplying by x^t is a
// right shift of the whole row by t bits.  The t low bits that fall off the
// bottom belong to the top of the row and are folded once more; for t < 64 they
// all come from lane 0's first two words and land in lane 63's last two.
__device__ __forceinline__ void crc_fold_row(uint32_t (&a)[4], const uint32_t (&nx)[4],
                                             int lane) {
    uint64_t low = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(a[1]) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane(a[0]);
    uint64_t wrap = (low << 2) ^ (low << 5) ^ (low << 16) ^ (low << 21) ^ (low << 32) ^
                    (low << 37) ^ (low << 43) ^ (low << 59);  // 64 - t for every t > 0
    bool top = lane == 63;
    uint32_t w0 = a[0], w1 = a[1];
    uint32_t w2 = a[2] ^ (top ? (uint32_t)wrap : 0u);
    uint32_t w3 = a[3] ^ (top ? (uint32_t)(wrap >> 32) : 0u);
    uint32_t w4 = lane_above(w0), w5 = lane_above(w1);
    // shifted copies of each adjacent pair; the x^27 and x^59 terms share theirs
    uint32_t p0_5 = fsh(w1, w0, 5), p0_21 = fsh(w1, w0, 21), p0_27 = fsh(w1, w0, 27);
    uint32_t p1_5 = fsh(w2, w1, 5), p1_21 = fsh(w2, w1, 21), p1_27 = fsh(w2, w1, 27);
    uint32_t p1_11 = fsh(w2, w1, 11), p1_16 = fsh(w2, w1, 16), p1_30 = fsh(w2, w1, 30);
    uint32_t p2_5 = fsh(w3, w2, 5), p2_21 = fsh(w3, w2, 21), p2_27 = fsh(w3, w2, 27);
    uint32_t p2_11 = fsh(w3, w2, 11), p2_16 = fsh(w3, w2, 16), p2_30 = fsh(w3, w2, 30);
    uint32_t p3_5 = fsh(w4, w3, 5), p3_21 = fsh(w4, w3, 21), p3_27 = fsh(w4, w3, 27);
    uint32_t p3_11 = fsh(w4, w3, 11), p3_16 = fsh(w4, w3, 16), p3_30 = fsh(w4, w3, 30);
    uint32_t p4_11 = fsh(w5, w4, 11), p4_16 = fsh(w5, w4, 16), p4_27 = fsh(w5, w4, 27);
    uint32_t p4_30 = fsh(w5, w4, 30);
    // word i: x^0 w[i], x^32 w[i+1], x^5 x^21 x^27 of pair i, x^43 x^48 x^59 x^62 of pair i+1
    a[0] = xor3(xor3(nx[0], w0, w1), xor3(p0_5, p0_21, p0_27), xor3(p1_11, p1_16, p1_27)) ^ p1_30;
    a[1] = xor3(xor3(nx[1], w1, w2), xor3(p1_5, p1_21, p1_27), xor3(p2_11, p2_16, p2_27)) ^ p2_30;
    a[2] = xor3(xor3(nx[2], w2, w3), xor3(p2_5, p2_21, p2_27), xor3(p3_11, p3_16, p3_27)) ^ p3_30;
    a[3] = xor3(xor3(nx[3], w3, w4), xor3(p3_5, p3_21, p3_27), xor3(p4_11, p4_16, p4_27)) ^ p4_30;
}

__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d, 64);
    return v;
}

// Collapse the accumulator row to the 32-bit value (row polynomial) * x^32 mod P.
__device__ inline uint32_t crc_row_residue(const uint32_t (&a)[4], int lane, const CrcTables *t) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) c = gf_shift32(c ^ a[i]);
    return wave_xor(gf_mulmod(c, t->lane_weight[lane]));
}

// pointers into LDS keep their address space, so that accesses are ds_read /
// ds_write and not flat instructions
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u128;

// crc_row_residue() with `pad` bytes of zero padding at the row's end divided out
// (pad < kRowBytes, the same in every lane), for a wave that has 4 KiB of LDS to
// spare (`tab`, 16-byte aligned; whatever it held is gone): a lane's four words are
// reduced through the byte-wise tables, staged in `tab` first, instead of bit by
// bit (128 dependent steps of three instructions), and the lane's weight comes
// with the padding already divided out (a multiplication less).
__device__ __forceinline__ void crc_stage_tables(lds_u32 *tab, const CrcTables *t, int lane) {
    typedef __attribute__((address_space(1))) const u32x4 gbl_u128;
    gbl_u128 *src = (gbl_u128 *)(uintptr_t)&t->slice[0][0];
    u32x4 q0 = src[lane], q1 = src[64 + lane], q2 = src[128 + lane], q3 = src[192 + lane];
    lds_u128 *dst = (lds_u128 *)tab;
    dst[lane] = q0;
    dst[64 + lane] = q1;
    dst[128 + lane] = q2;
    dst[192 + lane] = q3;
    asm volatile("" ::: "memory");
}

// The residue of an accumulator row through byte tables that crc_stage_tables has
// put in `tab`; `weight`: the lane's weight in the row (with any padding divided out).
__device__ __forceinline__ uint32_t crc_row_residue_tab(const uint32_t (&a)[4], uint32_t weight, const lds_u32 *tab) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c ^= a[i];
        c = xor3(tab[768 + (c & 0xffu)], tab[512 + ((c >> 8) & 0xffu)], tab[256 + ((c >> 16) & 0xffu)]) ^ tab[c >> 24];
    }
    return wave_xor(gf_mulmod(c, weight));
}

__device__ __forceinline__ uint32_t crc_row_residue_staged(const uint32_t (&a)[4], int lane, const CrcTables *t,
                                                           uint32_t pad, lds_u32 *tab) {
    typedef __attribute__((address_space(1))) const uint32_t gbl_u32;
    uint32_t weight = ((gbl_u32 *)(uintptr_t)&t->lane_pad_weight[0][0])[pad * 64 + (uint32_t)lane];
    crc_stage_tables(tab, t, lane);
    return crc_row_residue_tab(a, weight, tab);
}

// Byte mask of a little-endian word at absolute address `addr`: keep the bytes in [lo, hi).
__device__ __forceinline__ uint32_t byte_mask(uint64_t addr, uint64_t lo, uint64_t hi) {
    uint32_t m = 0xFFFFFFFFu;
    if (lo > addr) m = lo >= addr + 4 ? 0u : m << (8 * (uint32_t)(lo - addr));
    if (hi < addr + 4) m = hi <= addr ? 0u : m & (0xFFFFFFFFu >> (8 * (uint32_t)(addr + 4 - hi)));
    return m;
}

// Raw remainder of the bytes [lo, hi) of `base` that fall in rows [row0, row1) of
// the row grid anchored at `grid` (16-byte aligned absolute offset, grid <= lo).
// `first4` is the offset of the first message byte: the four bytes from there
// are complemented, which is the CRC's 0xFFFFFFFF initial value.  The result is
// (bytes as a polynomial, zero padded to the end of the last row) * x^32 mod P.
__device__ inline uint32_t crc_rows(const uint8_t *base, uint64_t grid, uint64_t lo, uint64_t hi,
                                    uint64_t first4, uint32_t row0, uint32_t row1, int lane,
                                    const CrcTables *t) {
    uint32_t acc[4] = {0, 0, 0, 0};  // folding a zero accumulator gives the row itself
    // rows that lie wholly inside the data and past its first four bytes need no
    // masks: eight of them are loaded before the first is folded, so that a wave
    // keeps 8 KiB in flight
    constexpr uint32_t kBurst = 8;
    uint32_t full_rows = hi > grid ? (uint32_t)((hi - grid) / kRowBytes) : 0;
    uint32_t plain_from = first4 < grid + kRowBytes ? 1 : 0, plain_to = row1 < full_rows ? row1 : full_rows;
    const uint8_t *lane_base = base + grid + (uint64_t)lane * 16;
    for (uint32_t r = row0; r < row1;) {
        if (r >= plain_from && r + kBurst <= plain_to) {
            u32x4 q[kBurst];
#pragma unroll
            for (uint32_t j = 0; j < kBurst; j++)
                q[j] = *reinterpret_cast<const u32x4 *>(lane_base + (uint64_t)(r + j) * kRowBytes);
#pragma unroll
            for (uint32_t j = 0; j < kBurst; j++) {
                uint32_t nx[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
                crc_fold_row(acc, nx, lane);
            }
            r += kBurst;
            continue;
        }
        uint64_t addr = grid + (uint64_t)r * kRowBytes + (uint64_t)lane * 16;
        uint32_t nx[4] = {0, 0, 0, 0};
        if (addr + 16 > lo && addr < hi) {
            uint4 v = *reinterpret_cast<const uint4 *>(base + addr);
            nx[0] = v.x;
            nx[1] = v.y;
            nx[2] = v.z;
            nx[3] = v.w;
            if (addr < lo || addr + 16 > hi) {
#pragma unroll
                for (int i = 0; i < 4; i++) nx[i] &= byte_mask(addr + 4 * i, lo, hi);
            }
        }
        // the complement covers four bytes even when the message is shorter
        if (addr + 16 > first4 && addr < first4 + 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) nx[i] ^= byte_mask(addr + 4 * i, first4, first4 + 4);
        }
        crc_fold_row(acc, nx, lane);
        r++;
    }
    return crc_row_residue(acc, lane, t);
}


// A wave's pass over one batch: rows are fetched in order with coalesced 16-byte
// loads, folded into the CRC accumulator and copied raw into a ring of the last
// ring rows in LDS, from which the same wave then parses records.  The
// batch is read from HBM once.  The caller loads the rows it will need next into
// registers (load), works on what is in the ring while they arrive, and then
// moves them in (take_loaded): both within one pass of its loop, so that no load
// is in flight across a loop edge.
// The ring comes in two sizes: eight rows let four waves a SIMD run, which suits
// batches of a few KiB; twelve rows hold a round of 64 records of up to 160 bytes
// (eight rows hold 42 of C2's 143-byte records) at three waves a SIMD, which suits
// batches of 64 KiB and more.
constexpr int kRingGuard = 64;  // the ring's first bytes again, so a window may run past the end
template <int kRows>
struct RingCfg {
    static constexpr int kRingRows = kRows;
    static constexpr int kRingBytes = kRows * kRowBytes;
    static constexpr int kRingWords = (kRingBytes + kRingGuard) / 4;
    static constexpr int kAhead = kRows == 8 ? 8 : kRows - 1;  // rows asked for a round ahead
    static constexpr bool kPow2 = (kRows & (kRows - 1)) == 0;
    __device__ static __forceinline__ uint32_t slot_of(uint32_t row) {
        return kPow2 ? row & (uint32_t)(kRows - 1) : row % (uint32_t)kRows;
    }
    __device__ static __forceinline__ uint32_t wrap(uint32_t at) {
        return kPow2 ? at & (uint32_t)(kRingBytes - 1) : at % (uint32_t)kRingBytes;
    }
};

template <int kRows, bool kFold = true>
struct BatchStream {
    typedef RingCfg<kRows> Cfg;
    static constexpr int kRingBytes = Cfg::kRingBytes, kAhead = Cfg::kAhead, kRows_ = kRows;
    // Offsets are relative to `grid`, the 16-byte boundary at or below the first
    // byte: the stream handles at most 4 GiB.
    uint32_t lo, hi;          // the bytes [lo, hi) are the data; lo < 16
    uint32_t rows, next_row;
    uint32_t fold_to;         // rows from here on pass through the ring without entering the CRC (see segment())
    uint32_t acc[4];
    lds_u32 *ring;
    int lane;
    const uint8_t *lane_ptr;  // this lane's 16 bytes of row 0
    // kFold false: the rows only pass through the ring for the walk, no CRC is taken
    // (a batch whose CRC comes from its chunks, a second walk of a batch); rows
    // nobody reads may then be skipped
    static constexpr bool fold = kFold;

    __device__ __forceinline__ void init(const uint8_t *base, uint64_t lo_, uint64_t hi_,
                                         lds_u32 *ring_, int lane_) {
        uint64_t grid = uni64(lo_) & ~uint64_t(15);
        lo = (uint32_t)(uni64(lo_) - grid);
        hi = (uint32_t)(uni64(hi_) - grid);
        rows = (hi + kRowBytes - 1) / kRowBytes;
        next_row = 0;
        fold_to = 0xffffffffu;
        ring = ring_;
        lane = lane_;
        lane_ptr = base + grid + (uint64_t)lane * 16;
        acc[0] = acc[1] = acc[2] = acc[3] = 0;
    }

    // The stream of one piece of a batch whose CRC is taken piece by piece: it
    // starts at row `row0` with an empty accumulator (folding nothing gives the piece
    // itself) and folds the rows below `to`; rows from there on only pass through
    // the ring, for the record that straddles the piece's end.
    __device__ __forceinline__ void segment(uint32_t row0, uint32_t to) {
        next_row = row0;
        fold_to = to < rows ? to : 0xffffffffu;
    }

    // The rows only pass through the ring for the walk (the batch's CRC comes from
    // its chunks); rows nobody reads may then be skipped.
    __device__ __forceinline__ void pass_through() { fold_to = 0; }

    // first row that does not touch a byte below offset `x`
    __device__ __forceinline__ uint32_t row_after(uint32_t x) const {
        uint32_t r = (x + kRowBytes - 1) / kRowBytes;
        return r < rows ? r : rows;
    }

    __device__ __forceinline__ bool edge(uint32_t r) const { return r == 0 || r + 1 >= rows; }

    // Where this lane reads its 16 bytes of row r.  Every lane loads, whatever its
    // bytes hold: lanes of the last row that lie past the data (and rows past the
    // last) re-read the last 16 data bytes instead, and fix_edge() discards what
    // they got.
    __device__ __forceinline__ const uint8_t *row_ptr(uint32_t r) const {
        uint32_t off = r * kRowBytes + (uint32_t)lane * 16;
        uint32_t last = (hi - 1) & ~15u;
        off = off < last ? off : last;
        return lane_ptr + (int64_t)((int32_t)off - lane * 16);
    }

    __device__ __forceinline__ u32x4 fetch(uint32_t r) const {
        return *reinterpret_cast<const u32x4 *>(row_ptr(r));
    }

    // Row r into the ring (raw: bytes outside [lo, hi) are never parsed) and into
    // the CRC.  The accumulator starts at zero, and folding zero gives zero.
    __device__ __forceinline__ void take(u32x4 v, uint32_t r) {
        uint32_t slot = Cfg::slot_of(r);
        *(lds_u128 *)(ring + slot * (kRowBytes / 4) + lane * 4) = v;
        if (slot == 0 && lane < kRingGuard / 16) *(lds_u128 *)(ring + kRingBytes / 4 + lane * 4) = v;
        if (fold && r < fold_to) {
            uint32_t nx[4] = {v.x, v.y, v.z, v.w};
            crc_fold_row(acc, nx, lane);
        }
    }

    // Without a CRC to take, rows that lie before everything a round up to row
    // `target` reads need not be loaded: the ring holds kRows rows.
    __device__ __forceinline__ void skip_before(uint32_t target) {
        if ((!fold || fold_to == 0) && target > next_row + (uint32_t)kRows_) next_row = target - (uint32_t)kRows_;
    }

    // The first and the last row were folded as loaded: take out the bytes that
    // are not data, and complement the first four data bytes (the CRC's initial
    // value 0xFFFFFFFF).  The fold is linear, so this is an XOR into the
    // accumulator, before the next row is folded.
    __device__ __forceinline__ void fix_edge(uint32_t r) {
        if (!fold || fold_to == 0) return;
        uint32_t slot = Cfg::slot_of(r);
        u32x4 raw = *(lds_u128 *)(ring + slot * (kRowBytes / 4) + lane * 4);
        uint64_t addr = (uint64_t)r * kRowBytes + (uint64_t)lane * 16;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t fix = raw[i] & ~byte_mask(addr + 4 * i, lo, hi);
            if (r == 0) fix ^= byte_mask(addr + 4 * i, lo, lo + 4);
            acc[i] ^= fix;
        }
    }

    // Issue the loads of rows next_row .. next_row + n - 1 (n <= kAhead).  All the
    // addresses are formed before the first load goes out: the compiler waits for
    // every load it thinks outstanding before it lets address arithmetic reuse a
    // register that a load fills, and between the loads that wait would be a real one.
    __device__ __forceinline__ void load(u32x4 (&q)[kAhead], uint32_t n) const {
        typedef __attribute__((address_space(1))) const u32x4 gbl_u128;  // global, not flat
        gbl_u128 *p[kAhead];
        if (next_row + kAhead < rows) {  // none of them is the last row: no clamp
#pragma unroll
            for (int j = 0; j < kAhead; j++)
                p[j] = (gbl_u128 *)(uintptr_t)(lane_ptr + (uint64_t)(next_row + j) * kRowBytes);
        } else {
#pragma unroll
            for (int j = 0; j < kAhead; j++) p[j] = (gbl_u128 *)(uintptr_t)row_ptr(next_row + j);
        }
#pragma unroll
        for (int j = 0; j < kAhead; j++) asm volatile("" : "+v"(p[j]));
#pragma unroll
        for (int j = 0; j < kAhead; j++)
            if ((uint32_t)j < n) q[j] = *p[j];
    }

    // Move the first n of the rows loaded into the ring and the CRC.
    __device__ __forceinline__ void take_loaded(const u32x4 (&q)[kAhead], uint32_t n) {
#pragma unroll
        for (int j = 0; j < kAhead; j++) {
            if ((uint32_t)j < n) {
                take(q[j], next_row + j);
                if (j == 0 && next_row == 0) fix_edge(0);
            }
        }
        next_row += n;
        if (n && next_row == rows && rows > 1 && fold_to > rows) fix_edge(rows - 1);
    }

    // Rows up to (not including) `target`, for the start of a batch and for what the
    // walk did not ask for in advance: up to eight at a time, all of them loaded
    // before the first is taken, so that the start of a batch pays one memory
    // latency and not one a row.
    __device__ __forceinline__ void pull(uint32_t target) {
        target = uni(target);
        if (target > rows) target = rows;
        skip_before(target);
#pragma nounroll
        while (next_row < target) {
            uint32_t n = uni(target - next_row < (uint32_t)kAhead ? target - next_row : (uint32_t)kAhead);
            u32x4 q[kAhead];
            load(q, n);
            take_loaded(q, n);
        }
    }

    // pull() with the rows loaded kAhead at a time into the caller's queue, all
    // before the first is taken: the start of a batch then pays one memory latency
    // or two, not one a row.
    __device__ __forceinline__ void pull_wide(uint32_t target, u32x4 (&q)[kAhead]) {
        target = uni(target);
        if (target > rows) target = rows;
        skip_before(target);
#pragma nounroll
        while (next_row < target) {
            uint32_t n = uni(target - next_row < (uint32_t)kAhead ? target - next_row : (uint32_t)kAhead);
            load(q, n);
            take_loaded(q, n);
        }
    }

    // The residue of the rows [row0, to) of segment(): the piece's bytes, zero padded
    // to the end of its last row, times x^32 mod P (what crc_rows() returns); of the
    // batch's last piece with the padding divided out.
    __device__ __forceinline__ uint32_t finish_segment(uint32_t to, const CrcTables *t) {
        to = to < rows ? to : rows;
        if (next_row < to) pull(to);
        // (the rows are through the ring: its first 4 KiB hold the tables from here on)
        return crc_row_residue_staged(acc, lane, t, to == rows ? rows * kRowBytes - hi : 0u, ring);
    }

    // the CRC-32C of [lo, hi)
    __device__ __forceinline__ uint32_t finish(const CrcTables *t) {
        if (!fold) return 0;
        pull(rows);
        if (hi == lo) return 0;
        return ~crc_row_residue_staged(acc, lane, t, rows * kRowBytes - hi, ring);
    }
};

#endif  // __HIPCC__

}  // namespace nisshi
