// MI355X (gfx950) Kafka RecordBatch v2 codec: kernels and the C ABI of
// include/nisshi_hip.h.  Reference behaviour: nisshi-sans-io/src/record.rs,
// src/record/{deflated,inflated,codec,header}.rs, src/primitive/varint.rs.
//
// Decode pipeline (all on one stream):
//   batch_header_kernel   one lane per batch: the 61-byte header, its checks, the record_data span
//   scan_batches_kernel   exclusive scans over batches: record slots, CRC work items
//   crc_chunks_kernel     batches of 512 KiB and more only: one wave per 64 KiB, CRC-32C by sparse folding
//   decode_fused_kernel   one wave per batch: CRC-32C streamed just ahead of the record walk;
//                         record starts by speculation, records matched against a learnt shape
//   slow_walk_kernel      what the streamed walk left: large batches and unusual records from
//                         global memory, the sequential walk for exact error outcomes, the totals
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "gf2_crc32c.h"
#include "nisshi_hip.h"
#include "wire.h"

namespace nisshi {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

// Internal per-batch flags
constexpr uint32_t kFlagExact = 1u;   // needs the sequential walk
constexpr uint32_t kFlagGlobal = 2u;  // needs the walk from global memory, from a resume point

struct Work {
    uint32_t *alloc_count;  // record slots the batch gets
    uint32_t *n_chunks;     // CRC work items of the batch
    uint32_t *chunk_base;   // exclusive scan of n_chunks, n + 1 entries
    uint32_t *partial;      // one residue per CRC work item
    uint32_t *flags;
    uint32_t *resume_k;     // kFlagGlobal: the first record not yet decoded
    uint32_t *resume_pos;   // and its offset in the record data
    uint64_t *totals;       // [0] record slots [1] header slots [2] failed [3] exact [4] out bytes [5] chunks
    uint64_t *tile_sum;     // the scans over very many batches: two sums a tile of kScanTile batches
    uint32_t *chunk_done;   // encode: the CRC work item has its residue already (from the copy of a long span)
};

// ------------------------------------------------------------ batch header

// deflated::Batch::try_from(Bytes), deflated.rs:108-213, in the reference's order
// of checks; then Compression::try_from (lib.rs:1685-1698).
__global__ void batch_header_kernel(const uint8_t *blob, const uint64_t *batch_off, uint32_t n,
                                    int by_value, nisshi_columns c, Work w) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    // the call's totals start at zero: their first users are the kernels after this one
    if (b < 8) w.totals[b] = 0;
    if (b >= n) return;
    const uint8_t *p = blob + batch_off[b];
    uint64_t len = batch_off[b + 1] - batch_off[b];
    int st = NISSHI_OK;
    int64_t base_offset = 0, base_timestamp = 0, max_timestamp = 0, producer_id = 0;
    int32_t batch_length = 0, ple = 0, lod = 0, base_sequence = 0;
    int8_t magic = 0;
    uint32_t crc = 0, record_count = 0, data_len = 0;
    int16_t attributes = 0, producer_epoch = 0;
    do {
        if (len < 8) { st = NISSHI_TRY_GET; break; }
        base_offset = (int64_t)be64(p);
        if (len < 12) { st = NISSHI_TRY_GET; break; }
        batch_length = (int32_t)be32(p + 8);
        if (len < 16) { st = NISSHI_TRY_GET; break; }
        ple = (int32_t)be32(p + 12);
        if (len < 17) { st = NISSHI_TRY_GET; break; }
        magic = (int8_t)p[16];
        if (len < 21) { st = NISSHI_TRY_GET; break; }
        crc = be32(p + 17);
        uint64_t rem = len - 21;
        if (batch_length < 0) { st = NISSHI_TRY_FROM_INT; break; }
        if (batch_length < 9) { st = NISSHI_OVERFLOW; break; }
        if ((uint64_t)batch_length - 9 > rem) { st = NISSHI_OVERFLOW; break; }
        if (rem < 40) { st = NISSHI_TRY_GET; break; }
        attributes = (int16_t)be16(p + 21);
        lod = (int32_t)be32(p + 23);
        base_timestamp = (int64_t)be64(p + 27);
        max_timestamp = (int64_t)be64(p + 35);
        producer_id = (int64_t)be64(p + 43);
        producer_epoch = (int16_t)be16(p + 51);
        base_sequence = (int32_t)be32(p + 53);
        record_count = be32(p + 57);
        if (batch_length < 49) { st = NISSHI_OVERFLOW; break; }
        data_len = (uint32_t)batch_length - 49;
        if ((uint64_t)data_len > len - 61) { st = NISSHI_OVERFLOW; break; }
        int codec = attributes & 7;
        if (codec > 4) { st = NISSHI_UNKNOWN_COMPRESSION; break; }
        if (codec != 0) { st = NISSHI_COMPRESSED; break; }
        // inflated::Batch::try_from(deflated::Batch) rebuilds attributes from
        // BatchAttribute with the compression cleared (inflated.rs:123-140): only
        // bits 3..6 survive.  The by-reference conversion keeps them verbatim.
        if (by_value) attributes = (int16_t)(attributes & 0x78);
    } while (0);
    c.base_offset[b] = base_offset;
    c.batch_length[b] = batch_length;
    c.partition_leader_epoch[b] = ple;
    c.magic[b] = magic;
    c.crc[b] = crc;
    c.attributes[b] = attributes;
    c.last_offset_delta[b] = lod;
    c.base_timestamp[b] = base_timestamp;
    c.max_timestamp[b] = max_timestamp;
    c.producer_id[b] = producer_id;
    c.producer_epoch[b] = producer_epoch;
    c.base_sequence[b] = base_sequence;
    c.record_count[b] = record_count;
    c.crc_computed[b] = 0;
    c.data_off[b] = batch_off[b] + 61;
    c.data_len[b] = data_len;
    c.hdr_base[b] = 0;
    c.hdr_count[b] = 0;
    c.status[b] = st;
    // A record is at least seven bytes, so a count beyond data_len / 7 can only
    // fail: give it no slots and let the sequential walk find the error.
    bool decodable = st == NISSHI_OK;
    bool fits = decodable && (uint64_t)record_count * 7 <= data_len;
    w.alloc_count[b] = fits ? record_count : 0;
    w.flags[b] = decodable && !fits ? kFlagExact : 0;
    // The CRC covers batch_length - 9 bytes from `attributes` on; the reference
    // computes it as soon as those bytes are present (deflated.rs:151-166), i.e.
    // for every status reached after that check.
    bool has_crc = len >= 21 && batch_length >= 9 && (uint64_t)batch_length - 9 <= len - 21;
    uint32_t chunks = 0;
    if (has_crc) {
        uint64_t lo = batch_off[b] + 21;
        uint64_t hi = lo + (uint64_t)batch_length - 9;
        uint64_t grid = lo & ~uint64_t(15);
        chunks = (uint32_t)((hi - grid + kChunkBytes - 1) / kChunkBytes);
        if (hi == lo) chunks = 1;
    }
    w.n_chunks[b] = chunks;
}

// ------------------------------------------------------------ batch-level scans

// Exclusive scan of one value a thread over a workgroup (the encoder's scans).
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &total) {
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    uint32_t offset = 0, sum = 0;
    int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; i++) {
        uint32_t s = lds[i];
        if (i < wid) offset += s;
        sum += s;
    }
    __syncthreads();
    total = sum;
    return offset + inc - v;
}

// Exclusive scans over batches: record slots and CRC work items together, as the
// two halves of one 64-bit sum.  A thread takes kScanItems consecutive batches,
// so a workgroup takes kScanTile in a pass.  Only batches of at least min_chunks
// chunks have CRC work items (the others get their CRC elsewhere).
//
// One workgroup goes through all the batches pass by pass (kScanAll).  From
// kScanTiledFrom batches on that is too slow (3 us a pass), and the scan is two
// launches of a workgroup a tile: kScanSum leaves the tile's sum in
// w.tile_sum, kScanApply starts from the sums of the tiles before its own.
constexpr int kScanItems = 8;
constexpr uint32_t kScanTile = 1024 * kScanItems;
constexpr uint32_t kScanTiledFrom = 4 * kScanTile + 1;
enum ScanMode : uint32_t { kScanAll = 0, kScanSum = 1, kScanApply = 2 };

// The sum of sums[0 .. count) in every thread of a workgroup of 1024.
__device__ inline uint64_t tile_sums_before(const uint64_t *sums, uint32_t count, uint32_t pitch, uint64_t *lds16) {
    uint64_t v = 0;
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) v += sums[(uint64_t)i * pitch];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) lds16[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t total = 0;
    for (uint32_t i = 0; i < (blockDim.x >> 6); i++) total += lds16[i];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(1024) void scan_batches_kernel(uint32_t n, nisshi_columns c, Work w,
                                                            uint32_t min_chunks, const uint64_t *before,
                                                            uint32_t mode) {
    __shared__ uint64_t wave_sum[16];
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // records in the low half, work items in the high half.  A call may decode one
    // slice of a blob: its record and header slots then go on where the slice
    // before it ended, whose totals are `before`.
    uint64_t run = before && mode != kScanSum ? (uint32_t)before[0] : 0;
    if (threadIdx.x == 0 && blockIdx.x == 0 && mode != kScanSum) w.totals[1] = before ? before[1] : 0;
    uint32_t first = 0, last = n;
    if (mode != kScanAll) {
        first = blockIdx.x * kScanTile;
        last = n - first < kScanTile ? n : first + kScanTile;
        if (mode == kScanApply) run += tile_sums_before(w.tile_sum, blockIdx.x, 1, wave_sum);
    }
    bool store = mode != kScanSum;
    for (uint32_t base = first; base < last; base += blockDim.x * kScanItems) {
        uint32_t b0 = base + threadIdx.x * kScanItems;
        uint64_t v[kScanItems], mine = 0;
        // 16-byte loads: the workspace arrays are padded past n, so whole groups can
        // be read (entries past n are masked); a dword a lane would cost four times
        // the load instructions on the one CU this runs on
        uint32_t cnt[kScanItems] = {}, chk[kScanItems] = {};
        if (b0 < n) {
#pragma unroll
            for (int i = 0; i < kScanItems; i += 4) {
                uint4 a = *reinterpret_cast<const uint4 *>(w.alloc_count + b0 + i);
                uint4 k = *reinterpret_cast<const uint4 *>(w.n_chunks + b0 + i);
                cnt[i] = a.x, cnt[i + 1] = a.y, cnt[i + 2] = a.z, cnt[i + 3] = a.w;
                chk[i] = k.x, chk[i + 1] = k.y, chk[i + 2] = k.z, chk[i + 3] = k.w;
            }
        }
#pragma unroll
        for (int i = 0; i < kScanItems; i++) {
            bool in = b0 + i < n;
            uint32_t items = in && chk[i] >= min_chunks ? chk[i] : 0;
            v[i] = ((uint64_t)items << 32) | (in ? cnt[i] : 0u);
            mine += v[i];
        }
        uint64_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_sum[wid] = inc;
        __syncthreads();
        uint64_t before = 0, total = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) {
            uint64_t s = wave_sum[i];
            if (i < wid) before += s;
            total += s;
        }
        __syncthreads();
        uint64_t e = run + before + inc - mine;
        uint32_t rb[kScanItems], cb[kScanItems];
#pragma unroll
        for (int i = 0; i < kScanItems; i++) {
            rb[i] = (uint32_t)e;
            cb[i] = (uint32_t)(e >> 32);
            e += v[i];
        }
        if (!store) {
        } else if (b0 + kScanItems <= n && (reinterpret_cast<uintptr_t>(c.rec_base) & 15) == 0) {
#pragma unroll
            for (int i = 0; i < kScanItems; i += 4)
                *reinterpret_cast<uint4 *>(c.rec_base + b0 + i) = make_uint4(rb[i], rb[i + 1], rb[i + 2], rb[i + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < kScanItems; i++)
                if (b0 + i < n) c.rec_base[b0 + i] = rb[i];
        }
        if (store && b0 < n) {  // chunk_base is a workspace array, padded like the others
#pragma unroll
            for (int i = 0; i < kScanItems; i += 4)
                *reinterpret_cast<uint4 *>(w.chunk_base + b0 + i) = make_uint4(cb[i], cb[i + 1], cb[i + 2], cb[i + 3]);
        }
        run += total;
    }
    if (threadIdx.x != 0) return;
    if (mode == kScanSum) {
        w.tile_sum[blockIdx.x] = run;
    } else if (last == n) {
        w.chunk_base[n] = (uint32_t)(run >> 32);
        w.totals[0] = (uint32_t)run;
        w.totals[5] = run >> 32;
    }
}

// ------------------------------------------------------------ CRC-32C

// The CRC work items first, first + stride, ..., one wave each: item -> (batch, chunk)
// by binary search in the scan.
__device__ __forceinline__ void crc_chunk_items(uint32_t first, uint32_t stride, const uint8_t *blob,
                                                const uint64_t *batch_off, const int32_t *batch_length, uint32_t n,
                                                uint32_t *crc_out, const Work &w, const CrcTables *t, int lane,
                                                const uint32_t *done = nullptr) {
    uint32_t total = w.chunk_base[n];
    for (uint32_t item = first; item < total; item += stride) {
        if (done && done[item]) continue;  // (never the only chunk of a batch)
        uint32_t lo_b = 0, hi_b = n;  // largest b with chunk_base[b] <= item and work items of its own
        while (hi_b - lo_b > 1) {
            uint32_t mid = (lo_b + hi_b) >> 1;
            if (w.chunk_base[mid] <= item) lo_b = mid; else hi_b = mid;
        }
        uint32_t b = lo_b;
        uint32_t chunk = item - w.chunk_base[b];
        uint32_t chunks = w.n_chunks[b];
        uint64_t lo = batch_off[b] + 21;
        uint64_t hi = lo + (uint64_t)batch_length[b] - 9;
        uint64_t grid = lo & ~uint64_t(15);
        uint32_t rows = (uint32_t)((hi - grid + kRowBytes - 1) / kRowBytes);
        uint32_t row0 = chunk * kChunkRows;
        uint32_t row1 = row0 + kChunkRows < rows ? row0 + kChunkRows : rows;
        uint64_t first4 = chunk == 0 ? lo : (uint64_t(1) << 62);
        uint32_t r = crc_rows(blob, grid, lo, hi, first4, row0, row1, lane, t);
        if (chunk + 1 == chunks) {
            // the last row was zero padded up to its end: divide the padding out
            uint64_t pad = grid + (uint64_t)rows * kRowBytes - hi;
            if (hi == lo) pad = 0;
            r = gf_mulmod(r, t->inv8[pad]);
        }
        if (lane == 0) {
            w.partial[item] = r;
            if (chunks == 1) crc_out[b] = hi == lo ? 0u : ~r;
        }
    }
}

// One wave per work item, as many passes as the grid needs.
__global__ void crc_chunks_kernel(const uint8_t *blob, const uint64_t *batch_off,
                                  const int32_t *batch_length, uint32_t n, uint32_t *crc_out,
                                  Work w, const CrcTables *t, const uint32_t *done) {
    crc_chunk_items(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), gridDim.x * kWavesPerBlock, blob, batch_off,
                    batch_length, n, crc_out, w, t, threadIdx.x & 63, done);
}

// The CRC of a batch of more than one chunk from its chunks' residues:
// R = sum over chunks of R_c * x^(8 * bytes after chunk c).
__device__ inline uint32_t crc_combine(const uint64_t *batch_off, const int32_t *batch_length,
                                       uint32_t b, uint32_t chunks, const Work &w, const CrcTables *t) {
    uint64_t lo = batch_off[b] + 21;
    uint64_t hi = lo + (uint64_t)batch_length[b] - 9;
    uint64_t grid = lo & ~uint64_t(15);
    uint64_t last_bytes = hi - (grid + (uint64_t)(chunks - 1) * kChunkBytes);
    const uint32_t *part = w.partial + w.chunk_base[b];
    uint32_t r = part[0], tail = part[chunks - 1];
    uint32_t weight = t->chunk_weight;
    // eight residues loaded before they are chained: one memory latency for eight
    // steps (a 4 MiB batch has 65, and a load a step took 60 us)
    for (uint32_t k = 1; k + 1 < chunks; k += 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) v[j] = k + j + 1 < chunks ? part[k + j] : 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            if (k + j + 1 < chunks) r = gf_mulmod(r, weight) ^ v[j];
    }
    r = gf_mulmod(r, gf_xpow8(t, last_bytes)) ^ tail;
    return ~r;
}

// The same by a whole wave (b and chunks the same in every lane), for batches of
// many chunks: a multiplication mod P is a chain of 32 dependent steps (0.45 us
// for a wave alone on its SIMD), and the chain above has one a chunk.  Here the
// residues are summed as a weighted prefix sum over the lanes,
//   v_i <- v_(i-d) * W^d + v_i,  d = 1, 2, 4 .. 32,
// after which lane 63 holds sum_j R_j * W^(63-j): six multiplications for 64 chunks.
__device__ inline uint32_t crc_combine_wave(const uint64_t *batch_off, const int32_t *batch_length,
                                            uint32_t b, uint32_t chunks, const Work &w, const CrcTables *t,
                                            int lane) {
    uint64_t lo = batch_off[b] + 21;
    uint64_t hi = lo + (uint64_t)batch_length[b] - 9;
    uint64_t grid = lo & ~uint64_t(15);
    uint64_t last_bytes = hi - (grid + (uint64_t)(chunks - 1) * kChunkBytes);
    const uint32_t *part = w.partial + w.chunk_base[b];
    uint32_t m = chunks - 1;  // residues of full chunks, each weighing W against the next
    uint32_t r = 0;
    // 64 at a time from the left; the first group is the short one, padded with zeros on its left
    for (uint32_t done = 0; done < m;) {
        uint32_t size = done == 0 && (m & 63) ? (m & 63) : 64;
        uint32_t pad = 64 - size;
        uint32_t v = (uint32_t)lane >= pad ? part[done + (uint32_t)lane - pad] : 0;
        uint32_t wd = t->chunk_weight;
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t x = __shfl_up(v, d, 64);
            if (lane < d) x = 0;
            v ^= gf_mulmod(x, wd);
            wd = gf_mulmod(wd, wd);
        }
        r = gf_mulmod(r, wd) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 63);  // wd = W^64
        done += size;
    }
    // x^(8 * last_bytes): the product of the table's entries for the bits set, over the lanes
    uint32_t f = lane < 40 && ((last_bytes >> lane) & 1) ? t->pow8[lane] : kOne;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) f = gf_mulmod(f, (uint32_t)__shfl_xor((int)f, d, 64));
    r = gf_mulmod(r, f) ^ part[chunks - 1];
    return ~r;
}

__global__ void crc_finish_kernel(const uint64_t *batch_off, const int32_t *batch_length,
                                  uint32_t n, uint32_t *crc_out, Work w, const CrcTables *t) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    uint32_t chunks = w.n_chunks[b];
    if (chunks < 2) return;
    crc_out[b] = crc_combine(batch_off, batch_length, b, chunks, w, t);
}

// ------------------------------------------------------------ record walk

// A store into a column whose base is the same in every lane: the base stays in
// scalar registers and the lane supplies a 32-bit byte offset, so the address
// costs no vector arithmetic.
template <class T>
__device__ __forceinline__ void put(T *base, uint32_t byte_off, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<uint8_t *>(base) + byte_off) = v;
}

// The record at index r0 + lane; r0 is uniform.  Each group of fields is one
// store of 16 bytes a lane.
__device__ __forceinline__ void store_record(const nisshi_columns &c, uint64_t r0, uint32_t lane,
                                             const RecordFields &f, uint32_t hdr_slot) {
    // the lane's share of the addresses is cheap to form here; left visible, the
    // compiler keeps one 64-bit pointer per column and lane alive across the walk
    asm volatile("" : "+v"(lane));
    put(reinterpret_cast<u32x4 *>(c.rkey + r0), lane * 16,
        u32x4{(uint32_t)f.length, (uint32_t)f.offset_delta, f.key_off, (uint32_t)f.key_len});
    put(reinterpret_cast<u32x4 *>(c.rval + r0), lane * 16,
        u32x4{f.val_off, (uint32_t)f.val_len, hdr_slot, (uint32_t)f.hdr_count});
    put(c.timestamp_delta + r0, lane * 8, f.timestamp_delta);
    put(c.rattributes + r0, lane, (uint8_t)f.attributes);
}

// One header's spans at `slot`.
__device__ __forceinline__ void store_header(const nisshi_columns &c, uint64_t slot, uint32_t ko,
                                             int32_t kl, uint32_t vo, int32_t vl) {
    *reinterpret_cast<u32x4 *>(c.hdr + slot) = u32x4{ko, (uint32_t)kl, vo, (uint32_t)vl};
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Keeps the compiler from moving LDS accesses across this point.  One wave's
// LDS instructions execute in order, so a block of one wave needs no more.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// Header slots are handed out from one cursor in runs, so that a batch's
// headers sit in a few contiguous pieces and a record's headers are always
// contiguous; the order of the pieces between batches is not defined.
constexpr uint32_t kHdrRunMax = 4096;

struct HeaderRuns {
    unsigned long long *cursor;
    uint64_t cap;
    uint32_t run_pos, run_end;   // slots in hand (slot numbers fit 32 bits: the columns hold them so)
    uint32_t total, first_slot;  // of the batch so far
    uint32_t run_max = kHdrRunMax;  // slots taken ahead at most

    // Slots for one round of `m` records, `left` records of the batch following
    // it; lane's record has hc headers (0 in lanes without a record).  Returns
    // false when the header columns are full.  When the round's slots are one
    // piece, `piece` is its first slot.
    __device__ __forceinline__ bool take(uint32_t hc, uint32_t m, uint32_t left, int lane,
                                         uint32_t &slot, uint32_t &piece, bool &one_piece) {
        // equal header counts in every record (the usual case) need no scan
        uint32_t hc0 = uni(hc);
        bool live = (uint32_t)lane < m;
        uint32_t inc = __all(hc == (live ? hc0 : 0u)) ? (live ? ((uint32_t)lane + 1) * hc0 : m * hc0)
                                                       : wave_inclusive_scan(hc, lane);
        uint32_t sum = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        // records whose headers fit in the run in hand use it, the rest a new run
        uint32_t room = run_end - run_pos;
        bool fits = inc <= room;
        uint64_t fit_mask = __ballot(fits);
        uint32_t n_fit = fit_mask == ~0ull ? 64 : (uint32_t)__ffsll((long long)~fit_mask) - 1;
        uint32_t used_old = n_fit ? (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)n_fit - 1) : 0;
        uint32_t rest = sum - used_old;
        uint64_t new_base = 0;
        uint32_t grab = 0;
        if (rest) {
            // take enough for the records still to come at this round's density
            uint64_t ahead = (uint64_t)sum * left / m;
            grab = rest + (uint32_t)(ahead < run_max ? ahead : run_max);
            if (lane == 0) new_base = atomicAdd(cursor, (unsigned long long)grab);
            new_base = uni64(new_base);
            if (new_base + grab > cap || new_base + grab > 0xffffffffull) return false;
        }
        one_piece = rest == 0 || used_old == 0;
        piece = rest ? (uint32_t)new_base : run_pos;
        slot = (fits ? run_pos : (uint32_t)new_base) + inc - hc - (fits ? 0 : used_old);
        if (rest) {
            run_pos = (uint32_t)new_base + rest;
            run_end = (uint32_t)new_base + grab;
        } else {
            run_pos += sum;
        }
        if (total == 0)
            first_slot = (uint32_t)__builtin_amdgcn_readlane((int)slot, __ffsll((long long)__ballot(hc > 0)) - 1);
        total += sum;
        return true;
    }

    // The same for m records of hc headers each (hc > 0, uniform): when the run in
    // hand holds them all, which it does but for a few rounds of a batch, the slots
    // follow from the lane number.
    __device__ __forceinline__ bool take_uniform(uint32_t hc, uint32_t m, uint32_t left, int lane,
                                                 uint32_t &slot, uint32_t &piece, bool &one_piece) {
        uint32_t sum = m * hc;
        if (sum <= run_end - run_pos) {
            slot = run_pos + (uint32_t)lane * hc;
            piece = run_pos;
            one_piece = true;
            if (total == 0) first_slot = run_pos;
            run_pos += sum;
            total += sum;
            return true;
        }
        return take((uint32_t)lane < m ? hc : 0u, m, left, lane, slot, piece, one_piece);
    }
};

// How a wave's walk of a batch ended.
constexpr int kWalkDone = 0;
constexpr int kWalkExact = 1;   // a record that does not decode, or whose length disagrees: exact_walk
constexpr int kWalkGlobal = 2;  // records too large for the LDS ring: walk_global
constexpr int kWalkChain = 3;   // records whose sizes or shapes change from one to the next: walk_chain

// ---- the walk from global memory
//
// One workgroup of kWalkWaves waves per batch.  Record starts are found up to
// 64 * kWalkWaves at a time by speculation: thread i assumes that the i-th next
// record starts i * S bytes on, S being the size of the record before, and
// decodes the length varint there.  Threads up to and including the first one
// whose size differs from S did start on a record, and the walk resumes after
// that thread's record: a wave's records count only when every wave before it
// found all 64 of its guesses right, which the waves tell each other through LDS.
// Each resolved record is then decoded by its thread and must end where its
// length said.  This is the walk for batches the streamed walk leaves: large
// batches, records too large for its ring, or of a shape it cannot express.
// Starts at record k0, byte pos0 of the record data.  Every thread of the
// workgroup calls it, with the same arguments; all return the same value.
constexpr int kWalkWaves = 4;      // the usual workgroup
constexpr int kWalkWavesWide = 8;  // where the call has few batches, so that idle workgroups of 512 threads are few too

struct WalkShare {            // one wave's word to the others, two rounds' worth
    uint32_t taken[2];        // records of the wave that count if the waves before it were full
    uint32_t full[2];         // all 64 guesses were records of size S
    uint32_t next_pos[2];     // where the walk goes on after the wave's last record
    uint32_t next_size[2];    // and that record's size
    uint32_t bad[2];          // the first guess of the round is no record
    // found while the records are decoded, read only after the round's second meeting point
    uint32_t wrong[2];        // a record did not decode, or did not end where its length said
    uint32_t full_cap[2];     // the header columns are full
};

template <int kW>
__device__ __forceinline__ int walk_global(const uint8_t *blob, uint32_t b, const nisshi_columns &c,
                                           HeaderRuns &runs, uint32_t &first_rec, uint32_t k0, uint32_t pos0,
                                           int lane, int wave, WalkShare *share) {
    uint32_t count = uni(c.record_count[b]);
    uint64_t rbase = uni(c.rec_base[b]);
    Reader rd(blob + uni64(c.data_off[b]), uni(c.data_len[b]));
    uint32_t pos = pos0;
    uint32_t S = count > k0 ? (rd.len - pos0) / (count - k0) : 0;
    uint32_t tid = (uint32_t)wave * kWave + (uint32_t)lane;
    constexpr uint32_t kWide = kWave * kW;
    int outcome = kWalkDone;
    uint32_t round = 0;
    for (uint32_t k = k0; k < count && outcome == kWalkDone; round++) {
        uint32_t par = round & 1;
        uint32_t m = count - k < kWide ? count - k : kWide;
        // Touch where the round after this one will look, so that its lines are on
        // their way from HBM while this round is parsed.
        uint32_t touched = 0;
        {
            uint64_t ahead = (uint64_t)pos + (uint64_t)(m + tid) * S;
            typedef __attribute__((address_space(1))) const uint32_t gbl_u32;
            if (ahead + 4 <= rd.len)
                touched = *(gbl_u32 *)((reinterpret_cast<uintptr_t>(rd.data) + ahead) & ~uintptr_t(3));
        }
        // ---- where the records start
        uint64_t guess64 = (uint64_t)pos + (uint64_t)tid * S;
        bool in = tid < m && guess64 < rd.len;
        uint32_t guess = in ? (uint32_t)guess64 : 0;
        int32_t len = 0;
        uint32_t used = 0;
        bool ok = in && fast_varint(rd.get64(guess), used, len);
        // the whole record lies inside the data (this also catches a varint cut short)
        ok = ok && len >= 0 && (uint64_t)guess + used + (uint32_t)len <= rd.len;
        uint32_t size = ok ? used + (uint32_t)len : 0;
        uint64_t okmask = __ballot(ok);
        uint64_t same = __ballot(ok && size == S);
        uint32_t good = ~same ? (uint32_t)__ffsll((long long)~same) - 1 : 64;
        uint32_t v = good;  // lanes below `good` are records of size S
        if (good < 64 && ((okmask >> good) & 1)) v = good + 1;
        uint32_t mine = m > (uint32_t)wave * kWave ? m - (uint32_t)wave * kWave : 0;  // the wave's share of m
        if (mine > (uint32_t)kWave) mine = kWave;
        if (v > mine) v = mine;
        v = uni(v);
        if (lane == 0) {
            share[wave].taken[par] = v;
            share[wave].full[par] = same == ~0ull ? 1 : 0;
            share[wave].next_size[par] = v ? (uint32_t)__builtin_amdgcn_readlane((int)size, (int)(v ? v - 1 : 0)) : 0;
            share[wave].next_pos[par] = v ? (uint32_t)__builtin_amdgcn_readlane((int)guess, (int)(v ? v - 1 : 0)) +
                                                (uint32_t)__builtin_amdgcn_readlane((int)size, (int)(v ? v - 1 : 0))
                                          : 0;
            share[wave].bad[par] = wave == 0 && !(okmask & 1) ? 1 : 0;
            share[wave].wrong[par] = 0;
            share[wave].full_cap[par] = 0;
        }
        __syncthreads();
        // ---- which waves' records count: all up to the first that is not full
        uint32_t taken_before = 0, n_taken = 0, my_taken = 0;
        bool open = true;
        uint32_t new_pos = pos, new_S = S;
#pragma unroll
        for (int q = 0; q < kW; q++) {
            uint32_t tq = share[q].taken[par];
            bool counts = open && tq > 0;
            if (q == wave) my_taken = counts ? tq : 0, taken_before = n_taken;
            if (counts) {
                n_taken += tq;
                new_pos = share[q].next_pos[par];
                new_S = share[q].next_size[par];
            }
            open = open && share[q].full[par] && tq == (uint32_t)kWave;
        }
        if (share[0].bad[par]) {
            outcome = kWalkExact;  // the same in every wave
            break;
        }
        // ---- one thread per record: the fields after the length
        bool live = (uint32_t)lane < my_taken;
        uint32_t end = guess + size;
        uint32_t p = guess + used;
        ok = live;
        RecordFields f;
        f.length = len;
        f.attributes = 0;
        f.timestamp_delta = 0;
        f.offset_delta = 0;
        f.key_off = f.val_off = 0;
        f.key_len = f.val_len = -1;
        f.hdr_count = 0;
        if (live) {
            uint32_t n;
            uint64_t w0 = rd.get64(p);  // attributes, then the timestamp delta
            f.attributes = (uint32_t)w0 & 0xffu;
            if (fast_varlong((uint32_t)(w0 >> 8), n, f.timestamp_delta)) {
                p += 1 + n;
            } else {
                p += 1;
                ok = rd.varlong(p, f.timestamp_delta) == NISSHI_OK;
            }
            ok = fast_varint(rd.get64(p), n, f.offset_delta) && ok;
            p += n;
            ok = fast_varint(rd.get64(p), n, f.key_len) && ok;
            p += n;
            f.key_off = p;
            ok = ok && f.key_len >= -1 && p <= end && (uint32_t)(f.key_len < 0 ? 0 : f.key_len) <= end - p;
            p += ok && f.key_len > 0 ? (uint32_t)f.key_len : 0u;
            ok = fast_varint(rd.get64(p), n, f.val_len) && ok;
            p += n;
            f.val_off = p;
            ok = ok && f.val_len >= -1 && p <= end && (uint32_t)(f.val_len < 0 ? 0 : f.val_len) <= end - p;
            p += ok && f.val_len > 0 ? (uint32_t)f.val_len : 0u;
            ok = fast_varint(rd.get64(p), n, f.hdr_count) && ok;
            p += n;
            // a header is at least two bytes: a count the record cannot hold is an error
            ok = ok && f.hdr_count >= 0 && p <= end && (uint32_t)f.hdr_count <= (end - p) / 2;
        }
        uint32_t hc = ok ? (uint32_t)f.hdr_count : 0;
        uint32_t slot = 0;
        bool no_room = false;
        if (__any(hc != 0)) {
            uint32_t piece;
            bool one_piece;
            uint32_t had = runs.total;
            // what the wave may still have to store in this walk: its share of the records to come
            uint32_t left = (count - k - taken_before - my_taken) / kW;
            if (!runs.take(hc, my_taken, left, lane, slot, piece, one_piece)) {
                no_room = true;
            } else {
                if (had == 0) first_rec = k + taken_before + (uint32_t)__ffsll((long long)__ballot(hc != 0)) - 1;
                for (uint32_t j = 0; j < hc && ok; j++) {
                    uint32_t n, ko, vo;
                    int32_t kl, vl;
                    ok = fast_varint(rd.get64(p), n, kl);
                    p += n;
                    ko = p;
                    ok = ok && kl >= -1 && p <= end && (uint32_t)(kl < 0 ? 0 : kl) <= end - p;
                    p += ok && kl > 0 ? (uint32_t)kl : 0u;
                    ok = fast_varint(rd.get64(p), n, vl) && ok;
                    p += n;
                    vo = p;
                    ok = ok && vl >= -1 && p <= end && (uint32_t)(vl < 0 ? 0 : vl) <= end - p;
                    p += ok && vl > 0 ? (uint32_t)vl : 0u;
                    if (ok) {
                        store_header(c, slot + j, ko, kl, vo, vl);
                    }
                }
            }
        }
        // the fields must end exactly where the length said
        bool wrong = __any(live && !no_room && (!ok || p != end));
        if (live && !no_room && !wrong) store_record(c, rbase + k + taken_before, (uint32_t)lane, f, slot);
        // what went wrong in any wave ends the walk in all of them: told at the next
        // round's meeting point, or at the one after the loop
        if (lane == 0) {
            if (wrong) share[wave].wrong[par] = 1;
            if (no_room) share[wave].full_cap[par] = 1;
        }
        __syncthreads();
        uint32_t any_bad = 0, any_full = 0;
#pragma unroll
        for (int q = 0; q < kW; q++) {
            any_bad |= share[q].wrong[par];
            any_full |= share[q].full_cap[par];
        }
        if (any_full) {
            if (tid == 0) c.status[b] = NISSHI_CAPACITY;
            outcome = kWalkDone;
            asm volatile("" ::"v"(touched));
            break;
        }
        if (any_bad) outcome = kWalkExact;
        k += n_taken;
        pos = new_pos;
        S = new_S;
        asm volatile("" ::"v"(touched));  // the touch is waited for here, a round after it was issued
    }
    return outcome;
}

// ---- the streamed walk
//
// The shape of a record is everything in it but the payload bytes and the values
// of attributes, timestamp_delta and offset_delta: the bytes of every length and
// count varint and the widths of the two deltas.  Records produced by one client
// mostly share a shape, and a record of a known shape needs no varint walk: its
// fields sit at known offsets.  The wave learns the shape of a record as a list
// of masked 32-bit compares (kept in LDS, as are the header spans relative to the
// record start) and tests the records that follow against it, 64 at a time.
constexpr uint32_t kShapeChecks = 24;  // a multiple of four
constexpr uint32_t kShapeHdrs = 16;
constexpr uint32_t kShapeWords = (kShapeChecks + kShapeHdrs + 3) * 4;  // and the learnt scalars

struct Shape {
    uint32_t n_checks;  // 0: none learnt
    uint32_t size;      // bytes of a record of this shape, its length varint included
    uint32_t n_len;     // bytes of the length varint
    int32_t length;     // its value
    uint32_t o_od;      // offsets from the start of the record
    uint32_t o_key, o_val;
    int32_t klen, vlen;
    uint32_t hc;
};

// Reads of the wave's LDS ring at any byte position.  `at` is a position in the
// record data plus the distance of the record data from the ring's origin.  The
// LDS is read in aligned dwords and the bytes wanted are shifted out of them:
// an LDS read that is not aligned stalls the LDS pipeline of the whole CU for
// tens of cycles, and the walk makes a dozen reads a round in every wave.  (A
// read may run up to eight bytes past the ring's end, into its guard.)
template <int kRows>
struct RingView {
    const lds_u32 *ring;
    __device__ __forceinline__ uint32_t read32(uint32_t at) const {
        const lds_u32 *p = ring + (RingCfg<kRows>::wrap(at) >> 2);
        return __builtin_amdgcn_alignbyte(p[1], p[0], at & 3);
    }
    __device__ __forceinline__ uint64_t read64(uint32_t at) const {
        const lds_u32 *p = ring + (RingCfg<kRows>::wrap(at) >> 2);
        uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, at & 3);
        uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, at & 3);
        return ((uint64_t)hi << 32) | lo;
    }
    // The five bytes at ring offset `ro` (already wrapped: ro < kRingBytes), which is
    // what a varint of up to five bytes or an attributes byte and a delta of up to
    // four need: two aligned dwords hold them at any alignment.  Bits above the
    // fortieth are undefined.
    __device__ __forceinline__ uint64_t read40(uint32_t ro) const {
        const lds_u32 *p = ring + (ro >> 2);
        uint32_t d0 = p[0], d1 = p[1];
        uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, ro & 3);
        uint32_t hi = d1 >> (8 * (ro & 3));
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ uint64_t read64_at(uint32_t ro) const {
        const lds_u32 *p = ring + (ro >> 2);
        uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, ro & 3);
        uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, ro & 3);
        return ((uint64_t)hi << 32) | lo;
    }
    // ring offset of a position less than one ring length past an offset inside the ring
    __device__ static __forceinline__ uint32_t wrap1(uint32_t x) {
        uint32_t y = x - (uint32_t)RingCfg<kRows>::kRingBytes;  // wraps to a huge number when x is inside
        return x < y ? x : y;
    }
};

// Learn the shape of the record of `size` bytes at ring position `at0`:
// Record::decode (record.rs:254-277) on that one record, by lane 0, which leaves
// the compares, the header spans and the scalars of the shape in LDS.  The record's
// length varint is known to be valid.  kWalkExact when the record does not decode
// or does not end where its length says, kWalkGlobal when the compares cannot
// express it (a varint of five bytes or more, too many fields).
//
// The fields are walked by one loop: 0 length, 1 timestamp delta (after the
// attributes byte), 2 offset delta, 3 key, 4 value, 5 header count, then a key
// and a value per header.  The bytes of every varint belong to the shape, except
// that of the two deltas only the widths do.
template <int kRows>
__device__ __forceinline__ void learn_shape(lds_u128 *tab, RingView<kRows> ring, uint32_t at0, uint32_t size) {
    lds_u128 *htab = tab + kShapeChecks;
    lds_u128 *out = htab + kShapeHdrs;
    uint32_t p = 0;  // offset in the record
    uint32_t cur_off = 0, cur_mask = 0, cur_pat = 0, noted = 0;
    uint32_t fields = 6;
    uint32_t n_len = 0, o_od = 0, o_key = 0, o_val = 0, hc = 0, ko = 0;
    int32_t length = 0, klen = 0, vlen = 0, kl = 0;
    bool wide = false, bad = false;
    for (uint32_t i = 0; i < fields && !bad; i++) {
        bool delta = i == 1 || i == 2;
        if (i == 1) p += 1;
        uint64_t w = ring.read64(at0 + p);
        uint32_t used;
        int32_t v;
        bad = !fast_varint(w, used, v);
        wide = wide || used > 4;
        // into the compare in hand if it fits, else start the next
        uint32_t bytes = used >= 4 ? 0xffffffffu : ~(0xffffffffu << (8 * used));
        uint32_t mk = delta ? bytes & 0x80808080u : bytes;
        if (p + used > cur_off + 4) {
            if (noted < kShapeChecks) tab[noted] = u32x4{cur_off, cur_mask, cur_pat, 0};
            noted++;
            cur_off = p;
            cur_mask = 0;
            cur_pat = 0;
        }
        uint32_t sh = (p - cur_off) * 8;
        cur_mask |= mk << sh;
        cur_pat |= ((uint32_t)w & mk) << sh;
        if (i == 2) o_od = p;
        p += used;
        if (i == 0) {
            n_len = used;
            length = v;
        } else if (i == 5) {
            // a header is at least two bytes: a count the record cannot hold is an error
            bad = bad || v < 0 || p > size || (uint32_t)v > (size - p) / 2;
            hc = bad ? 0 : (uint32_t)v;
            fields = 6 + 2 * (hc <= kShapeHdrs ? hc : 0);
        } else if (!delta) {
            // Octets::decode: -1 is None, any other negative length or one past the record is an error
            bad = bad || v < -1 || p > size || (uint32_t)(v < 0 ? 0 : v) > size - p;
            uint32_t off = p;
            p += !bad && v > 0 ? (uint32_t)v : 0u;
            if (i == 3) {
                klen = v;
                o_key = off;
            } else if (i == 4) {
                vlen = v;
                o_val = off;
            } else if (!(i & 1)) {
                kl = v;
                ko = off;
            } else {
                htab[(i - 7) >> 1] = u32x4{ko, (uint32_t)kl, off, (uint32_t)v};
            }
        }
    }
    if (noted < kShapeChecks) tab[noted] = u32x4{cur_off, cur_mask, cur_pat, 0};
    noted++;
    // pad to a multiple of four with compares that always hold
    for (uint32_t i = noted; i < ((noted + 3) & ~3u) && i < kShapeChecks; i++) tab[i] = u32x4{0, 0, 0, 0};
    int st = kWalkDone;
    // a varint too wide for the fast decode may be valid: the slower walk decides
    if (wide || hc > kShapeHdrs || noted > kShapeChecks) st = kWalkGlobal;
    // the fields must end exactly where the length said
    else if (bad || p != size) st = kWalkExact;
    out[0] = u32x4{(noted + 3) & ~3u, n_len, (uint32_t)length, o_od};
    out[1] = u32x4{o_key, o_val, (uint32_t)klen, (uint32_t)vlen};
    out[2] = u32x4{hc, (uint32_t)st, 0, 0};
}

// The record walk of one batch by the wave that streams it.  Each round pulls
// the rows it is about to parse through the CRC into the LDS ring, finds up to 64
// record starts by speculation (lane i assumes that the i-th next record starts
// i * S bytes on and decodes the length varint there; lanes before the first one
// whose size differs from S did start on a record), tests those records against
// the shape in hand, learning the first one's shape when it does not fit, and
// emits the records that fit.  On kWalkGlobal *k_out and *pos_out say where the
// slower walk resumes.
template <int kRows>
__device__ __forceinline__ int walk_streamed(uint32_t b, const nisshi_columns &c, HeaderRuns &runs,
                                             BatchStream<kRows> &stream, lds_u128 *tab, int lane,
                                             uint32_t *k_out, uint32_t *pos_out) {
    uint32_t count = uni(c.record_count[b]);
    uint64_t rbase = uni(c.rec_base[b]);
    uint32_t data_len = uni(c.data_len[b]);
    // the record data's offset in the stream: data_off - (batch start + 21) + stream.lo
    uint32_t bias = 40 + stream.lo;
    constexpr int kRingBytes = RingCfg<kRows>::kRingBytes, kAhead = RingCfg<kRows>::kAhead;
    RingView<kRows> ring{stream.ring};
    lds_u128 *htab = tab + kShapeChecks;
    Shape shape;
    shape.n_checks = 0;
    shape.size = 0;
    shape.hc = 0;
    // The stores of a round are issued in the next round, after that round's rows
    // have been taken from the look-ahead queue and before the queue is refilled:
    // the wait for the queued loads then never waits for stores just issued, and
    // every load has a whole round to arrive in.  What is kept per lane is the
    // record's start, its three free fields and its header slot; the rest follows
    // from the shape, which changes only after the flush.
    struct {
        bool any, one_piece;
        uint32_t m, r0, piece;
        uint32_t start, attributes, slot;
        int32_t ts, od;
    } late;
    late.any = false;
    auto flush = [&]() {
        if (!late.any) return;
        late.any = false;
        bool live = (uint32_t)lane < late.m;
        if (live) {
            RecordFields f;
            f.length = shape.length;
            f.attributes = late.attributes;
            f.timestamp_delta = late.ts;
            f.offset_delta = late.od;
            f.key_off = late.start + shape.o_key;
            f.key_len = shape.klen;
            f.val_off = late.start + shape.o_val;
            f.val_len = shape.vlen;
            f.hdr_count = (int32_t)shape.hc;
            store_record(c, late.r0, (uint32_t)lane, f, late.slot);
        }
        if (shape.hc == 0) return;
        if (late.one_piece) {
            uint32_t o = (late.slot - late.piece) * 16;
            for (uint32_t j = 0; j < shape.hc; j++, o += 16) {
                u32x4 e = htab[j];
                if (live)
                    put(reinterpret_cast<u32x4 *>(c.hdr + late.piece), o,
                        u32x4{late.start + e.x, e.y, late.start + e.z, e.w});
            }
        } else {
            for (uint32_t j = 0; j < shape.hc; j++) {
                u32x4 e = htab[j];
                if (live) store_header(c, late.slot + j, late.start + e.x, (int32_t)e.y, late.start + e.z, (int32_t)e.w);
            }
        }
    };
    // how many of the first m lanes' records, counted from lane 0, have the shape in hand
    auto fitting = [&](uint32_t at, uint32_t m) -> uint32_t {
        if (shape.n_checks == 0) return 0;
        uint32_t diff = 0;
        // four compares at a time so that the ring reads overlap (the table is
        // padded with compares that always hold)
        for (uint32_t s = 0; s < shape.n_checks; s += 4) {
            u32x4 e0 = tab[s], e1 = tab[s + 1], e2 = tab[s + 2], e3 = tab[s + 3];
            uint32_t w0 = ring.read32(at + e0.x), w1 = ring.read32(at + e1.x);
            uint32_t w2 = ring.read32(at + e2.x), w3 = ring.read32(at + e3.x);
            diff |= ((w0 ^ e0.z) & e0.y) | ((w1 ^ e1.z) & e1.y) | ((w2 ^ e2.z) & e2.y) | ((w3 ^ e3.z) & e3.y);
        }
        uint64_t fits = __ballot((uint32_t)lane < m && diff == 0);
        return uni(~fits ? (uint32_t)__ffsll((long long)~fits) - 1 : 64);
    };
    // speculate only as far as the ring reaches
    constexpr uint32_t kReach = kRingBytes - 2 * kRowBytes;
    // the rows a round at byte p reads: `left` records to go, of size sz, at most
    // `cap` of them within the ring's reach
    auto window = [&](uint32_t p, uint32_t sz, uint32_t cap, uint32_t left) -> uint32_t {
        uint32_t n = left < cap ? left : cap;
        return stream.row_after(bias + p + n * sz + 64);
    };
    auto reach = [&](uint32_t sz) -> uint32_t {  // records of size sz a round can hold
        uint32_t n = sz ? kReach / sz : (uint32_t)kWave;
        return n < (uint32_t)kWave ? n : (uint32_t)kWave;
    };
    uint32_t pos = 0, k = 0;
    uint32_t S = count ? data_len / count : 0;
    uint32_t cap = reach(S);
    int st = kWalkDone;
    // rounds in a row that found fewer than kChainBelow of the records they looked
    // for: after two the sizes are taken to change from record to record, which is
    // the chained walk's case
    constexpr uint32_t kChainBelow = 8;
    uint32_t strikes = 0;
#ifdef NISSHI_PHASES
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tmark = __builtin_readcyclecounter();
#define PHASE(i) { uint64_t now_ = __builtin_readcyclecounter(); ph[i] += now_ - tmark; tmark = now_; }
#else
#define PHASE(i)
#endif
    // One pass of the loop is one round.  It first asks for the rows of the round
    // after it, assuming that this round finds all the records it looks for; the
    // rows arrive while the round is parsed from the ring and are moved into the
    // ring at the end of the pass, as far as the round after turned out to need
    // them.  (One exit from the loop, and no jumps out of nested blocks: the values
    // that steer it then stay in scalar registers.)
    while (k < count) {
        pos = uni(pos);
        S = uni(S);
        k = uni(k);
        cap = uni(cap);
        if (S > kReach) {  // records the ring cannot hold
            st = kWalkGlobal;
            break;
        }
        PHASE(0)
        u32x4 q[kAhead];
        // All there already, but for the first round and a new S.  (Through the
        // queue only with the twelve-row ring: with eight rows at 128 registers the
        // compiler spills.)
        if (kRows == 12)
            stream.pull_wide(window(pos, S, cap, count - k), q);
        else
            stream.pull(window(pos, S, cap, count - k));
        PHASE(1)
        uint32_t lanes = count - k < cap ? count - k : cap;
        uint32_t ahead_to = window(pos + lanes * S, S, cap, count - k - lanes);
        uint32_t n_ahead = ahead_to > stream.next_row ? ahead_to - stream.next_row : 0;
        if (n_ahead > (uint32_t)kAhead) n_ahead = kAhead;
        stream.load(q, n_ahead);
        PHASE(2)
        flush();  // the stores of the round before, behind the loads: no wait in between
        lds_order();
        PHASE(3)
        // ---- which records the round takes: m of them, one a lane from `pos` on
        uint32_t start = pos + (uint32_t)lane * S;
        uint32_t at = start + bias;
        uint32_t m = 0;
        // The usual round: the records have the size of the shape in hand and lie
        // inside the data, so testing them against the shape finds them; no
        // separate look at their lengths is needed (the shape holds the bytes of
        // the length).
        if (shape.n_checks && shape.size == S && (uint64_t)pos + (uint64_t)lanes * S <= data_len)
            m = fitting(at, lanes);
        if (m == 0) {
            // At most twice: when lane 0 is a record of another size and the ring
            // already holds what a round of that size reads, the round is looked for
            // again at once instead of in a pass of its own.
            bool again = true;
            for (int attempt = 0; attempt < 2 && again; attempt++) {
                again = false;
                lanes = count - k < cap ? count - k : cap;
                start = pos + (uint32_t)lane * S;
                at = start + bias;
                // ---- where the records start: lane i decodes the length at pos + i * S
                uint64_t guess64 = (uint64_t)pos + (uint64_t)lane * S;
                bool in = (uint32_t)lane < lanes && guess64 < data_len;
                if (!in) start = 0, at = bias;
                int32_t len = 0;
                uint32_t used = 0;
                bool ok = in && fast_varint(ring.read64(at), used, len);
                // the whole record lies inside the data (this also catches a varint cut short)
                ok = ok && len >= 0 && (uint64_t)start + used + (uint32_t)len <= data_len;
                uint32_t size = ok ? used + (uint32_t)len : 0;
                uint64_t okmask = __ballot(ok);
                uint64_t same = __ballot(ok && size == S);
                uint32_t same_size = uni(~same ? (uint32_t)__ffsll((long long)~same) - 1 : 64);
                if (!(okmask & 1)) {
                    st = kWalkExact;
                } else if (same_size == 0) {
                    S = uni(size);  // lane 0 is a record, of another size: guess again with that
                    cap = reach(S);
                    again = attempt == 0 && S <= kReach && window(pos, S, cap, count - k) <= stream.next_row;
                } else {
                    // ---- which of them have the shape in hand; learn the first one's if it has not
                    m = fitting(at, same_size);
                    if (m == 0) {
                        if (lane == 0) learn_shape(tab, ring, pos + bias, S);
                        lds_order();
                        const lds_u128 *in4 = htab + kShapeHdrs;
                        u32x4 a = in4[0], b4 = in4[1], c4 = in4[2];
                        st = (int)uni(c4.y);
                        shape.n_checks = st == kWalkDone ? uni(a.x) : 0;
                        shape.size = S;
                        shape.n_len = uni(a.y);
                        shape.length = (int32_t)uni(a.z);
                        shape.o_od = uni(a.w);
                        shape.o_key = uni(b4.x);
                        shape.o_val = uni(b4.y);
                        shape.klen = (int32_t)uni(b4.z);
                        shape.vlen = (int32_t)uni(b4.w);
                        shape.hc = uni(c4.x);
                        m = fitting(at, same_size);
                        if (st == kWalkDone && m == 0) st = kWalkExact;  // cannot happen: a record fits its own shape
                    }
                }
            }
            if (st != kWalkDone) {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // as at the end of the pass
                break;
            }
        }
        strikes = m < kChainBelow && lanes >= 2 * kChainBelow ? strikes + 1 : 0;
        if (strikes >= 2) {
            st = kWalkChain;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // as at the end of the pass
            break;
        }
        PHASE(5)
        if (m) {
            // ---- the free fields of the records that fit
            uint32_t w_used;
            int64_t ts;
            uint64_t r1 = ring.read64(at + shape.n_len);  // attributes, then the timestamp delta
            late.attributes = (uint32_t)r1 & 0xffu;
            fast_varlong((uint32_t)(r1 >> 8), w_used, ts);
            late.ts = (int32_t)ts;  // a shape's timestamp delta has at most four bytes
            fast_varint(ring.read64(at + shape.o_od), w_used, late.od);
            late.slot = 0;
            late.one_piece = true;
            late.piece = 0;
            bool full = false;
            if (shape.hc)
                full = !runs.take_uniform(shape.hc, m, count - k - m, lane, late.slot, late.piece, late.one_piece);
            if (full) {
                if (lane == 0) c.status[b] = NISSHI_CAPACITY;
                k = count;  // ends the walk, the batch has its status
            } else {
                late.any = true;
                late.m = m;
                late.r0 = (uint32_t)rbase + k;
                late.start = start;
                pos += m * S;
                k += m;
            }
        }
        lds_order();
        PHASE(6)
        // ---- the rows asked for at the top, as far as the next round reads
        uint32_t need = k < count && S <= kReach ? window(pos, S, cap, count - k) : stream.next_row;
        uint32_t n_take = need > stream.next_row ? need - stream.next_row : 0;
        stream.take_loaded(q, n_take < n_ahead ? n_take : n_ahead);
        // Rows loaded and not taken are dropped.  Say that no load is outstanding
        // here (s_waitcnt vmcnt(0); in the usual pass none is, the rows taken have
        // arrived): otherwise the compiler, which cannot count across the loop
        // edge, waits before every load of the next pass.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        lds_order();
        PHASE(7)
    }
#ifdef NISSHI_PHASES
    if (lane == 0 && (b % 1024) == 7)
        printf("b=%u other=%lu pull=%lu load=%lu flush=%lu spec=%lu fit=%lu emit=%lu take=%lu\n", b, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
#endif
    // a shape that failed to be learnt has overwritten nothing the flush needs:
    // learning follows the flush of its round
    flush();
    *k_out = k;
    *pos_out = pos;
    return st;
}

// ---- the chained walk
//
// Records whose sizes change from one to the next cannot be found by guessing a
// stride.  Here the wave follows the length varints as the reference's loop does
// (deflated.rs:437-440 with record.rs:254-277), one hop a record: the position is
// the same in every lane, so the hop runs on the scalar unit (an LDS read of two
// dwords, the varint, an add) and costs the vector pipe a handful of instructions.  Up
// to 64 starts, or as many as end inside what the ring holds, go one to a lane;
// then every lane decodes its record's fields from the ring (Record::decode, any
// shape: wide varints, any number of headers) and must end where the length said.
// Starts at record *k_io, byte *pos_io of the record data and leaves there where it
// stopped.  kWalkExact: a record does not decode as its length says; kWalkGlobal:
// the next record is larger than the ring.

// The hops of the chained walk over records whose length varint has kBytes bytes
// (1: records under 64 bytes, 2: under 8 KiB), written by hand: the scalar unit
// is one to a CU, shared by all its waves, so every scalar instruction of a hop
// is a cycle no other wave of the CU gets; the compiler's form of this loop took
// about sixty, this one takes seventeen (two-byte lengths) or fourteen.
//
// The loop runs with one lane enabled, the lane whose record is being looked at:
// EXEC starts as lane n0 and moves up a lane a hop, so that the vector moves
// below write that lane only.  A hop: the two aligned dwords that hold the
// length (vector address arithmetic and LDS reads under the one-lane EXEC), the
// bytes shifted into place and read back to a scalar; the form of the length
// (continuation bits as expected, not negative: else out); its value plus kBytes
// is the record's size; out if that is more than `left`, the bytes up to the
// round's limit; the lane notes `left` (its start, counted back from the limit);
// `left` and the ring offset `ro` move on; `rem`, the records still wanted less
// one, counts down and ends the loop by its borrow.
// Leaves EXEC as it found it.  Records found: the starting rem - the final rem
// (in 32-bit arithmetic: the borrow leaves rem at 0xFFFFFFFF).
template <int kBytes, uint32_t kRing>
__device__ __forceinline__ void hop_loop(uint32_t ring_at, uint32_t n0, uint32_t &ro, uint32_t &left, uint32_t &rem,
                                         uint32_t &my_left) {
    static_assert(kBytes == 1 || kBytes == 2, "one- and two-byte lengths only");
    uint32_t va, vsh, d0, d1, vlo, lo, t, a, h;
    uint64_t save;
    if (kBytes == 2) {
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "s_lshl_b64 exec, 1, %[n0]\n"
            "1:\n\t"
            "v_add_u32_e32 %[va], %[ro], %[base]\n\t"
            "v_and_b32_e32 %[vsh], 3, %[va]\n\t"
            "v_and_b32_e32 %[va], -4, %[va]\n\t"
            "ds_read_b32 %[d0], %[va]\n\t"
            "ds_read_b32 %[d1], %[va] offset:4\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbyte_b32 %[vlo], %[d1], %[d0], %[vsh]\n\t"
            "s_nop 0\n\t"                              // a lane read of a register a vector instruction has just written needs a wait state
            "v_readfirstlane_b32 %[lo], %[vlo]\n\t"
            "s_and_b32 %[t], %[lo], 0x8081\n\t"      // second byte ends the varint, first continues, value not negative
            "s_cmp_lg_u32 %[t], 0x80\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_bfe_u32 %[a], %[lo], 0x60001\n\t"     // zigzag: the value is the 13 bits above bit 0
            "s_bfe_u32 %[h], %[lo], 0x70008\n\t"
            "s_lshl_b32 %[h], %[h], 6\n\t"
            "s_add_u32 %[a], %[a], %[h]\n\t"
            "s_add_u32 %[a], %[a], 2\n\t"
            "s_sub_u32 %[t], %[left], %[a]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "v_mov_b32_e32 %[st], %[left]\n\t"
            "s_mov_b32 %[left], %[t]\n\t"
            "s_add_u32 %[ro], %[ro], %[a]\n\t"
            "s_sub_u32 %[t], %[ro], %[ring]\n\t"
            "s_min_u32 %[ro], %[ro], %[t]\n\t"
            "s_lshl_b64 exec, exec, 1\n\t"
            "s_sub_u32 %[rem], %[rem], 1\n\t"
            "s_cbranch_scc0 1b\n"
            "2:\n\t"
            "s_mov_b64 exec, %[save]\n\t"
            : [ro] "+s"(ro), [left] "+s"(left), [rem] "+s"(rem), [st] "+v"(my_left), [va] "=&v"(va), [vsh] "=&v"(vsh),
              [d0] "=&v"(d0), [d1] "=&v"(d1), [vlo] "=&v"(vlo), [lo] "=&s"(lo), [t] "=&s"(t), [a] "=&s"(a),
              [h] "=&s"(h), [save] "=&s"(save)
            : [base] "v"(ring_at), [n0] "s"(n0), [ring] "i"(kRing)
            : "memory", "scc");
    } else {
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "s_lshl_b64 exec, 1, %[n0]\n"
            "1:\n\t"
            "v_add_u32_e32 %[va], %[ro], %[base]\n\t"
            "v_and_b32_e32 %[vsh], 3, %[va]\n\t"
            "v_and_b32_e32 %[va], -4, %[va]\n\t"
            "ds_read_b32 %[d0], %[va]\n\t"
            "ds_read_b32 %[d1], %[va] offset:4\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_alignbyte_b32 %[vlo], %[d1], %[d0], %[vsh]\n\t"
            "s_nop 0\n\t"                              // a lane read of a register a vector instruction has just written needs a wait state
            "v_readfirstlane_b32 %[lo], %[vlo]\n\t"
            "s_and_b32 %[t], %[lo], 0x81\n\t"        // the first byte ends the varint, value not negative
            "s_cmp_lg_u32 %[t], 0\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_bfe_u32 %[a], %[lo], 0x60001\n\t"
            "s_add_u32 %[a], %[a], 1\n\t"
            "s_sub_u32 %[t], %[left], %[a]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "v_mov_b32_e32 %[st], %[left]\n\t"
            "s_mov_b32 %[left], %[t]\n\t"
            "s_add_u32 %[ro], %[ro], %[a]\n\t"
            "s_sub_u32 %[t], %[ro], %[ring]\n\t"
            "s_min_u32 %[ro], %[ro], %[t]\n\t"
            "s_lshl_b64 exec, exec, 1\n\t"
            "s_sub_u32 %[rem], %[rem], 1\n\t"
            "s_cbranch_scc0 1b\n"
            "2:\n\t"
            "s_mov_b64 exec, %[save]\n\t"
            : [ro] "+s"(ro), [left] "+s"(left), [rem] "+s"(rem), [st] "+v"(my_left), [va] "=&v"(va), [vsh] "=&v"(vsh),
              [d0] "=&v"(d0), [d1] "=&v"(d1), [vlo] "=&v"(vlo), [lo] "=&s"(lo), [t] "=&s"(t), [a] "=&s"(a),
              [save] "=&s"(save)
            : [base] "v"(ring_at), [n0] "s"(n0), [ring] "i"(kRing)
            : "memory", "scc");
        (void)h;
    }
}

// LongVarInt::decode at ring offset `ro` for deltas of more than four bytes: up
// to ten.  False on an eleventh byte (Overflow in the reference).
template <int kRows>
__device__ __forceinline__ bool ring_varlong(const RingView<kRows> &ring, uint32_t ro, uint32_t &used, int64_t &out) {
    uint64_t w = ring.read64_at(ro);
    uint64_t w2 = ring.read40(RingView<kRows>::wrap1(ro + 8));
    uint64_t acc = 0;
    used = 0;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        uint32_t byte = (uint32_t)((i < 8 ? w >> (8 * i) : w2 >> (8 * (i - 8))) & 0xffu);
        if (!done) {
            acc |= (uint64_t)(byte & 0x7fu) << (7 * i);
            used = i + 1;
            done = !(byte & 0x80u);
        }
    }
    out = (int64_t)(acc >> 1) ^ -(int64_t)(acc & 1u);
    return done;
}

template <int kRows>
__device__ __forceinline__ int walk_chain(uint32_t b, const nisshi_columns &c, HeaderRuns &runs,
                                          BatchStream<kRows> &stream, int lane, uint32_t *k_io, uint32_t *pos_io) {
    typedef RingView<kRows> Ring;
    uint32_t count = uni(c.record_count[b]);
    uint64_t rbase = uni(c.rec_base[b]);
    uint32_t data_len = uni(c.data_len[b]);
    uint32_t bias = 40 + stream.lo;  // the record data's offset in the stream
    constexpr uint32_t kRingBytes = RingCfg<kRows>::kRingBytes;
    constexpr int kAhead = RingCfg<kRows>::kAhead;
    constexpr uint32_t kReach = kRingBytes - 2 * kRowBytes;  // a round's records end within this of its start
    Ring ring{stream.ring};
    uint32_t k = uni(*k_io), pos = uni(*pos_io);
    int st = kWalkDone;
#ifdef NISSHI_PHASES
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tmark = __builtin_readcyclecounter();
    uint32_t n_rounds = 0, k_first = k;
#endif
    while (k < count) {
        pos = uni(pos);
        k = uni(k);
        PHASE(0)
        // the ring holds the rows up to `have`; this round reads up to kReach + 64 bytes from pos
        u32x4 q[kAhead];
        uint32_t target = stream.row_after(bias + pos + kReach + 64);
        if (kRows == 12)
            stream.pull_wide(target, q);
        else
            stream.pull(target);
        // the rows after those, asked for now and moved in at the end of the pass, as
        // far as the next round's start then allows
        uint32_t n_ahead = stream.rows - stream.next_row;
        if (n_ahead > (uint32_t)kAhead) n_ahead = kAhead;
        stream.load(q, n_ahead);
        lds_order();
        PHASE(1)
        // ---- the hops: one record start a lane
        uint32_t want = count - k < (uint32_t)kWave ? count - k : (uint32_t)kWave;
        uint32_t limit = data_len - pos < kReach ? data_len : pos + kReach;
        uint32_t p = pos, n = 0, ro = RingCfg<kRows>::wrap(pos + bias);
        uint32_t my_start = 0;
        {
            // lengths of one byte and of two, each by a loop of its own (hop_loop); a
            // change of form, or anything else that stops a loop, is looked at here
            uint32_t my_left = 0;  // the lane's start as the distance to `limit`
            uint32_t ring_at = (uint32_t)(uintptr_t)ring.ring;
            for (int pass = 0; pass < 4 && n < want && p < limit; pass++) {
                const lds_u32 *w = ring.ring + (ro >> 2);
                uint32_t d0 = uni(w[0]), d1 = uni(w[1]);
                uint32_t lo = (uint32_t)((((uint64_t)d1 << 32) | d0) >> (8 * (ro & 3)));
                uint32_t left = limit - p, rem = want - n - 1;
                if (!(lo & 0x80u))
                    hop_loop<1, kRingBytes>(ring_at, n, ro, left, rem, my_left);
                else if (!(lo & 0x8000u))
                    hop_loop<2, kRingBytes>(ring_at, n, ro, left, rem, my_left);
                else
                    break;
                uint32_t found = want - n - 1 - rem;
                if (found == 0) break;
                n += found;
                p = limit - left;
            }
            my_start = limit - my_left;
        }
        PHASE(2)
        // why the loop ended, when it did not find all it wanted: the general decode
        // of the length at p decides
        int stop = 0;  // 1: the record does not lie inside the data; 2: it is larger than the ring
        while (n < want) {
            if (p >= data_len) {
                stop = 1;
                break;
            }
            const lds_u32 *w = ring.ring + (ro >> 2);
            uint32_t d0 = uni(w[0]), d1 = uni(w[1]);
            uint64_t v = (((uint64_t)d1 << 32) | d0) >> (8 * (ro & 3));
            uint32_t used;
            int32_t len;
            bool ok = fast_varint(v, used, len);
            uint32_t sz = used + (uint32_t)len;
            if (!ok || len < 0 || sz > data_len - p) {
                stop = 1;
                break;
            }
            if (sz > limit - p) {
                if (n == 0) stop = 2;
                break;
            }
            // a length of three bytes and more that fits: take it and go on one by one
            my_start = (uint32_t)lane == n ? p : my_start;
            n++;
            p += sz;
            ro += sz;
            ro = ro >= kRingBytes ? ro - kRingBytes : ro;
        }
        n = uni(n);
        if (stop) {
            st = stop == 1 ? kWalkExact : kWalkGlobal;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // the rows asked for are dropped
            break;
        }
        PHASE(3)
        // ---- a lane a record: the fields
        bool live = (uint32_t)lane < n;
        uint32_t start = live ? my_start : pos;
        uint32_t ro0 = RingCfg<kRows>::wrap(start + bias);
        RecordFields f;
        uint32_t used, q_used;
        bool ok = fast_varint(ring.read40(ro0), used, f.length) && live;  // known to be good in live lanes
        uint32_t end = start + used + (uint32_t)f.length;
        uint32_t o = used;  // offset in the record
        auto at = [&](uint32_t off) { return Ring::wrap1(ro0 + off); };
        {
            uint64_t w0 = ring.read40(at(o));  // attributes, then the timestamp delta
            f.attributes = (uint32_t)w0 & 0xffu;
            o += 1;
            bool quick = fast_varlong((uint32_t)(w0 >> 8), q_used, f.timestamp_delta);
            if (!quick) ok = ring_varlong(ring, at(o), q_used, f.timestamp_delta) && ok;
            o += q_used;
        }
        uint32_t body = end - start;  // the record's bytes, its length varint included
        ok = fast_varint(ring.read40(at(o)), q_used, f.offset_delta) && ok;
        o += q_used;
        ok = fast_varint(ring.read40(at(o)), q_used, f.key_len) && ok;
        o += q_used;
        f.key_off = start + o;
        ok = ok && f.key_len >= -1 && o <= body && (uint32_t)(f.key_len < 0 ? 0 : f.key_len) <= body - o;
        o += ok && f.key_len > 0 ? (uint32_t)f.key_len : 0u;
        ok = fast_varint(ring.read40(at(o)), q_used, f.val_len) && ok;
        o += ok ? q_used : 0u;
        f.val_off = start + o;
        ok = ok && f.val_len >= -1 && o <= body && (uint32_t)(f.val_len < 0 ? 0 : f.val_len) <= body - o;
        o += ok && f.val_len > 0 ? (uint32_t)f.val_len : 0u;
        ok = fast_varint(ring.read40(at(o)), q_used, f.hdr_count) && ok;
        o += ok ? q_used : 0u;
        // a header is at least two bytes: a count the record cannot hold is an error
        ok = ok && f.hdr_count >= 0 && o <= body && (uint32_t)f.hdr_count <= (body - o) / 2;
        uint32_t hc = live && ok ? (uint32_t)f.hdr_count : 0;
        uint32_t slot = 0;
        bool no_room = false;
        PHASE(4)
        if (__any(hc != 0)) {
            uint32_t piece;
            bool one_piece;
            if (!runs.take(hc, n, count - k - n, lane, slot, piece, one_piece)) {
                no_room = true;
            } else {
                uint32_t most = 0;
                {
                    uint32_t m = hc;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        uint32_t t = (uint32_t)__shfl_xor((int)m, d, 64);
                        m = m > t ? m : t;
                    }
                    most = uni(m);
                }
                for (uint32_t j = 0; j < most; j++) {
                    bool mine = j < hc && ok;
                    int32_t kl = 0, vl = 0;
                    uint32_t ko, vo, n1, n2;
                    bool good = fast_varint(ring.read40(at(mine ? o : 0u)), n1, kl);
                    uint32_t o1 = o + n1;
                    ko = start + o1;
                    good = good && kl >= -1 && o1 <= body && (uint32_t)(kl < 0 ? 0 : kl) <= body - o1;
                    o1 += good && kl > 0 ? (uint32_t)kl : 0u;
                    good = fast_varint(ring.read40(at(mine && good ? o1 : 0u)), n2, vl) && good;
                    o1 += n2;
                    vo = start + o1;
                    good = good && vl >= -1 && o1 <= body && (uint32_t)(vl < 0 ? 0 : vl) <= body - o1;
                    o1 += good && vl > 0 ? (uint32_t)vl : 0u;
                    if (mine) {
                        ok = good;
                        if (good) {
                            o = o1;
                            store_header(c, slot + j, ko, kl, vo, vl);
                        }
                    }
                }
            }
        }
        if (no_room) {
            if (lane == 0) c.status[b] = NISSHI_CAPACITY;
            k = count;  // ends the walk, the batch has its status
            __builtin_amdgcn_s_waitcnt(0x0F70);
            break;
        }
        // the fields must end exactly where the length said
        if (__any(live && (!ok || o != body))) {
            st = kWalkExact;
            __builtin_amdgcn_s_waitcnt(0x0F70);
            break;
        }
        PHASE(5)
        if (live) store_record(c, rbase + k, (uint32_t)lane, f, slot);
        k += n;
        pos = p;
        lds_order();
        PHASE(6)
#ifdef NISSHI_PHASES
        n_rounds++;
#endif
        // ---- the rows asked for at the top, as far as the next round may hold them
        uint32_t need = k < count ? stream.row_after(bias + pos + kReach + 64) : stream.next_row;
        uint32_t n_take = need > stream.next_row ? need - stream.next_row : 0;
        stream.take_loaded(q, n_take < n_ahead ? n_take : n_ahead);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // no load is outstanding across the loop edge
        lds_order();
        PHASE(7)
    }
#ifdef NISSHI_PHASES
    if (lane == 0 && (b % 1024) == 7)
        printf("chain b=%u rounds=%u records=%u other=%lu pull=%lu hops=%lu tail=%lu fields=%lu headers=%lu store=%lu take=%lu\n", b,
               n_rounds, k - k_first, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
#endif
    *k_io = k;
    *pos_io = pos;
    return st;
}

// One wave per batch: CRC-32C of the batch streamed just ahead of the record
// walk through an LDS ring, so the batch is read from HBM once.  Batches of
// kBigChunks chunks or more get their CRC from crc_chunks_kernel instead (one
// wave cannot stream them fast enough) and their records from slow_walk_kernel.
constexpr uint32_t kBigChunks = 8;

template <int kRows, int kWaves, int kPack>
__global__ __launch_bounds__(64 * kPack) __attribute__((amdgpu_waves_per_eu(kWaves, kWaves))) void decode_fused_kernel(const uint8_t *blob,
                                                          const uint64_t *batch_off, uint32_t n,
                                                          nisshi_columns c, Work w, uint64_t rec_cap,
                                                          uint64_t hdr_cap, const CrcTables *t,
                                                          uint32_t chunk_groups) {
    // kPack waves a workgroup, each with a batch, a ring and a shape table of its
    // own: no wave waits for another (there is no barrier below)
    __shared__ __attribute__((aligned(16))) uint32_t rings[kPack][RingCfg<kRows>::kRingWords];
    __shared__ __attribute__((aligned(16))) uint32_t shape_tabs[kPack][kShapeWords];
    int lane = threadIdx.x & 63;
    uint32_t wave = uni(threadIdx.x >> 6);
    uint32_t *ring = rings[wave], *shape_tab = shape_tabs[wave];
    // The first chunk_groups workgroups take the CRC work items of the batches of
    // kBigChunks chunks and more (none, most of the time): in this grid rather than
    // in a launch of their own, and ahead of the batches so that they start first.
    if (blockIdx.x < chunk_groups) {
        crc_chunk_items(blockIdx.x * kPack + wave, chunk_groups * kPack, blob, batch_off, c.batch_length, n,
                        c.crc_computed, w, t, lane);
        return;
    }
    uint32_t b = (blockIdx.x - chunk_groups) * kPack + wave;
    if (b >= n) return;
    uint32_t chunks = uni(w.n_chunks[b]);
    if (chunks == 0 || chunks >= kBigChunks) return;
    bool do_walk = uni(c.status[b]) == NISSHI_OK && !(uni(w.flags[b]) & kFlagExact);
    if (do_walk && (uint64_t)c.rec_base[b] + c.record_count[b] > rec_cap) {
        if (lane == 0) c.status[b] = NISSHI_CAPACITY;
        do_walk = false;
    }
    BatchStream<kRows> stream;
    uint64_t lo = batch_off[b] + 21;
    stream.init(blob, lo, lo + (uint64_t)uni(c.batch_length[b]) - 9, (lds_u32 *)ring, lane);
    if (do_walk) {
        HeaderRuns runs{reinterpret_cast<unsigned long long *>(w.totals + 1), hdr_cap, 0, 0, 0, 0};
        uint32_t k = 0, pos = 0;
        int st = walk_streamed(b, c, runs, stream, (lds_u128 *)shape_tab, lane, &k, &pos);
        // records of changing sizes, or of shapes the compares cannot express: hop by hop
        if (st == kWalkChain || st == kWalkGlobal) st = walk_chain(b, c, runs, stream, lane, &k, &pos);
        if (lane == 0) {
            c.hdr_count[b] = runs.total;
            c.hdr_base[b] = runs.first_slot;
            if (st == kWalkExact) w.flags[b] |= kFlagExact;
            if (st == kWalkGlobal) {
                w.flags[b] |= kFlagGlobal;
                w.resume_k[b] = k;
                w.resume_pos[b] = pos;
            }
        }
    }
    uint32_t v = stream.finish(t);
    if (lane == 0) c.crc_computed[b] = v;
}

// One lane per flagged batch: Record::decode in sequence, exactly as
// Vec<Record>::try_from(deflated::Batch) does (deflated.rs:431-442).  The first
// error ends the batch with its status; records are stored only while the batch
// has slots (a batch whose count cannot fit has none and only gets its status).
__device__ __forceinline__ void exact_walk(const uint8_t *blob, uint32_t b, const nisshi_columns &c,
                                        const Work &w, uint64_t rec_cap, uint64_t hdr_cap) {
    uint32_t count = c.record_count[b];
    uint32_t slots = w.alloc_count[b];
    uint32_t rbase = c.rec_base[b];
    if ((uint64_t)rbase + slots > rec_cap) {
        c.status[b] = NISSHI_CAPACITY;
        return;
    }
    unsigned long long *hdr_cursor = reinterpret_cast<unsigned long long *>(w.totals + 1);
    Reader rd(blob + c.data_off[b], c.data_len[b]);
    uint32_t pos = 0, hdr_total = 0, first_slot = 0;
    int st = NISSHI_OK;
    for (uint32_t k = 0; k < count && st == NISSHI_OK; k++) {
        RecordFields f;
        st = read_record_head(rd, pos, f);
        if (st) break;
        uint64_t slot = 0;
        bool keep = k < slots && f.hdr_count > 0;
        if (keep) {
            // a header is at least two bytes: a count the data cannot hold fails below
            uint32_t want = (uint32_t)f.hdr_count;
            if ((uint64_t)want * 2 > rd.len - pos) {
                keep = false;
            } else {
                slot = atomicAdd(hdr_cursor, (unsigned long long)want);
                if (slot + want > hdr_cap) {
                    c.status[b] = NISSHI_CAPACITY;
                    return;
                }
                if (hdr_total == 0) first_slot = (uint32_t)slot;
            }
        }
        for (int32_t j = 0; j < f.hdr_count; j++) {
            uint32_t ko, vo;
            int32_t kl, vl;
            if ((st = rd.octets(pos, ko, kl))) break;
            if ((st = rd.octets(pos, vo, vl))) break;
            if (keep) {
                store_header(c, slot + j, ko, kl, vo, vl);
            }
        }
        if (st) break;
        if (k < slots) store_record(c, (uint64_t)rbase + k, 0, f, (uint32_t)slot);
        hdr_total += (uint32_t)f.hdr_count;
    }
    c.status[b] = st;
    c.hdr_count[b] = st == NISSHI_OK ? hdr_total : 0;
    c.hdr_base[b] = first_slot;
}

// Batches of a few hundred bytes, a lane each: a workgroup a batch spends more on
// starting than on the batch (a producer that does not linger sends one record a
// batch).  The CRC bit by bit, the records by the sequential walk (exact_walk:
// Record::decode in order, the reference's outcomes by construction).  The batch
// is then taken off the other kernels' hands (no CRC work item, no flag).
constexpr uint32_t kTinyBytes = 1024;

__global__ __launch_bounds__(256) void tiny_batch_kernel(const uint8_t *blob, const uint64_t *batch_off, uint32_t n,
                                                         nisshi_columns c, Work w, uint64_t rec_cap,
                                                         uint64_t hdr_cap) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    uint64_t at = batch_off[b];
    if (batch_off[b + 1] - at > kTinyBytes || w.n_chunks[b] == 0) return;  // no CRC item: the header did not get that far
    const uint8_t *p = blob + at + 21;
    uint32_t bytes = (uint32_t)c.batch_length[b] - 9;
    uint32_t r = 0xffffffffu;
    for (uint32_t i = 0; i < bytes; i++) {
        r ^= p[i];
#pragma unroll
        for (int k = 0; k < 8; k++) r = gf_mulx(r);
    }
    c.crc_computed[b] = bytes ? ~r : 0u;
    w.n_chunks[b] = 0;
    uint32_t flags = w.flags[b];
    if (c.status[b] == NISSHI_OK) {
        exact_walk(blob, b, c, w, rec_cap, hdr_cap);
        if (flags & kFlagExact) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 3), 1ull);
    }
    w.flags[b] = 0;
}

// Everything the streamed walk left, one wave per batch: the CRC of batches of
// kBigChunks chunks or more from their chunks' residues and their record walk
// from global memory; flagged batches from where the streamed walk stopped; the
// sequential walk for batches that need the reference's exact outcome; and the
// batch's share of the totals, its status being final here.
template <int kW>
struct SlowShare {
    WalkShare walk[kW];
    uint32_t hdr_total[kW], hdr_first_rec[kW], hdr_first_slot[kW];
};

template <int kW>
__device__ __forceinline__ void slow_batch(const uint8_t *blob, const uint64_t *batch_off, uint32_t b,
                                           const nisshi_columns &c, const Work &w, uint64_t rec_cap, uint64_t hdr_cap,
                                           const CrcTables *t, SlowShare<kW> &sh) {
    WalkShare *share = sh.walk;
    uint32_t *hdr_total = sh.hdr_total, *hdr_first_rec = sh.hdr_first_rec, *hdr_first_slot = sh.hdr_first_slot;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool lead = threadIdx.x == 0;
    uint32_t flags = uni(w.flags[b]);
    uint32_t chunks = uni(w.n_chunks[b]);
    bool big = chunks >= kBigChunks;
    // the usual batch was done by the streamed walk: nothing left
    if (!big && flags == 0 && uni(c.status[b]) == NISSHI_OK) return;
    if (big && wave == 0) {
        uint32_t v = crc_combine_wave(batch_off, c.batch_length, b, chunks, w, t, lane);
        if (lane == 0) c.crc_computed[b] = v;
    }
    bool ok = uni(c.status[b]) == NISSHI_OK;
    if (ok && !(flags & kFlagExact) && (big || (flags & kFlagGlobal))) {
        HeaderRuns runs{reinterpret_cast<unsigned long long *>(w.totals + 1), hdr_cap, 0, 0, 0, 0};
        runs.run_max = kHdrRunMax * 4 / kW;  // the waves together take no more ahead than four would
        uint32_t k = 0, pos = 0, before = 0, before_slot = 0;
        bool fits = true;
        if (flags & kFlagGlobal) {
            k = uni(w.resume_k[b]);
            pos = uni(w.resume_pos[b]);
            before = uni(c.hdr_count[b]);  // the streamed walk's headers come first
            before_slot = uni(c.hdr_base[b]);
        } else if ((uint64_t)c.rec_base[b] + c.record_count[b] > rec_cap) {
            fits = false;
        }
        if (!fits) {
            if (lead) c.status[b] = NISSHI_CAPACITY;
        } else {
            uint32_t first_rec = 0xffffffffu;
            int st = walk_global<kW>(blob, b, c, runs, first_rec, k, pos, lane, wave, share);
            // the batch's headers: every wave took runs of its own
            if (lane == 0) {
                hdr_total[wave] = runs.total;
                hdr_first_rec[wave] = runs.total ? first_rec : 0xffffffffu;
                hdr_first_slot[wave] = runs.first_slot;
            }
            __syncthreads();
            if (lead) {
                uint32_t total = before, first = 0xffffffffu, slot = before_slot;
                for (int q = 0; q < kW; q++) {
                    total += hdr_total[q];
                    if (!before && hdr_first_rec[q] < first) first = hdr_first_rec[q], slot = hdr_first_slot[q];
                }
                c.hdr_count[b] = total;
                c.hdr_base[b] = slot;
            }
            if (st != kWalkDone) flags |= kFlagExact;
        }
    }
    if (lead) {
        if (ok && (flags & kFlagExact) && c.status[b] == NISSHI_OK) exact_walk(blob, b, c, w, rec_cap, hdr_cap);
        if (c.status[b] != NISSHI_OK) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), 1ull);
        if (flags & kFlagExact) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 3), 1ull);
    }
}

// A workgroup a batch; or, where nearly every batch has nothing left (blobs of
// small batches), per_group batches in turn after a look at all of them at once.
template <bool kGrouped, int kW>
__global__ __launch_bounds__(kWave * kW) void slow_walk_kernel(const uint8_t *blob, const uint64_t *batch_off,
                                                                       uint32_t n, nisshi_columns c, Work w,
                                                                       uint64_t rec_cap, uint64_t hdr_cap,
                                                                       const CrcTables *t, uint32_t per_group) {
    __shared__ SlowShare<kW> sh;
    if (!kGrouped) {
        if (blockIdx.x < n) slow_batch<kW>(blob, batch_off, blockIdx.x, c, w, rec_cap, hdr_cap, t, sh);
        return;
    }
    uint64_t mine = (uint64_t)blockIdx.x * per_group + threadIdx.x;  // per_group <= the workgroup's threads
    bool left = threadIdx.x < per_group && mine < n &&
                (w.n_chunks[mine] >= kBigChunks || w.flags[mine] != 0 || c.status[mine] != NISSHI_OK);
    if (!__syncthreads_or(left)) return;
    for (uint32_t i = 0; i < per_group; i++) {
        uint64_t b = (uint64_t)blockIdx.x * per_group + i;
        if (b >= n) return;
        slow_batch<kW>(blob, batch_off, (uint32_t)b, c, w, rec_cap, hdr_cap, t, sh);
        __syncthreads();  // the next batch reuses the shared state
    }
}

// Sum of record_count over decodable batches (nisshi_hip_count_dev)
__global__ void count_kernel(const uint8_t *blob, const uint64_t *batch_off, uint32_t n,
                             unsigned long long *out) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (b < n) {
        const uint8_t *p = blob + batch_off[b];
        uint64_t len = batch_off[b + 1] - batch_off[b];
        if (len >= 61) {
            int32_t bl = (int32_t)be32(p + 8);
            int codec = be16(p + 21) & 7;
            uint32_t rc = be32(p + 57);
            if (bl >= 49 && (uint64_t)bl - 49 <= len - 61 && codec == 0 &&
                (uint64_t)rc * 7 <= (uint64_t)bl - 49)
                v = rc;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// Single-buffer CRC (nisshi_hip_crc32c_dev): the same chunk kernel over one region.
__global__ void crc_buffer_chunks_kernel(const uint8_t *data, uint64_t len, uint32_t chunks,
                                         uint32_t *partial, const CrcTables *t) {
    uint32_t item = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (item >= chunks) return;
    uint64_t base = reinterpret_cast<uintptr_t>(data) & 15;  // grid in absolute address terms
    const uint8_t *aligned = data - base;
    uint64_t lo = base, hi = base + len;
    uint32_t rows = (uint32_t)((hi + kRowBytes - 1) / kRowBytes);
    uint32_t row0 = item * kChunkRows;
    uint32_t row1 = row0 + kChunkRows < rows ? row0 + kChunkRows : rows;
    uint64_t first4 = item == 0 ? lo : (uint64_t(1) << 62);
    uint32_t r = crc_rows(aligned, 0, lo, hi, first4, row0, row1, lane, t);
    if (item + 1 == chunks) r = gf_mulmod(r, t->inv8[(uint64_t)rows * kRowBytes - hi]);
    if (lane == 0) partial[item] = r;
}

__global__ void crc_buffer_finish_kernel(const uint8_t *data, uint64_t len, uint32_t chunks,
                                         const uint32_t *partial, uint32_t *out,
                                         const CrcTables *t) {
    uint64_t base = reinterpret_cast<uintptr_t>(data) & 15;
    uint64_t last_bytes = base + len - (uint64_t)(chunks - 1) * kChunkBytes;
    uint32_t r = partial[0];
    if (chunks > 1) {
        for (uint32_t k = 1; k + 1 < chunks; k++) r = gf_mulmod(r, t->chunk_weight) ^ partial[k];
        r = gf_mulmod(r, gf_xpow8(t, last_bytes)) ^ partial[chunks - 1];
    }
    *out = ~r;
}

}  // namespace nisshi

#include "encode.inc"
#include "storage.inc"
#include "fetch.inc"
#include "compact.inc"
#include "api.inc"
#include "pool.inc"
