// MI355X (gfx950) Kafka RecordBatch v2 codec: kernels and the C ABI of
// include/nisshi_hip.h.  Reference behaviour: nisshi-sans-io/src/record.rs,
// src/record/{deflated,inflated,codec,header}.rs, src/primitive/varint.rs.
//
// Decode pipeline (all on one stream):
//   batch_header_kernel   one lane per batch: the 61-byte header, its checks, the record_data span;
//                         a batch's record slots, expected header slots and CRC work items as sums
//                         within its block of 256 batches and one sum a block
//   block_prefix_kernel   running sums over the blocks (only for very many batches or tiny ones)
//   tiny_batch_kernel     calls of batches of a few hundred bytes: one lane per batch
//   decode_fused_kernel   one wave per batch: CRC-32C streamed just ahead of the record walk;
//                         records of one shape 64 at a time, records that change from one to the
//                         next by a chained walk; the CRC chunks of batches of 512 KiB and more by
//                         the workgroups at the front of the grid
//   second_pass_kernel    what the fused kernel listed: large batches, batches whose headers did
//                         not fit their expected slots, the sequential walk for exact error outcomes
#include <hip/hip_runtime.h>
#include <atomic>

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

#ifndef NISSHI_EXACT_INLINE
#define NISSHI_EXACT_INLINE __forceinline__
#endif
#ifndef NISSHI_LARGE_INLINE
#define NISSHI_LARGE_INLINE __forceinline__
#endif

namespace nisshi {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

// Internal per-batch flags
constexpr uint32_t kFlagExact = 1u;   // needs the sequential walk
constexpr uint32_t kFlagMore = 2u;    // more headers than the slots it was expected to need: walked again
constexpr uint32_t kFlagChunks = 8u;  // the batch's chunks are work items of chunk_kernel: its CRC comes from their residues
constexpr uint32_t kFlagSeg = 4u;     // CRC and records chunk by chunk, a wave each (chunk_kernel); decode_fused_kernel checks that the chunks join up
constexpr uint32_t kFlagGap = 16u;    // such a batch whose records are not numbered 0, 1, 2 ...: a chunk's first index is the count of the records before it

struct Work {
    uint32_t *alloc_count;  // record slots the batch gets
    uint32_t *n_chunks;     // CRC work items of the batch
    uint32_t *chunk_base;   // encode: exclusive scan of n_chunks, n + 1 entries; decode: see blk_chunks
    uint32_t *partial;      // one residue per CRC work item
    uint32_t *seg_info;     // four words per CRC work item of a batch decoded chunk by chunk: see segment_wave
    uint32_t *flags;
    // Where a batch's records and headers go follows from sums over the batches
    // before it, in two levels: a batch's place among the kSumBlock batches of its
    // block (rec_local, hdr_local) and a sum a block (blk_sum: record slots in the
    // low half, header slots in the high half).  Header slots are those the batch
    // is expected to need (hdr_pred: the header count of its first record times its
    // record count); a batch that turns out to need more gets all of its headers
    // behind the expected ones of all batches, in batch order (hdr_need, blk_need).
    uint32_t *hdr_pred, *hdr_need, *rec_local, *hdr_local;
    uint64_t *blk_sum;
    uint32_t *blk_need;
    // the CRC work items of the batches of kBigChunks chunks and more, numbered the same way: chunk_base holds
    // a batch's first item among those of its block, blk_chunks the items of a block
    uint32_t *blk_chunks;
    uint32_t *list;         // batches with work left after the streamed walk, in no particular order
    uint64_t *totals;       // [0] record slots [1] header slots [2] failed [3] exact [4] out bytes [5] chunks
                            // [6] end of the expected header slots [7] entries of `list`
    uint64_t *tile_sum;     // the scans over very many batches: two sums a tile of kScanTile batches
    uint32_t *shape_guess;  // kShapeWords words a block of kSumBlock batches: the shape of the first record of the block's first batch (guess_shape)
    uint32_t *chunk_done;   // encode: the CRC work item has its residue already (from the copy of a long span); decode: the records
                            // that start in the chunk under the call's tag, of batches whose chunks count them (kFlagGap)
    // Two words of page-locked host memory that the call leaves for the host's next call: [0] the entries
    // its list had, [1] the CRC work items its large batches have (or would have had).  May be NULL.
    uint32_t *hint;
};

constexpr uint32_t kBigChunks = 8;        // from this many 64 KiB chunks on a batch's CRC is taken chunk by chunk
constexpr uint32_t kSegChunks = 2;        // from this many on, a batch of small records may be decoded chunk by chunk too
constexpr uint32_t kSegRecordMax = 1024;  // "small": the first record has at most this many bytes
constexpr uint32_t kSumBlock = 256;       // batches a block of the two-level sums
constexpr uint32_t kSumDirect = 128;      // up to this many blocks the consumers add the block sums themselves

// ------------------------------------------------------------ batch header

// The header count of a batch's first record (0 when that record does not decode),
// and with `all` the sum over all its records by the sequential walk (0 when any
// of them does not decode): what the batch is expected to need of header slots.
__device__ inline uint32_t expected_headers(const uint8_t *data, uint32_t data_len, uint32_t count, bool all,
                                            uint32_t &first_size) {
    first_size = 0;  // the bytes of the first record, its length varint included; 0: not known
    Reader rd(data, data_len);
    uint32_t pos = 0;
    uint64_t sum = 0;
    uint32_t first = 0;
    for (uint32_t k = 0; k < (all ? count : (count ? 1u : 0u)); k++) {
        RecordFields f;
        if (read_record_head(rd, pos, f)) return 0;
        if ((uint64_t)f.hdr_count * 2 > rd.len - pos) return 0;  // a header is at least two bytes
        if (!all) {
            first = (uint32_t)f.hdr_count;
            if (f.length >= 0) {
                uint32_t z = (uint32_t)f.length << 1;
                first_size = (uint32_t)f.length + (z < (1u << 7) ? 1u : z < (1u << 14) ? 2u : z < (1u << 21) ? 3u : z < (1u << 28) ? 4u : 5u);
            }
            break;
        }
        for (int32_t j = 0; j < f.hdr_count; j++) {
            uint32_t o;
            int32_t n;
            if (rd.octets(pos, o, n) || rd.octets(pos, o, n)) return 0;
        }
        sum += (uint32_t)f.hdr_count;
    }
    if (all) return (uint32_t)sum;
    uint64_t expect = (uint64_t)first * count;
    return expect * 2 <= data_len ? (uint32_t)expect : 0;
}

// Exclusive scan of one 64-bit value a thread over a workgroup of kSumBlock threads.
__device__ inline uint64_t block_exclusive_scan64(uint64_t v, uint64_t *lds, uint64_t &total) {
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    uint64_t offset = 0, sum = 0;
    for (int i = 0; i < (int)(kSumBlock >> 6); i++) {
        uint64_t t = lds[i];
        if (i < wid) offset += t;
        sum += t;
    }
    __syncthreads();  // the sums may be written again
    total = sum;
    return offset + inc - v;
}

// deflated::Batch::try_from(Bytes), deflated.rs:108-213, in the reference's order
// of checks; then Compression::try_from (lib.rs:1685-1698).  One batch of
// batch_header_kernel; returns its record slots (low half) and the header slots
// it is expected to need (high half).
__device__ inline uint64_t batch_header_one(const uint8_t *blob, const uint64_t *batch_off, uint32_t b, int by_value,
                                            const nisshi_columns &c, const Work &w, uint32_t count_small,
                                            uint32_t seg_chunks, bool chunks_on, uint32_t &chunk_items) {
    const uint8_t *p = blob + batch_off[b];
    uint64_t len = batch_off[b + 1] - batch_off[b];
    int st = NISSHI_OK;
    int64_t base_offset = 0, base_timestamp = 0, max_timestamp = 0, producer_id = 0;
    int32_t batch_length = 0, ple = 0, lod = 0, base_sequence = 0;
    int8_t magic = 0;
    uint32_t crc = 0, record_count = 0, data_len = 0;
    int16_t attributes = 0, producer_epoch = 0;
    if (len >= 61) {
        // The whole header is there (all but broken batches): its 61 bytes by sixteen
        // aligned loads that do not wait for one another, shifted into place, then
        // the reference's checks in its order on what they brought.  (Below, for a
        // batch cut short, every read follows the check that allows it: a dozen
        // memory latencies in a row.)
        uintptr_t at = reinterpret_cast<uintptr_t>(p);
        uintptr_t a0 = at & ~uintptr_t(3), last = (at + len - 1) & ~uintptr_t(3);
        typedef __attribute__((address_space(1))) const uint32_t gbl_u32;
        uint32_t d[17], e[16];
#pragma unroll
        for (int i = 0; i < 17; i++) {
            uintptr_t a = a0 + 4 * i <= last ? a0 + 4 * i : last;
            d[i] = *(gbl_u32 *)a;
        }
        // The first record's head is read below, a few bytes at a time, each read
        // waiting for the one before: ask for the lines it will touch now, so that
        // those reads find them in the cache instead of going to memory one after
        // the other (the values loaded here are not used).
        uint32_t touch[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            uintptr_t a = a0 + 64 + 64 * i <= last ? a0 + 64 + 64 * i : last;
            touch[i] = *(gbl_u32 *)a;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) asm volatile("" ::"v"(touch[i]));
#pragma unroll
        for (int i = 0; i < 16; i++) e[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], (uint32_t)(at & 3));
        auto b32 = [&](int off) -> uint32_t {  // big-endian 32 bits at a constant byte offset
            uint32_t v = off & 3 ? __builtin_amdgcn_alignbyte(e[(off >> 2) + 1], e[off >> 2], off & 3) : e[off >> 2];
            return __builtin_bswap32(v);
        };
        auto b64 = [&](int off) -> uint64_t { return ((uint64_t)b32(off) << 32) | b32(off + 4); };
        base_offset = (int64_t)b64(0);
        batch_length = (int32_t)b32(8);
        ple = (int32_t)b32(12);
        magic = (int8_t)(e[4] & 0xff);
        crc = b32(17);
        if (batch_length < 0) {
            st = NISSHI_TRY_FROM_INT;
        } else if (batch_length < 9 || (uint64_t)batch_length - 9 > len - 21) {
            st = NISSHI_OVERFLOW;
        } else {
            attributes = (int16_t)(b32(21) >> 16);
            lod = (int32_t)b32(23);
            base_timestamp = (int64_t)b64(27);
            max_timestamp = (int64_t)b64(35);
            producer_id = (int64_t)b64(43);
            producer_epoch = (int16_t)(b32(51) >> 16);
            base_sequence = (int32_t)b32(53);
            record_count = b32(57);
            int codec = attributes & 7;
            if (batch_length < 49) {
                st = NISSHI_OVERFLOW;
            } else {
                data_len = (uint32_t)batch_length - 49;
                if ((uint64_t)data_len > len - 61) st = NISSHI_OVERFLOW;
                else if (codec > 4) st = NISSHI_UNKNOWN_COMPRESSION;
                else if (codec != 0) st = NISSHI_COMPRESSED;
                else if (by_value) attributes = (int16_t)(attributes & 0x78);
            }
        }
    } else do {
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
    if (c.enc_len) c.enc_len[b] = 0;
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
    // header slots: small batches of a call of small batches are counted exactly
    // (they are walked a lane each, tiny_batch_kernel), the others expected from
    // their first record
    uint32_t expect = 0, first_size = 0;
    if (fits && record_count)
        expect = expected_headers(p + 61, data_len, record_count, count_small && len <= count_small, first_size);
    w.hdr_pred[b] = expect;
    // A large batch of small records is decoded chunk by chunk (segment_wave); a
    // large batch of anything else gets its CRC from its chunks and its records
    // from one wave.  Either way its chunks are work items.
    bool segmented = fits && chunks >= seg_chunks && first_size != 0 && first_size <= kSegRecordMax;
    // A batch that does not number its records 0, 1, 2 ... (a compacted one: last_offset_delta + 1 is not
    // its record count): its chunks' waves count their records before they walk them.
    bool gapped = segmented && (uint32_t)lod + 1 != record_count;
    // (chunks_on false: the call has no launch that works on chunks -- the call before it had no such
    // batch; the items are still counted, for the call after this one, and the batch is one wave's)
    chunk_items = segmented || chunks >= kBigChunks ? chunks : 0;
    if (chunk_items && chunks_on)
        w.flags[b] |= !segmented ? kFlagChunks : gapped ? kFlagSeg | kFlagChunks | kFlagGap : kFlagSeg | kFlagChunks;
    w.hdr_need[b] = 0;
    return ((uint64_t)expect << 32) | (fits ? record_count : 0u);
}

__device__ inline void guess_shape(const uint8_t *blob, const uint64_t *batch_off, uint32_t n, uint32_t blk, const Work &w);

// The workgroups past the blocks' (one a block, launched when w.shape_guess is
// there) each leave a guess at the shape of their block's records: guess_shape.
__global__ __launch_bounds__(kSumBlock) void batch_header_kernel(const uint8_t *blob, const uint64_t *batch_off, uint32_t n,
                                    int by_value, nisshi_columns c, Work w, uint32_t count_small, uint32_t seg_chunks,
                                    uint32_t chunks_on) {
    __shared__ uint64_t wave_sums[kSumBlock >> 6];
    uint32_t n_blocks = (n + kSumBlock - 1) / kSumBlock;
    if (blockIdx.x >= n_blocks) {
        if (threadIdx.x == 0) guess_shape(blob, batch_off, n, blockIdx.x - n_blocks, w);
        return;
    }
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    // the call's totals start at zero: their first users are the kernels after this one
    if (b < 8) w.totals[b] = 0;
    uint64_t mine = 0;  // record slots in the low half, expected header slots in the high half
    uint32_t items = 0;  // CRC work items: those of large batches
    if (b < n) mine = batch_header_one(blob, batch_off, b, by_value, c, w, count_small, seg_chunks, chunks_on != 0, items);
    uint64_t total;
    uint64_t before = block_exclusive_scan64(mine, wave_sums, total);
    if (b < n) {
        w.rec_local[b] = (uint32_t)before;
        w.hdr_local[b] = (uint32_t)(before >> 32);
    }
    // large batches are rare: their items are only scanned where the block has any
    uint32_t items_total = 0, items_before = 0;
    if (__syncthreads_or(items != 0)) {
        uint64_t t64;
        items_before = (uint32_t)block_exclusive_scan64(items, wave_sums, t64);
        items_total = (uint32_t)t64;
    }
    if (b < n) w.chunk_base[b] = items_before;
    if (threadIdx.x == 0) {
        w.blk_sum[blockIdx.x] = total;
        w.blk_need[blockIdx.x] = 0;
        w.blk_chunks[blockIdx.x] = items_total;
    }
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

// The scans over batches of the encoder (encode.inc): a thread takes kScanItems
// consecutive batches, so a workgroup takes kScanTile in a pass.  One workgroup goes
// through all the batches pass by pass (kScanAll); from kScanTiledFrom batches on
// that is too slow (3 us a pass), and a scan is two launches of a workgroup a tile:
// kScanSum leaves the tile's sum in w.tile_sum, kScanApply starts from the sums of
// the tiles before its own.
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

// ------------------------------------------------------------ CRC-32C

// One CRC work item: chunk `chunk` of batch b, by one wave; its residue goes to partial[item].
__device__ __forceinline__ void crc_chunk_one(uint32_t item, uint32_t b, uint32_t chunk, const uint8_t *blob,
                                              const uint64_t *batch_off, const int32_t *batch_length,
                                              uint32_t *crc_out, const Work &w, const CrcTables *t, int lane) {
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

// The CRC work items first, first + stride, ..., one wave each: item -> (batch, chunk)
// by binary search in the scan over all batches (the encoder's numbering).
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
        crc_chunk_one(item, lo_b, item - w.chunk_base[lo_b], blob, batch_off, batch_length, crc_out, w, t, lane);
    }
}

// The number of the first CRC work item of (large) batch b in the decoder's
// two-level numbering; every lane of the wave gets it.
__device__ __forceinline__ uint32_t chunk_first(const Work &w, uint32_t b, bool prefixed, int lane) {
    uint32_t blk = b / kSumBlock;
    uint32_t v = 0;
    if (prefixed) {
        v = w.blk_chunks[blk];
    } else {
        for (uint32_t j = (uint32_t)lane; j < blk; j += 64) v += w.blk_chunks[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    }
    return uni(v + w.chunk_base[b]);
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

// The same by a whole wave (b and chunks the same in every lane): the residues of
// the full chunks 64 at a time, one a lane, each multiplied by its weight within
// the group from the table of W^k (lane 63 is the group's last chunk: W^0) and
// added up over the lanes; a group weighs W^64 against the next; x^(8 * last_bytes)
// from the table a byte of the exponent.  Four multiplications for a batch of up
// to 65 chunks, where the chain above has one a chunk.
__device__ inline uint32_t crc_combine_wave(const uint64_t *batch_off, const int32_t *batch_length,
                                            uint32_t b, uint32_t chunks, const uint32_t *part, const CrcTables *t,
                                            int lane) {
    uint64_t lo = batch_off[b] + 21;
    uint64_t hi = lo + (uint64_t)batch_length[b] - 9;
    uint64_t grid = lo & ~uint64_t(15);
    uint64_t last_bytes = hi - (grid + (uint64_t)(chunks - 1) * kChunkBytes);
    uint32_t m = chunks - 1;  // residues of full chunks, each weighing W against the next
    uint32_t weight = t->chunk_wpow[63 - lane];
    uint32_t r = 0;
    // 64 at a time from the left; the first group is the short one, padded with zeros on its left
    for (uint32_t done = 0; done < m;) {
        uint32_t size = done == 0 && (m & 63) ? (m & 63) : 64;
        uint32_t pad = 64 - size;
        uint32_t v = (uint32_t)lane >= pad ? part[done + (uint32_t)lane - pad] : 0;
        uint32_t g = wave_xor(gf_mulmod(v, weight));
        r = (done ? gf_mulmod(r, t->chunk_pow[6]) : 0u) ^ g;  // W^64
        done += size;
    }
    r = gf_mulmod(r, gf_xpow8(t, last_bytes)) ^ part[chunks - 1];
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

// ---- where a batch's records and headers go
//
// Slots are the same whatever the order the batches are decoded in: a batch's
// records start at the sum of the record slots of the batches before it, its
// headers at the sum of the header slots those batches are expected to need
// (batch_header_kernel), and a batch that needs more than it was expected to gets
// all of its headers behind the expected slots of all batches, again in batch
// order.  So a batch's headers are one run, in record order, and two decodes of a
// blob give the same columns.

// Sum over the blocks before `blk` of the two-level sums (record slots in the low
// half, header slots in the high half); every lane of the wave gets it.
// `prefixed`: block_prefix_kernel has turned the block sums into running sums.
__device__ __forceinline__ uint64_t blocks_before(const Work &w, uint32_t blk, bool prefixed, int lane) {
    if (prefixed) return w.blk_sum[blk];
    uint64_t v = 0;
    for (uint32_t j = (uint32_t)lane; j < blk; j += 64) v += w.blk_sum[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// The first record slot and the first expected header slot of batch b.  `before`:
// the totals of the slice decoded before this one (the slots go on from there), or NULL.
// In two steps, so that a caller can ask for these values together with others and
// wait for all of them once: the loads, then the sums.
struct BasesLoad {
    uint64_t blocks;  // this lane's share of the blocks before (the running sum itself when prefixed)
    uint32_t rec_local, hdr_local, rec_before, hdr_before;
};

__device__ __forceinline__ BasesLoad batch_bases_load(const Work &w, uint32_t b, bool prefixed, const uint64_t *before,
                                                      int lane) {
    BasesLoad q;
    uint32_t blk = b / kSumBlock;
    q.blocks = 0;
    if (prefixed) {
        q.blocks = w.blk_sum[blk];
    } else {
        for (uint32_t j = (uint32_t)lane; j < blk; j += 64) q.blocks += w.blk_sum[j];
    }
    q.rec_local = w.rec_local[b];
    q.hdr_local = w.hdr_local[b];
    q.rec_before = before ? (uint32_t)before[0] : 0u;
    q.hdr_before = before ? (uint32_t)before[1] : 0u;
    return q;
}

__device__ __forceinline__ void batch_bases_sum(const BasesLoad &q, bool prefixed, uint32_t &rec_base, uint32_t &hdr_base) {
    uint64_t sum = q.blocks;
    if (!prefixed) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    }
    rec_base = uni((uint32_t)sum + q.rec_local + q.rec_before);
    hdr_base = uni((uint32_t)(sum >> 32) + q.hdr_local + q.hdr_before);
}

__device__ __forceinline__ void batch_bases(const Work &w, uint32_t b, bool prefixed, const uint64_t *before, int lane,
                                            uint32_t &rec_base, uint32_t &hdr_base) {
    BasesLoad q = batch_bases_load(w, b, prefixed, before, lane);
    batch_bases_sum(q, prefixed, rec_base, hdr_base);
}

// The first header slot of a batch whose headers did not fit its expected slots:
// behind the expected slots of all batches (totals[6]), after the needs of the
// batches before it.  Read only in kernels after the one that found the needs.
__device__ __forceinline__ uint32_t extra_base(const Work &w, uint32_t b, int lane) {
    uint32_t blk = b / kSumBlock;
    uint32_t v = 0;
    for (uint32_t j = (uint32_t)lane; j < blk; j += 64) v += w.blk_need[j];
    for (uint32_t i = blk * kSumBlock + (uint32_t)lane; i < b; i += 64) v += w.hdr_need[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    return uni(v + (uint32_t)w.totals[6]);
}

// Turns the block sums into running sums (one workgroup): from kSumDirect blocks
// on, and for calls of small batches, whose kernels take a lane a batch.
__global__ __launch_bounds__(1024) void block_prefix_kernel(uint32_t n_blocks, Work w) {
    __shared__ uint64_t wave_sums[16];
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t run = 0;
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < n_blocks ? w.blk_sum[i] : 0, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_sums[wid] = inc;
        __syncthreads();
        uint64_t offset = 0, total = 0;
        for (int q = 0; q < 16; q++) {
            uint64_t t = wave_sums[q];
            if (q < wid) offset += t;
            total += t;
        }
        __syncthreads();
        if (i < n_blocks) w.blk_sum[i] = run + offset + inc - v;
        run += total;
    }
    // the CRC work items of large batches the same way, their total behind the last block
    __syncthreads();
    uint32_t crun = 0;
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < n_blocks ? w.blk_chunks[i] : 0, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_sums[wid] = inc;
        __syncthreads();
        uint64_t offset = 0, total = 0;
        for (int q = 0; q < 16; q++) {
            uint64_t t = wave_sums[q];
            if (q < wid) offset += t;
            total += t;
        }
        __syncthreads();
        if (i < n_blocks) w.blk_chunks[i] = crun + (uint32_t)(offset + inc - v);
        crun += (uint32_t)total;
    }
    if (threadIdx.x == 0) w.blk_chunks[n_blocks] = crun;
}

// The header slots of one batch while it is walked.
struct HeaderSlots {
    uint32_t first, next, limit;  // the batch's slots are [first, limit); next: the first one not taken
    uint64_t cap;                 // slots the header columns have
    uint32_t count;               // headers of the batch so far
    bool over;                    // more headers than slots: none is stored from here on, the walk only counts
    bool full;                    // the header columns end before the batch's slots do

    __device__ __forceinline__ void init(uint32_t first_, uint32_t slots, uint64_t cap_) {
        first = next = first_;
        limit = first_ + slots;
        cap = cap_;
        count = 0;
        over = full = false;
    }

    __device__ __forceinline__ bool advance(uint32_t sum) {
        count += sum;
        if (over || full) return false;
        if (sum > limit - next) {
            over = true;
            return false;
        }
        if ((uint64_t)next + sum > cap) {
            full = true;
            return false;
        }
        next += sum;
        return true;
    }

    // Slots for a round of m records from lane 0 up; the lane's record has hc
    // headers (0 in lanes without a record).  `slot`: the lane's first.  False:
    // the round's headers are not to be stored (over, or full).
    __device__ __forceinline__ bool take(uint32_t hc, uint32_t m, int lane, uint32_t &slot) {
        uint32_t hc0 = uni(hc);
        bool live = (uint32_t)lane < m;
        // equal header counts in every record (the usual case) need no scan
        uint32_t inc = __all(hc == (live ? hc0 : 0u)) ? (live ? ((uint32_t)lane + 1) * hc0 : m * hc0)
                                                       : wave_inclusive_scan(hc, lane);
        uint32_t sum = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        slot = next + inc - hc;
        return advance(sum);
    }

    // The same for m records of hc headers each.
    __device__ __forceinline__ bool take_uniform(uint32_t hc, uint32_t m, int lane, uint32_t &slot) {
        slot = next + (uint32_t)lane * hc;
        return advance(m * hc);
    }
};

// How a wave's walk of a batch ended.
constexpr int kWalkDone = 0;
constexpr int kWalkExact = 1;   // a record that does not decode, or whose length disagrees: exact_walk
constexpr int kWalkChain = 3;   // records whose sizes or shapes change from one to the next: walk_chain
constexpr int kWalkFull = 4;    // the walk of a piece of a batch only: the header slots ran out

// ---- the streamed walk
//
// The shape of a record is everything in it but the payload bytes and the values
// of attributes, timestamp_delta and offset_delta: the bytes of every length and
// count varint and the widths of the two deltas.  Records produced by one client
// mostly share a shape, and a record of a known shape needs no varint walk: its
// fields sit at known offsets.  The wave learns the shape of a record and tests the
// records that follow against it, 64 at a time.
//
// A shape has two parts.  The head -- the length varint, the attributes byte and
// the two deltas -- is held as a few scalars (the bytes of the length, the widths
// of the deltas) and tested by arithmetic on what a lane reads there anyway.  The
// rest, from the key's length on, is a list of masked 32-bit compares (kept in LDS,
// as are the header spans) whose offsets count from the end of the offset delta.
// So when the deltas of a batch's records grow a byte (record 64, record 8192 of
// every batch a producer numbers 0, 1, 2, ...) and the length with them, the shape
// is re-based from the first such record by changing the scalars alone
// (walk_streamed), and nothing is learnt again.
constexpr uint32_t kShapeChecks = 24;  // a multiple of four
constexpr uint32_t kShapeHdrs = 16;
constexpr uint32_t kShapeWords = (kShapeChecks + kShapeHdrs + 3) * 4;  // and the learnt scalars

struct Shape {
    uint32_t n_checks;  // 0: none learnt
    uint32_t size;      // bytes of a record of this shape, its length varint included
    uint32_t n_len;     // bytes of the length varint
    int32_t length;     // its value
    uint32_t lenpat;    // its bytes (little-endian, the n_len low bytes)
    uint32_t wts, wod;  // bytes of the timestamp delta and of the offset delta
    uint32_t o_fix;     // n_len + 1 + wts + wod: where the key's length starts; the table's offsets count from here
    uint32_t o_key, o_val;  // offsets from the start of the record
    int32_t klen, vlen;
    uint32_t hc;
    bool minimal;       // every length and count varint from the key's on is written the shortest way
    __device__ __forceinline__ bool len_minimal() const {  // and so is the record's length
        uint32_t z = ((uint32_t)length << 1) ^ (uint32_t)(length >> 31);
        return n_len == (38 - (uint32_t)__clz(z | 1u)) / 7;
    }

    __device__ __forceinline__ uint32_t lenmask() const { return n_len >= 4 ? 0xffffffffu : ~(0xffffffffu << (8 * n_len)); }
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
// or does not end where its length says, kWalkChain when the compares cannot
// express it (a varint of five bytes or more, too many fields).
//
// The fields are walked by one loop: 0 length, 1 timestamp delta (after the
// attributes byte), 2 offset delta, 3 key, 4 value, 5 header count, then a key
// and a value per header.  Of the first three the scalars of the head are noted;
// the bytes of every varint from the key's length on go into the compares, at
// offsets from where that length starts, and the spans of key, value and headers
// are noted from there too.
//
// Tab: where the table goes (LDS, or global memory for a guess handed to other
// waves); Src: where the record's bytes come from, read64(position).  kGuard: the
// source may not be read at or past the record's end (global memory).
template <bool kGuard, bool kMinimal, class Tab, class Src>
__device__ __forceinline__ int learn_shape_any(Tab tab, Src &ring, uint32_t at0, uint32_t size) {
    Tab htab = tab + kShapeChecks;
    Tab out = htab + kShapeHdrs;
    uint32_t p = 0;  // offset in the record
    uint32_t cur_off = 0, cur_mask = 0, cur_pat = 0, noted = 0;
    uint32_t fields = 6;
    uint32_t n_len = 0, wts = 0, wod = 0, o_fix = 0, lenpat = 0, o_key = 0, o_val = 0, hc = 0, ko = 0;
    int32_t length = 0, klen = 0, vlen = 0, kl = 0;
    bool wide = false, bad = false, minimal = true;
    for (uint32_t i = 0; i < fields && !bad; i++) {
        if (i == 1) p += 1;
        if (kGuard && p >= size) {
            bad = true;
            break;
        }
        uint64_t w = ring.read64(at0 + p);
        uint32_t used;
        int32_t v;
        bad = !fast_varint(w, used, v);
        wide = wide || used > 4;
        uint32_t bytes = used >= 4 ? 0xffffffffu : ~(0xffffffffu << (8 * used));
        // written the shortest way?  (the length and the two deltas are the head's: see walk_streamed)
        if (kMinimal && i > 2) {
            uint32_t z = ((uint32_t)v << 1) ^ (uint32_t)(v >> 31);
            minimal = minimal && used == (38 - (uint32_t)__clz(z | 1u)) / 7;
        }
        if (i < 3) {  // the head: scalars
            if (i == 0) {
                n_len = used;
                length = v;
                lenpat = (uint32_t)w & bytes;
            } else if (i == 1) {
                wts = used;
            } else {
                wod = used;
            }
            p += used;
            o_fix = p;
            continue;
        }
        uint32_t q = p - o_fix;  // the varint's offset from the end of the head
        // into the compare in hand if it fits, else start the next
        if (q + used > cur_off + 4) {
            if (noted < kShapeChecks) tab[noted] = u32x4{cur_off, cur_mask, cur_pat, 0};
            noted++;
            cur_off = q;
            cur_mask = 0;
            cur_pat = 0;
        }
        uint32_t sh = (q - cur_off) * 8;
        cur_mask |= bytes << sh;
        cur_pat |= ((uint32_t)w & bytes) << sh;
        p += used;
        if (i == 5) {
            // a header is at least two bytes: a count the record cannot hold is an error
            bad = bad || v < 0 || p > size || (uint32_t)v > (size - p) / 2;
            hc = bad ? 0 : (uint32_t)v;
            fields = 6 + 2 * (hc <= kShapeHdrs ? hc : 0);
        } else {
            // Octets::decode: -1 is None, any other negative length or one past the record is an error
            bad = bad || v < -1 || p > size || (uint32_t)(v < 0 ? 0 : v) > size - p;
            uint32_t off = p - o_fix;
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
    if (wide || hc > kShapeHdrs || noted > kShapeChecks) st = kWalkChain;
    // the fields must end exactly where the length said
    else if (bad || p != size) st = kWalkExact;
    out[0] = u32x4{(noted + 3) & ~3u, n_len | (wts << 8) | (wod << 16), (uint32_t)length, lenpat};
    out[1] = u32x4{o_key, o_val, (uint32_t)klen, (uint32_t)vlen};
    out[2] = u32x4{hc, (uint32_t)st, size, minimal ? 1u : 0u};
    return st;
}

template <int kRows>
__device__ __forceinline__ void learn_shape(lds_u128 *tab, RingView<kRows> ring, uint32_t at0, uint32_t size) {
    learn_shape_any<false, kRows != 8>(tab, ring, at0, size);  // (the kernel of small batches does not ask how the varints are written)
}

// A batch's record data in global memory as a source of learn_shape_any.
struct GlobalBytes {
    Reader rd;
    __device__ __forceinline__ uint64_t read64(uint32_t at) { return rd.get64(at); }
};

// The shape of the first record of the first batch of block `blk`, left in global
// memory as a first guess for the waves that walk the block's batches: batches that
// lie side by side mostly come from one producer and share a shape, and a wave that
// finds its batch's records fitting the guess need not learn the shape itself.  (A
// guess is only ever tested against records, as a learnt shape is: a wrong or
// missing guess costs the wave its own learning and nothing else.)  By one lane.
__device__ inline void guess_shape(const uint8_t *blob, const uint64_t *batch_off, uint32_t n, uint32_t blk, const Work &w) {
    u32x4 *tab = reinterpret_cast<u32x4 *>(w.shape_guess) + (uint64_t)blk * (kShapeWords / 4);
    uint32_t b = blk * kSumBlock;
    int st = kWalkExact;
    if (b < n) {
        const uint8_t *p = blob + batch_off[b];
        uint64_t len = batch_off[b + 1] - batch_off[b];
        if (len > 61) {
            int32_t batch_length = (int32_t)be32(p + 8);
            uint32_t codec = be16(p + 21) & 7u, count = be32(p + 57);
            if (batch_length > 49 && (uint64_t)batch_length - 49 <= len - 61 && codec == 0 && count &&
                (uint64_t)count * 7 <= (uint64_t)batch_length - 49) {
                uint32_t data_len = (uint32_t)batch_length - 49;
                GlobalBytes src{Reader(p + 61, data_len)};
                uint32_t used;
                int32_t length;
                if (fast_varint(src.read64(0), used, length) && length >= 0 && (uint64_t)used + (uint32_t)length <= data_len)
                    st = learn_shape_any<true, true>(tab, src, 0, used + (uint32_t)length);
            }
        }
    }
    if (st != kWalkDone) tab[kShapeChecks + kShapeHdrs] = u32x4{0, 0, 0, 0};  // no compares: no guess
}

// The record walk of one batch by the wave that streams it.  Each round pulls
// the rows it is about to parse through the CRC into the LDS ring, finds up to 64
// record starts by speculation (lane i assumes that the i-th next record starts
// i * S bytes on and decodes the length varint there; lanes before the first one
// whose size differs from S did start on a record), tests those records against
// the shape in hand, learning the first one's shape when it does not fit, and
// emits the records that fit.  On kWalkChain *k_out and *pos_out say where the
// chained walk goes on.
//
// The walk starts at record *k_out, byte *pos_out.  `seg` (the same in every lane):
// the walk of one piece of a batch, which takes the records that start below
// pos_end and leaves the batch's status alone.
template <int kRows, bool kFold>
__device__ __forceinline__ int walk_streamed(uint32_t b, const nisshi_columns &c, uint32_t rec_base, HeaderSlots &slots,
                                             BatchStream<kRows, kFold> &stream, lds_u128 *tab, int lane,
                                             uint32_t *k_out, uint32_t *pos_out, bool seg = false,
                                             uint32_t pos_end = 0xffffffffu, bool guessed = false,
                                             bool *minimal_out = nullptr) {
    uint32_t count = uni(c.record_count[b]);
    uint64_t rbase = rec_base;
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
    // Whether the walk's records are all written the shortest way, for the size the encoder
    // can then take on trust: kept track of by the kernel of large batches only (the
    // twelve-row ring).  In the kernel of small batches, which is the one that is short of
    // registers, it costs the walk 4 % and saves the encoder a read of a hundred records.
    constexpr bool kSized = kRows != 8;
    bool all_minimal = true;  // every shape so far has its lengths and counts written the shortest way
    // the shape whose table lies in `tab` (learn_shape_any has just left it there, or it is a guess)
    auto load_shape = [&]() -> int {
        const lds_u128 *in4 = htab + kShapeHdrs;
        u32x4 a = in4[0], b4 = in4[1], c4 = in4[2];
        int learnt = (int)uni(c4.y);
        shape.n_checks = learnt == kWalkDone ? uni(a.x) : 0;
        shape.size = uni(c4.z);
        uint32_t widths = uni(a.y);
        shape.n_len = widths & 0xffu;
        shape.wts = (widths >> 8) & 0xffu;
        shape.wod = (widths >> 16) & 0xffu;
        shape.o_fix = shape.n_len + 1 + shape.wts + shape.wod;
        shape.length = (int32_t)uni(a.z);
        shape.lenpat = uni(a.w);
        shape.o_key = shape.o_fix + uni(b4.x);
        shape.o_val = shape.o_fix + uni(b4.y);
        shape.klen = (int32_t)uni(b4.z);
        shape.vlen = (int32_t)uni(b4.w);
        shape.hc = uni(c4.x);
        shape.minimal = uni(c4.w) != 0;
        if (kSized) all_minimal = all_minimal && shape.minimal && shape.len_minimal();
        return learnt;
    };
    // `tab` holds a shape already (guess_shape's, of a batch nearby): the walk starts
    // with it in hand as if it had learnt it from a record before
    if (guessed) load_shape();
    // The stores of a round are issued in the next round, after that round's rows
    // have been taken from the look-ahead queue and before the queue is refilled:
    // the wait for the queued loads then never waits for stores just issued, and
    // every load has a whole round to arrive in.  What is kept per lane is the
    // record's start, its three free fields and its header slot; the rest follows
    // from the shape, which changes only after the flush.
    struct {
        bool any, headers;  // headers: the round's headers have slots
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
        if (shape.hc == 0 || !late.headers) return;
        // the round's headers are one run of slots from `piece` on
        uint32_t o = (late.slot - late.piece) * 16;
        for (uint32_t j = 0; j < shape.hc; j++, o += 16) {
            u32x4 e = htab[j];
            if (live)
                put(reinterpret_cast<u32x4 *>(c.hdr + late.piece), o,
                    u32x4{late.start + shape.o_fix + e.x, e.y, late.start + shape.o_fix + e.z, e.w});
        }
    };
    // How many of the first m lanes' records, counted from lane 0, have the shape in
    // hand.  The head by arithmetic: the bytes of the length, and the widths of the
    // two deltas, whose values (and the attributes byte) the lane keeps for the
    // records that fit; the rest by the table's compares.
    // (they go straight into `late`: the round before was flushed before any test of this one)
    // (kSized) A delta of several bytes whose last byte is zero (not written the shortest way)
    // fits no shape: no client writes one, and the sequential walk takes the batch.  So the
    // records of a walk that ends here are written the shortest way if their shapes are.
    auto fitting = [&](uint32_t at, uint32_t m) -> uint32_t {
        if (shape.n_checks == 0) return 0;
        uint32_t diff = 0;
        uint32_t at2 = at + shape.o_fix;
        // four compares at a time so that the ring reads overlap (the table is
        // padded with compares that always hold)
        for (uint32_t s = 0; s < shape.n_checks; s += 4) {
            u32x4 e0 = tab[s], e1 = tab[s + 1], e2 = tab[s + 2], e3 = tab[s + 3];
            uint32_t w0 = ring.read32(at2 + e0.x), w1 = ring.read32(at2 + e1.x);
            uint32_t w2 = ring.read32(at2 + e2.x), w3 = ring.read32(at2 + e3.x);
            diff |= ((w0 ^ e0.z) & e0.y) | ((w1 ^ e1.z) & e1.y) | ((w2 ^ e2.z) & e2.y) | ((w3 ^ e3.z) & e3.y);
        }
        // (the head last: the values it leaves in `late` are then not held across the compares)
        diff |= (ring.read32(at) ^ shape.lenpat) & shape.lenmask();
        {
            uint32_t w_used;
            int64_t ts;
            uint64_t r1 = ring.read40(RingCfg<kRows>::wrap(at + shape.n_len));  // attributes, then the timestamp delta
            late.attributes = (uint32_t)r1 & 0xffu;
            bool ok = fast_varlong((uint32_t)(r1 >> 8), w_used, ts);  // a shape's timestamp delta has at most four bytes
            late.ts = (int32_t)ts;
            diff |= w_used ^ shape.wts;
            uint32_t o_used;
            uint32_t od_word = ring.read32(at + shape.n_len + 1 + shape.wts);
            (void)fast_varint((uint64_t)od_word, o_used, late.od);  // (a fifth byte: another width)
            diff |= o_used ^ shape.wod;
            if (kSized && shape.wts > 1) ok = ok && ((uint32_t)(r1 >> (8 * shape.wts)) & 0xffu) != 0;
            if (kSized && shape.wod > 1) ok = ok && ((od_word >> (8 * (shape.wod - 1))) & 0xffu) != 0;
            diff |= ok ? 0u : 1u;
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
    uint32_t pos = uni(*pos_out), k = uni(*k_out);
    uint32_t S = count ? data_len / count : 0;
    uint32_t cap = reach(S);
    int st = kWalkDone;
    // rounds in a row that found fewer than kChainBelow of the records they looked
    // for: after two the sizes are taken to change from record to record, which is
    // the chained walk's case
    constexpr uint32_t kChainBelow = 8;
    uint32_t strikes = 0;
#ifdef NISSHI_PHASES
    uint64_t ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
    while (k < count && pos < pos_end) {
        pos = uni(pos);
        S = uni(S);
        k = uni(k);
        cap = uni(cap);
        if (S > kReach) {  // records the ring cannot hold: the chained walk reads them from global memory
            st = kWalkChain;
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
        bool deferred = false;  // the round only found a new size, whose rows are not in the ring yet
        // The usual round: the records have the size of the shape in hand and lie
        // inside the data, so testing them against the shape finds them; no
        // separate look at their lengths is needed (the shape holds the bytes of
        // the length).
        if (shape.n_checks && shape.size == S && (uint64_t)pos + (uint64_t)lanes * S <= data_len)
            m = fitting(at, lanes);
        PHASE(8)
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
                uint64_t len_bytes = ring.read64(at);
                bool ok = in && fast_varint(len_bytes, used, len);
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
                    if (kSized) deferred = !again;  // (the kernel of small batches has no register for it)
                } else {
                    // ---- which of them have the shape in hand; learn the first one's if it has not
                    PHASE(9)
                    m = fitting(at, same_size);
                    PHASE(10)
                    if (m == 0 && shape.n_checks && uni(used) <= 4) {
                        // The record at `pos` may differ from the shape only in its head: deltas a
                        // byte wider (or narrower) and a length that grew by as much.  Then the rest
                        // of the shape stands, that much further on: take the head's scalars from
                        // this record and test again.  (A shape that does not fit after that is
                        // learnt below like any other: nothing is taken on trust here either.)
                        uint32_t n_len = uni(used);
                        int32_t length = (int32_t)uni((uint32_t)len);
                        uint32_t t_used, o_used;
                        int64_t ts;
                        int32_t od;
                        uint64_t r1 = ring.read40(RingCfg<kRows>::wrap(at + n_len));
                        bool head = fast_varlong((uint32_t)(r1 >> 8), t_used, ts);
                        head = fast_varint((uint64_t)ring.read32(at + n_len + 1 + t_used), o_used, od) && head && o_used <= 4;
                        uint32_t wts = uni(t_used), wod = uni(o_used);
                        int32_t grown = (int32_t)(wts + wod) - (int32_t)(shape.wts + shape.wod);
                        if (uni((uint32_t)head) && length == shape.length + grown) {
                            uint32_t moved = (uint32_t)grown + n_len - shape.n_len;  // (wraps when the head shrank)
                            shape.n_len = n_len;
                            shape.length = length;
                            shape.wts = wts;
                            shape.wod = wod;
                            shape.o_fix += moved;
                            shape.o_key += moved;
                            shape.o_val += moved;
                            shape.size = S;
                            shape.lenpat = uni((uint32_t)len_bytes) & shape.lenmask();
                            if (kSized) all_minimal = all_minimal && shape.len_minimal();
                            m = fitting(at, same_size);
                        }
                    }
                    PHASE(11)
                    if (m == 0) {
                        if (lane == 0) learn_shape(tab, ring, pos + bias, S);
                        lds_order();
                        st = load_shape();
                        shape.size = S;
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
        if (seg && m) {  // only the records that start inside the piece
            uint64_t inside = __ballot((uint32_t)lane < m && start < pos_end);
            m = (uint32_t)__builtin_popcountll(inside);
        }
        // (a round that looked for no records yet says nothing about them)
        if (!deferred) strikes = m < kChainBelow && lanes >= 2 * kChainBelow ? strikes + 1 : 0;
        if (strikes >= 2) {
            st = kWalkChain;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // as at the end of the pass
            break;
        }
        PHASE(5)
        if (m) {
            // ---- (the free fields of the records that fit were read when they were tested)
            late.slot = slots.next;  // (a record without headers notes where its headers would have started, as every walk does)
            late.piece = slots.next;
            late.headers = shape.hc ? slots.take_uniform(shape.hc, m, lane, late.slot) : false;
            if (slots.full) {
                if (seg)
                    st = kWalkFull;
                else if (lane == 0)
                    c.status[b] = NISSHI_CAPACITY;
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
        printf("b=%u other=%lu pull=%lu load=%lu flush=%lu spec=%lu fit=%lu emit=%lu take=%lu | fit1=%lu len=%lu fit2=%lu rebase=%lu\n", b, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], ph[8], ph[9], ph[10], ph[11]);
#endif
    // a shape that failed to be learnt has overwritten nothing the flush needs:
    // learning follows the flush of its round
    flush();
    *k_out = k;
    *pos_out = pos;
#ifdef NISSHI_DIAG_MIN
    if (lane == 0 && kSized && (!all_minimal || st != kWalkDone))
        printf("walk b=%u seg=%d k=%u pos=%u st=%d all=%d shape.min=%d len_min=%d n_len=%u length=%d wts=%u wod=%u\n", b, (int)seg, k, pos, st,
               (int)all_minimal, (int)shape.minimal, (int)shape.len_minimal(), shape.n_len, shape.length, shape.wts, shape.wod);
#endif
    // (a walk handed on to the chained walk: that one goes on keeping track)
    if (kSized && minimal_out) *minimal_out = all_minimal && (st == kWalkDone || st == kWalkChain);
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
// A record larger than what the ring holds is a round of its own: its fields are
// read from global memory by one lane, in sequence.
// Starts at record *k_io, byte *pos_io of the record data and leaves there where it
// stopped.  kWalkExact: a record does not decode as its length says.

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

// One record at byte `pos` of batch b's record data, read from global memory:
// Record::decode (record.rs:254-277) by every lane alike, stored by lane 0 as
// record `slot_r`.  For records the LDS ring cannot hold.  Returns 0 and the
// record's end, 1 when it does not decode or does not end where its length says
// (the sequential walk decides), 2 when the header columns are full.
__device__ NISSHI_LARGE_INLINE int large_record(const uint8_t *blob, uint32_t b, const nisshi_columns &c,
                                                      uint64_t slot_r, uint32_t pos, uint32_t data_len,
                                                      HeaderSlots &slots, int lane, uint32_t *end_out) {
    Reader rd(blob + uni64(c.data_off[b]), data_len);
    uint32_t q0 = pos;
    RecordFields f;
    int bad = read_record_head(rd, q0, f);
    uint32_t len_bytes = 0;
    {
        uint32_t t = pos;
        int32_t unused;
        bad = bad || rd.varint(t, unused);
        len_bytes = t - pos;
    }
    uint32_t end = pos + len_bytes + (uint32_t)f.length;  // inside the data: the hops checked it
    bad = bad || (uint64_t)(uint32_t)f.hdr_count * 2 > data_len - q0;
    uint32_t slot = 0;
    bool keep = false;
    if (!bad) keep = slots.take(lane == 0 ? (uint32_t)f.hdr_count : 0u, 1, lane, slot);
    slot = uni(slot);
    for (int32_t j = 0; !bad && j < f.hdr_count; j++) {
        uint32_t ko, vo;
        int32_t kl, vl;
        bad = rd.octets(q0, ko, kl) || rd.octets(q0, vo, vl);
        if (!bad && keep && lane == 0) store_header(c, slot + (uint32_t)j, ko, kl, vo, vl);
    }
    if (bad || q0 != end) return 1;
    if (slots.full) return 2;
    if (lane == 0) store_record(c, slot_r, 0, f, slot);
    *end_out = end;
    return 0;
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

// The record starts of one round of the chained walk found by sixteen chains of hops side by side
// instead of one after the other.  One chain of hops is a chain of dependent LDS reads: 240 cycles a
// record, 64 records a round, half of what a wave spends on such a batch.  Here the round's window is
// cut into sixteen stretches; the first chain starts at `pos`, which is a record's start, and each of
// the others at a *candidate* near the beginning of its stretch: the first byte position there that
// looks like the head of a record (a length varint of two bytes that is not negative, and an
// attributes byte of zero, found by four lanes that test 48 positions each, four at a time in a
// register).  All chains hop in step, every lane its own (the four lanes of a chain the same one), a
// chain noting the starts it passes in a list in LDS.  A chain ends when it lands exactly on the start
// of a chain ahead of it: from there on that chain's hops are the very hops it would have made
// itself.  A chain that passes a start without landing on it leaves that chain out (its candidate was
// no record's start, or is not reached) and aims at the next.  The round's records are then the
// lists of the chains that were landed on, one after the other from the first: the same starts the
// single chain finds, by construction -- nothing of a chain is used unless hops from `pos` led to
// where it started.  `scratch`: 544 bytes of LDS (the shape table, which the chained walk does not
// use).  Leaves the number of records found (at most `want`), where the next one starts, and lane i's
// start; false when the round is not worth it (a short window), nothing done.
constexpr uint32_t kAbreast = 16, kAbreastCap = 15, kAbreastScan = 12;  // chains, starts a chain, dwords a lane scans
template <int kRows>
__device__ __forceinline__ bool chains_abreast(const RingView<kRows> &ring, lds_u32 *scratch, uint32_t bias, uint32_t pos,
                                               uint32_t limit, uint32_t want, uint32_t mean, int lane, uint32_t &n_out,
                                               uint32_t &p_out, uint32_t &my_start) {
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    constexpr uint32_t kRing = RingCfg<kRows>::kRingBytes;
    constexpr uint32_t kScanBytes = kAbreastScan * 4, kQuad = 64 / kAbreast;
    lds_u16 *list = (lds_u16 *)scratch;                 // [kAbreast][kAbreastCap + 1]: starts less `pos`, then the chain's end
    lds_u16 *starts = list + kAbreast * (kAbreastCap + 1);  // [kAbreast]: where each chain starts, less `pos`
    uint32_t span = limit - pos, expect = want * mean + mean;
    span = expect < span ? expect : span;
    uint32_t stride = span / kAbreast;
    if (stride < kQuad * kScanBytes + 32) return false;
    uint32_t c = (uint32_t)lane / kQuad, sub = (uint32_t)lane % kQuad;
#ifdef NISSHI_PHASES
    uint64_t t0 = __builtin_readcyclecounter();
#endif
    // ---- the candidates
    uint32_t start = 0xffffffffu;  // the chain's start as a position in the record data
    {
        uint32_t q0 = ((pos + c * stride + bias) & ~15u) + sub * kScanBytes;  // in the stream's bytes: the ring's grid
        const lds_u32 *rp = ring.ring + (RingCfg<kRows>::wrap(q0) >> 2);
        uint32_t d[kAbreastScan + 1];
#pragma unroll
        for (uint32_t j = 0; j < kAbreastScan; j += 4) {
            u32x4 v = *reinterpret_cast<const lds_u128 *>(rp + j);
            d[j] = v.x, d[j + 1] = v.y, d[j + 2] = v.z, d[j + 3] = v.w;
        }
        d[kAbreastScan] = rp[kAbreastScan];
        uint32_t first = 0xffffu;
#pragma unroll
        for (int j = (int)kAbreastScan - 1; j >= 0; j--) {
            uint32_t w = d[j], nx = d[j + 1];
            uint32_t b1 = __builtin_amdgcn_alignbit(nx, w, 8), b2 = __builtin_amdgcn_alignbit(nx, w, 16);
            uint32_t nz = ((b2 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | b2;  // bit 7 of a byte: the byte two on is not zero
            // the byte continues, is even (no negative length), the next ends the varint, the one after is zero
            uint32_t cand = w & ~(w << 7) & ~b1 & ~nz & 0x80808080u;
            uint32_t at = (uint32_t)j * 4 + ((uint32_t)__builtin_ctz(cand | 0x80000000u) >> 3);
            first = cand ? at : first;
        }
        uint32_t found = first != 0xffffu ? q0 + first - bias : 0xffffffffu;
        uint32_t o = (uint32_t)__shfl_xor((int)found, 1, 64);
        found = o < found ? o : found;
        o = (uint32_t)__shfl_xor((int)found, 2, 64);
        found = o < found ? o : found;
        start = c == 0 ? pos : found;
        if (c && start != 0xffffffffu && start <= pos) start = 0xffffffffu;  // (never at or before `pos`)
        if (start != 0xffffffffu && start + 7 > limit) start = 0xffffffffu;
    }
    // which chains there are, a bit a chain
    uint64_t quads = __ballot(start != 0xffffffffu) & 0x1111111111111111ull;
    quads = (quads | (quads >> 3)) & 0x0303030303030303ull;
    quads = (quads | (quads >> 6)) & 0x000f000f000f000full;
    quads = (quads | (quads >> 12)) & 0x000000ff000000ffull;
    uint32_t exist = (uint32_t)(quads | (quads >> 24)) & 0xffffu;
    if (sub == 0) starts[c] = (uint16_t)(start - pos);
    lds_order();
    auto chain_after = [&](uint32_t t) -> uint32_t {  // the next chain there is after chain t; kAbreast: none
        uint32_t ahead = exist & ~((2u << t) - 1u);
        return ahead ? (uint32_t)__builtin_ctz(ahead) : kAbreast;
    };
#ifdef NISSHI_PHASES
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t steps_done = 0;
#endif
    // ---- the hops
    bool alive = start != 0xffffffffu, landed = false;
    uint32_t pc = alive ? start : pos, roc = RingCfg<kRows>::wrap(pc + bias), cnt = 0;
    uint32_t tgt = chain_after(c);
    uint32_t tpos = tgt < kAbreast ? pos + starts[tgt] : 0xffffffffu;
#pragma unroll 1
    for (uint32_t step = 0; step < kAbreastCap; step++) {
        bool act = alive && !landed;
        if (!__any(act)) break;
#ifdef NISSHI_PHASES
        steps_done++;
#endif
        uint32_t used;
        int32_t len;
        bool ok = fast_varint(ring.read40(roc), used, len) && len >= 0;
        uint32_t size = used + (uint32_t)len;
        bool take = act && ok && size <= limit - pc;
        if (act && !take) alive = false;  // the chain stops here: no record, or one that ends beyond the round
        if (take) {
            list[c * (kAbreastCap + 1) + cnt] = (uint16_t)(pc - pos);
            cnt++;
            pc += size;
            roc += size;
            roc = roc >= kRing ? roc - kRing : roc;
#pragma unroll 1
            for (uint32_t t = 0; t < kAbreast && pc > tpos; t++) {  // passed a start without landing on it
                tgt = chain_after(tgt);
                tpos = tgt < kAbreast ? pos + starts[tgt] : 0xffffffffu;
            }
            landed = pc == tpos;
            // (the last chain has nothing to land on: it ends where the round is expected to)
            if (tgt >= kAbreast && pc >= pos + span) alive = false;
        }
    }
    list[c * (kAbreastCap + 1) + cnt] = (uint16_t)(pc - pos);  // where the chain ended
    lds_order();
#ifdef NISSHI_PHASES
    uint64_t t2 = __builtin_readcyclecounter();
#endif
    // ---- the chains that were landed on, in order.  What the sixteen chains ended with is read
    // into scalar registers (sixteen reads of a lane at a fixed place), the path from the first chain
    // is followed there, and every lane finds its chain by comparing its number with where the
    // chains' records begin in the round.
    uint32_t ended = cnt | (tgt << 8) | (landed ? 0x10000u : 0u);
    uint32_t on_path = 1, off = 0, end_chain = 0, end_at = 0;
    bool open = true;  // the round's end has not been found yet
    uint32_t chain_of = 0, base_of = 0;
#pragma unroll
    for (uint32_t j = 0; j < kAbreast; j++) {
        uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)ended, (int)(j * kQuad));
        bool mine = open && (on_path >> j & 1u);
        uint32_t have = e & 0xffu, room = want - off;
        uint32_t some = have < room ? have : room;
        uint32_t from = mine ? off : 0xffffu;  // (a chain off the path begins nowhere)
        chain_of = (uint32_t)lane >= from ? j : chain_of;
        base_of = (uint32_t)lane >= from ? from : base_of;
        if (mine) {
            off += some;
            end_chain = j, end_at = some;
            bool goes_on = some == have && (e & 0x10000u) != 0;  // all of it taken, and it landed on a chain
            open = goes_on;
            if (goes_on) {
                uint32_t nxt = (e >> 8) & 0xffu;
                on_path |= 1u << nxt;
                end_chain = nxt, end_at = 0;
            }
        }
    }
    uint32_t idx_of = (uint32_t)lane - base_of;
    n_out = uni(off);
    p_out = uni(pos + list[uni(end_chain) * (kAbreastCap + 1) + uni(end_at)]);
    my_start = pos + list[chain_of * (kAbreastCap + 1) + idx_of];
#ifdef NISSHI_PHASES
    uint64_t t3 = __builtin_readcyclecounter();
    if (lane == 0 && pos < 20000)
        printf("abreast pos=%u scan=%lu hops=%lu steps=%u join=%lu exist=%x n=%u\n", pos, t1 - t0, t2 - t1, steps_done, t3 - t2, exist, n_out);
#endif
    return true;
}

template <int kRows, bool kFold>
__device__ __forceinline__ int walk_chain(const uint8_t *blob, uint32_t b, const nisshi_columns &c, uint32_t rec_base,
                                          HeaderSlots &slots, BatchStream<kRows, kFold> &stream, int lane,
                                          uint32_t *k_io, uint32_t *pos_io, bool seg = false,
                                          uint32_t pos_end = 0xffffffffu, bool *minimal_io = nullptr,
                                          lds_u32 *scratch = nullptr) {
    typedef RingView<kRows> Ring;
    // (as in walk_streamed: whether every varint is written the shortest way, in the kernel of large batches)
    constexpr bool kSized = kRows != 8;
    bool all_minimal = true;
    uint32_t count = uni(c.record_count[b]);
    uint64_t rbase = rec_base;
    uint32_t data_len = uni(c.data_len[b]);
    uint32_t bias = 40 + stream.lo;  // the record data's offset in the stream
    constexpr uint32_t kRingBytes = RingCfg<kRows>::kRingBytes;
    constexpr int kAhead = RingCfg<kRows>::kAhead;
    constexpr uint32_t kReach = kRingBytes - 2 * kRowBytes;  // a round's records end within this of its start
    Ring ring{stream.ring};
    uint32_t k = uni(*k_io), pos = uni(*pos_io);
    int st = kWalkDone;
    // Rounds of records of 64 bytes and more on average find their starts by chains of hops side
    // by side (chains_abreast), in the kernel of large batches, which has the registers for it.
    uint32_t mean = count ? data_len / count : 0;
    bool abreast = kRows == 12 && scratch != nullptr && mean >= 64;
#ifdef NISSHI_NO_ABREAST
    abreast = false;
#endif
#ifdef NISSHI_PHASES
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tmark = __builtin_readcyclecounter();
    uint32_t n_rounds = 0, k_first = k;
#endif
    while (k < count && pos < pos_end) {
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
        bool found_abreast = false;
        if (abreast && want >= 32) {
            const lds_u32 *w = ring.ring + (ro >> 2);
            uint32_t d0 = uni(w[0]), d1 = uni(w[1]);
            uint32_t lo = (uint32_t)((((uint64_t)d1 << 32) | d0) >> (8 * (ro & 3)));
            uint32_t n2 = 0, p2 = pos, s2 = 0;
            // (the candidates are looked for by the form of a length of two bytes)
            if ((lo & 0x8080u) == 0x80u && chains_abreast<kRows>(ring, scratch, bias, pos, limit, want, mean, lane, n2, p2, s2)) {
                found_abreast = true;
                n = n2;
                p = p2;
                my_start = s2;
                ro = RingCfg<kRows>::wrap(p + bias);
                if (n < 24) abreast = false;  // not this batch's kind of records: one chain from here on
            }
        }
        if (!found_abreast) {
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
        if (stop == 1) {
            st = kWalkExact;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // the rows asked for are dropped
            break;
        }
        if (seg && stop == 2) {  // (a piece leaves such records to the walk of the whole batch)
            st = kWalkExact;
            __builtin_amdgcn_s_waitcnt(0x0F70);
            break;
        }
        if (seg) {  // only the records that start inside the piece
            uint64_t inside = __ballot((uint32_t)lane < n && my_start < pos_end);
            uint32_t n_in = (uint32_t)__builtin_popcountll(inside);
            if (n_in < n) {
                p = (uint32_t)__builtin_amdgcn_readlane((int)my_start, (int)n_in);
                n = n_in;
            }
        }
        if (stop == 2) {
            // ---- a record larger than the ring's reach, on its own (large_record); the rows
            // between here and the next record still pass through the stream for the CRC
            __builtin_amdgcn_s_waitcnt(0x0F70);
            uint32_t end = 0;
            if (kSized) all_minimal = false;  // (its varints are not looked at)
            int outcome = large_record(blob, b, c, rbase + k, pos, data_len, slots, lane, &end);
            if (outcome == 1) {
                st = kWalkExact;
                break;
            }
            if (outcome == 2) {
                if (lane == 0) c.status[b] = NISSHI_CAPACITY;
                k = count;
                break;
            }
            k += 1;
            pos = end;
            continue;
        }
        PHASE(3)
        // ---- a lane a record: the fields
        bool live = (uint32_t)lane < n;
        uint32_t start = live ? my_start : pos;
        uint32_t ro0 = RingCfg<kRows>::wrap(start + bias);
        RecordFields f;
        uint32_t used, q_used;
        uint32_t slack = 1;  // zero: one of the record's varints is longer than its value needs
        auto varint = [&](uint64_t w, uint32_t &n_used, int32_t &v) -> bool {
            return kSized ? fast_varint_noting(w, n_used, v, slack) : fast_varint(w, n_used, v);
        };
        bool ok = varint(ring.read40(ro0), used, f.length) && live;  // known to be good in live lanes
        uint32_t end = start + used + (uint32_t)f.length;
        uint32_t o = used;  // offset in the record
        auto at = [&](uint32_t off) { return Ring::wrap1(ro0 + off); };
        {
            uint64_t w0 = ring.read40(at(o));  // attributes, then the timestamp delta
            f.attributes = (uint32_t)w0 & 0xffu;
            o += 1;
            bool quick = fast_varlong((uint32_t)(w0 >> 8), q_used, f.timestamp_delta);
            if (!quick) ok = ring_varlong(ring, at(o), q_used, f.timestamp_delta) && ok;
            // (a delta of five bytes and more is not looked at: taken as not the shortest)
            if (kSized) {
                uint32_t last = quick ? (((uint32_t)(w0 >> 8) >> (8 * (q_used - 1))) & 0xffu) | ((q_used - 2u) >> 31) : 0u;
                slack = last < slack ? last : slack;
            }
            o += q_used;
        }
        uint32_t body = end - start;  // the record's bytes, its length varint included
        ok = varint(ring.read40(at(o)), q_used, f.offset_delta) && ok;
        o += q_used;
        ok = varint(ring.read40(at(o)), q_used, f.key_len) && ok;
        o += q_used;
        f.key_off = start + o;
        ok = ok && f.key_len >= -1 && o <= body && (uint32_t)(f.key_len < 0 ? 0 : f.key_len) <= body - o;
        o += ok && f.key_len > 0 ? (uint32_t)f.key_len : 0u;
        ok = varint(ring.read40(at(o)), q_used, f.val_len) && ok;
        o += ok ? q_used : 0u;
        f.val_off = start + o;
        ok = ok && f.val_len >= -1 && o <= body && (uint32_t)(f.val_len < 0 ? 0 : f.val_len) <= body - o;
        o += ok && f.val_len > 0 ? (uint32_t)f.val_len : 0u;
        ok = varint(ring.read40(at(o)), q_used, f.hdr_count) && ok;
        o += ok ? q_used : 0u;
        // a header is at least two bytes: a count the record cannot hold is an error
        ok = ok && f.hdr_count >= 0 && o <= body && (uint32_t)f.hdr_count <= (body - o) / 2;
        uint32_t hc = live && ok ? (uint32_t)f.hdr_count : 0;
        uint32_t slot = slots.next;  // (a round without headers: where they would have started, as every walk notes it)
        bool no_room = false;
        PHASE(4)
        if (__any(hc != 0)) {
            bool keep = slots.take(hc, n, lane, slot);
            if (slots.full) {
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
                    uint32_t slack_h = 1;
                    bool good = kSized ? fast_varint_noting(ring.read40(at(mine ? o : 0u)), n1, kl, slack_h)
                                       : fast_varint(ring.read40(at(mine ? o : 0u)), n1, kl);
                    uint32_t o1 = o + n1;
                    ko = start + o1;
                    good = good && kl >= -1 && o1 <= body && (uint32_t)(kl < 0 ? 0 : kl) <= body - o1;
                    o1 += good && kl > 0 ? (uint32_t)kl : 0u;
                    good = (kSized ? fast_varint_noting(ring.read40(at(mine && good ? o1 : 0u)), n2, vl, slack_h)
                                   : fast_varint(ring.read40(at(mine && good ? o1 : 0u)), n2, vl)) && good;
                    o1 += n2;
                    vo = start + o1;
                    good = good && vl >= -1 && o1 <= body && (uint32_t)(vl < 0 ? 0 : vl) <= body - o1;
                    o1 += good && vl > 0 ? (uint32_t)vl : 0u;
                    if (mine) {
                        ok = good;
                        slack = slack_h < slack ? slack_h : slack;
                        if (good) {
                            o = o1;
                            if (keep) store_header(c, slot + j, ko, kl, vo, vl);
                        }
                    }
                }
            }
        }
        if (no_room) {
            if (seg)
                st = kWalkFull;
            else if (lane == 0)
                c.status[b] = NISSHI_CAPACITY;
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
        if (kSized) all_minimal = all_minimal && !__any(live && slack == 0);
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
    if (kSized && minimal_io) *minimal_io = *minimal_io && all_minimal && st == kWalkDone;
    *k_io = k;
    *pos_io = pos;
    return st;
}

// One lane: Record::decode in sequence, exactly as
// Vec<Record>::try_from(deflated::Batch) does (deflated.rs:431-442).  The first
// error ends the batch with that status.  kStore false: only the outcome and the
// header count (what the batch needs of header slots); true: records are stored
// while the batch has record slots (a batch whose count cannot fit has none and
// only gets its status), headers from `hdr_first` on, of which it has `hdr_slots`.
template <bool kStore>
__device__ NISSHI_EXACT_INLINE int exact_walk(const uint8_t *blob, uint32_t b, const nisshi_columns &c, const Work &w,
                                          uint32_t rbase, uint32_t hdr_first, uint32_t hdr_slots, uint64_t hdr_cap,
                                          uint32_t *hdr_total_out) {
    uint32_t count = c.record_count[b];
    uint32_t slots = w.alloc_count[b];
    Reader rd(blob + c.data_off[b], c.data_len[b]);
    uint32_t pos = 0, hdr_total = 0;
    int st = NISSHI_OK;
    bool short_of_slots = false;
    for (uint32_t k = 0; k < count && st == NISSHI_OK; k++) {
        RecordFields f;
        st = read_record_head(rd, pos, f);
        if (st) break;
        // a header is at least two bytes: a count the data cannot hold fails below
        bool keep = kStore && k < slots && (uint64_t)(uint32_t)f.hdr_count * 2 <= rd.len - pos;
        if (keep && f.hdr_count > 0) {
            // headers beyond the batch's slots or the columns' end are not stored; the walk
            // goes on, since an error further on is the batch's outcome before that is
            if ((uint64_t)hdr_total + (uint32_t)f.hdr_count > hdr_slots ||
                (uint64_t)hdr_first + hdr_total + (uint32_t)f.hdr_count > hdr_cap) {
                short_of_slots = true;
                keep = false;
            }
        }
        for (int32_t j = 0; j < f.hdr_count; j++) {
            uint32_t ko, vo;
            int32_t kl, vl;
            if ((st = rd.octets(pos, ko, kl))) break;
            if ((st = rd.octets(pos, vo, vl))) break;
            if (keep) store_header(c, (uint64_t)hdr_first + hdr_total + (uint32_t)j, ko, kl, vo, vl);
        }
        if (st) break;
        if (kStore && k < slots) store_record(c, (uint64_t)rbase + k, 0, f, hdr_first + hdr_total);
        hdr_total += (uint32_t)f.hdr_count;
    }
    if (st == NISSHI_OK && short_of_slots) st = NISSHI_CAPACITY;
    *hdr_total_out = st == NISSHI_OK ? hdr_total : 0;
    return st;
}

// What the streamed walk leaves of a batch goes on a list for second_pass_kernel.
__device__ __forceinline__ void list_append(const Work &w, uint32_t b) {
    unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 7), 1ull);
    w.list[at] = b;
}

// A batch whose headers did not fit the slots it was expected to need: it gets
// `need` slots behind the expected ones (extra_base) and is walked again.
__device__ __forceinline__ void note_need(const Work &w, uint32_t b, uint32_t need) {
    w.hdr_need[b] = need;
    atomicAdd(w.blk_need + b / kSumBlock, need);
    atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 1), (unsigned long long)need);
}

// The walk of one batch by one wave: streamed (records of one shape, 64 at a
// time), chained where the records change from one to the next.
template <int kRows, bool kFold>
__device__ __forceinline__ int walk_batch(const uint8_t *blob, uint32_t b, const nisshi_columns &c, uint32_t rec_base,
                                          HeaderSlots &slots, BatchStream<kRows, kFold> &stream, lds_u128 *shape_tab,
                                          int lane) {
    uint32_t k = 0, pos = 0;
    int st = walk_streamed(b, c, rec_base, slots, stream, shape_tab, lane, &k, &pos);
    if (st == kWalkChain) st = walk_chain(blob, b, c, rec_base, slots, stream, lane, &k, &pos);
    return st;
}


// ---- large batches of small records, a wave a 64 KiB chunk
//
// The wave of chunk j of such a batch takes the chunk's share of the CRC and the
// records that start inside the chunk.  It cannot know where the chunk's first
// record starts, nor which record of the batch that is, so it makes a guess and
// the guesses are checked when all the chunks are done (segments_join): the
// start is the first position at or after the chunk's first byte from which three
// records in a row look like records (find_record_start), the record's index is its
// offset delta (a producer numbers the records of a batch 0, 1, 2 ...; a batch that
// does not is decoded by one wave, as a large batch of large records is), and
// every record has as many headers as the batch's first.  Per chunk the wave
// leaves {start, index, end, index at the end} in w.seg_info; a batch is decoded
// when chunk 0 starts at byte 0 with record 0, every chunk ends where the next
// starts, with its index, and the last ends at record_count (chunks in which no
// record was found may follow, the tail of the last record).  By induction every
// chunk then started on a record, with the index the sequential walk would have
// given it, so the columns are the sequential walk's.  When the chunks do not join
// up, whatever they stored is overwritten by the walk of the whole batch.
constexpr uint32_t kSegFailed = 0xffffffffu;
constexpr uint32_t kSegNone = 0xfffffffeu;  // no record starts in the chunk: right when the batch's records end before it
constexpr uint32_t kSegScanBytes = kSegRecordMax + 64;

// One record's claim to be a record, at byte q of the record data: its size and
// its offset delta.  Only a filter; nothing is concluded from it.  View: where the
// bytes are read, read64(position + bias) -- the wave's ring, or the batch in memory.
template <class View>
__device__ __forceinline__ bool plausible_record(const View &ring, uint32_t bias, uint32_t q, uint32_t data_len,
                                                 uint32_t &size, int32_t &od, uint32_t *key_at = nullptr) {
    uint32_t used, t_used, o_used;
    int32_t len;
    int64_t ts;
    uint64_t w0 = ring.read64(q + bias);
    bool ok = q < data_len && fast_varint(w0, used, len) && used <= 2 && len >= 6;
    size = ok ? used + (uint32_t)len : 0;
    ok = ok && size <= data_len - q && ((w0 >> (8 * used)) & 0xffu) == 0;  // attributes: no bit is in use
    uint32_t at = ok ? q + used + 1 : 0;
    ok = fast_varlong((uint32_t)ring.read64(at + bias), t_used, ts) && ok;
    at += ok ? t_used : 0;
    ok = fast_varint(ring.read64(at + bias), o_used, od) && ok;
    if (key_at) *key_at = at + o_used;
    return ok && od >= 0;
}

// Whether the fields of a plausible record of `size` bytes at q, from the key's
// length at `at` on, end where the record's length says (the headers of a record
// of more than kHoldsHeaders are not followed: it only must not have ended yet).
// A filter as well, for find_record_start where the deltas say less.
constexpr int kHoldsHeaders = 8;
template <class View>
__device__ __forceinline__ bool record_holds(const View &ring, uint32_t bias, uint32_t q, uint32_t size, uint32_t at, bool ok) {
    uint32_t end = q + size, used;
    int32_t v, hc;
    for (int span = 0; span < 2; span++) {  // the key, the value: a length (-1: null) and its bytes
        ok = fast_varint(ring.read64((ok ? at : 0u) + bias), used, v) && ok && v >= -1 && (v < 0 || (uint32_t)v <= size);
        at += ok ? used + (v > 0 ? (uint32_t)v : 0u) : 0u;
        ok = ok && at < end;
    }
    ok = fast_varint(ring.read64((ok ? at : 0u) + bias), used, hc) && ok && hc >= 0;
    at += ok ? used : 0u;
#pragma unroll 1
    for (int h = 0; h < 2 * kHoldsHeaders; h++) {  // a header's key, then its value
        bool more = ok && h < 2 * hc;
        if (!__any(more)) break;
        bool is = fast_varint(ring.read64((more ? at : 0u) + bias), used, v) && v >= -1 && (v < 0 || (uint32_t)v <= size);
        ok = ok && (!more || is);
        at += more && is ? used + (v > 0 ? (uint32_t)v : 0u) : 0u;
        ok = ok && at <= end;
    }
    return ok && (hc <= kHoldsHeaders ? at == end : at <= end);
}

// The first position in [x, x + kSegScanBytes) of the record data where three
// records in a row are plausible, with offset deltas that count up by one (fewer
// when the batch's last record comes first); `gapped`, of a batch that does not
// number its records 0, 1, 2 ...: with offset deltas that grow, none above the
// batch's last, `lod` (fewer when the record data ends), the first of the three
// with fields that end where its length says (record_holds).  The view holds the rows
// that cover the scan and three records of kSegRecordMax bytes behind it.
template <class View>
__device__ __forceinline__ bool find_record_start(const View &ring, uint32_t bias, uint32_t x, uint32_t data_len,
                                                  uint32_t count, int lane, uint32_t &pos0, uint32_t &k0,
                                                  bool gapped = false, uint32_t lod = 0) {
    for (uint32_t base = x; base < x + kSegScanBytes; base += kWave) {
        uint32_t q = base + (uint32_t)lane, size;
        int32_t od0, od;
        uint32_t key_at;
        bool ok = plausible_record(ring, bias, q, data_len, size, od0, &key_at);
        ok = ok && size <= kSegRecordMax && (gapped ? (uint32_t)od0 <= lod : (uint32_t)od0 < count);
        // (deltas that only have to grow say little: bytes that end on the next record's first
        // byte would do for a first record, were their fields not looked at)
        if (gapped) ok = record_holds(ring, bias, q, size, key_at, ok);
        uint32_t next = q + size;
        int32_t want = od0 + 1;
#pragma unroll 1
        for (int hop = 0; hop < 2; hop++) {
            // the batch's last record has none behind it
            bool more = ok && (gapped ? next < data_len : (uint32_t)want < count);
            bool good = plausible_record(ring, bias, more ? next : 0u, data_len, size, od);
            good = good && size <= kSegRecordMax && (gapped ? od >= want && (uint32_t)od <= lod : od == want);
            ok = ok && (!more || good);
            next += more ? size : 0;
            want = gapped && more ? od + 1 : want + 1;
        }
        uint64_t found = __ballot(ok);
        if (found) {
            uint32_t first = (uint32_t)__ffsll((long long)found) - 1;
            pos0 = base + first;
            k0 = (uint32_t)__builtin_amdgcn_readlane(od0, (int)first);
            return true;
        }
    }
    return false;
}

// ---- the first record index of every chunk of a batch that does not number its
// records 0, 1, 2 ... (kFlagGap)
//
// The chunk's wave counts the records that start in its chunk without walking them
// (records_from): records of one size S lie S bytes apart, so the lanes look at 64
// places at once -- is there a record of S bytes at start + i S? -- spread over the
// chunk; where one is not, the places between it and the last that was are looked at
// the same way until the first record of another size is known, and the count goes
// on from it with its size (a producer's records grow a byte where a delta does).
// Places not looked at are taken to hold records: the count is a guess, as the
// chunk's start is, and segments_join checks both -- a chunk's walk ends at the
// record index that the next one started with only if every count before was right.
// A chunk whose sizes change more often than kIndexTries times, or twice in a row
// after a record or two, has its records counted from length to length: by chains of
// hops side by side (chains_from), or one by one (hops_from).  The wave
// leaves its count for the waves of the chunks behind it and adds up those of the
// chunks before it (decode_unit).
constexpr int kIndexTries = 16;
constexpr uint32_t kCountFailed = 0xffffu;  // a chunk's count as it is left for the chunks behind it: sixteen bits under the call's tag
constexpr uint64_t kIndexWait = 50000;      // how long a chunk's wave waits for the counts before it: 500 us of the 100 MHz clock

// The batch in memory.  A read keeps the 32 bytes around it in registers: the head
// of a record -- its length, attributes and deltas -- is then one trip to memory.
struct MemoryView {
    const uint8_t *grid;  // the 16-byte boundary at or below the batch's first CRC byte
    uint32_t limit;       // bytes from there that may be read: to the end of the 16 bytes that hold the batch's last
    mutable uint32_t held_at = 0xffffffffu;  // where the words held start (a multiple of 8)
    mutable uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    __device__ __forceinline__ MemoryView(const uint8_t *grid_, uint32_t limit_) : grid(grid_), limit(limit_) {}
    __device__ __forceinline__ uint64_t word(uint32_t at) const {
        return at + 8 <= limit ? *reinterpret_cast<const uint64_t *>(grid + at) : 0;
    }
    __device__ __forceinline__ uint64_t read64(uint32_t at) const {
        uint32_t a = at & ~7u, sh = (at & 7u) * 8;
        if (a - held_at > 16u) {  // (below what is held wraps to a large number)
            held_at = a;
            h0 = word(a), h1 = word(a + 8), h2 = word(a + 16), h3 = word(a + 24);
        }
        uint32_t d = a - held_at;  // 0, 8 or 16
        uint64_t a0 = d & 16 ? h2 : h0, a1 = d & 16 ? h3 : h1;
        uint64_t lo = d & 8 ? a1 : a0, hi = d & 8 ? h2 : a1;
        return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    }
};

// The records that start in [pos, pos_end), the first at pos, counted one by one:
// from a record's length to the next record (its length in two bytes at most, no
// bit of its attributes in use: what plausible_record asks of a record's first
// bytes), for the records whose sizes change too often for records_from's strides.
// The chunk passes through the lanes' registers kHopRows KiB at a time, sixteen bytes
// of a KiB a lane: a hop reads its bytes from the lanes that hold them, and memory is
// waited for once a set of rows (a set's last row is the next set's first: a hop's
// bytes may lie across two rows).  (The next set on its way while the count goes
// through this one was tried: with loads under way around the loop the compiler has
// every hop wait for all of them.)  Up to kHopsMax records a
// chunk (kSegFailed beyond: thirty-two bytes a record on average fill a chunk with
// that many).
constexpr uint32_t kHopsMax = 2048;
constexpr int kHopRows = 8;
struct HopRows {
    u32x4 q[kHopRows + 1];
    __device__ __forceinline__ void load(const uint8_t *grid, uint32_t limit, uint32_t r, int lane) {
#pragma unroll
        for (int j = 0; j <= kHopRows; j++) {
            uint32_t at = (r + (uint32_t)j) * kRowBytes + (uint32_t)lane * 16;
            q[j] = at + 16 <= limit ? *reinterpret_cast<const u32x4 *>(grid + at) : u32x4{0, 0, 0, 0};
        }
    }
};

// The hops that start in rows r .. r + kHopRows - 1, which `rows` holds.  False: the count is over (*result).
// A hop is a few vector instructions and two dozen scalar ones: with the first word of the lane behind it a
// lane holds the four bytes at any of its sixteen positions, and the one lane that holds the hop's position
// hands them over.  Whether the bytes were a record's is looked at a row at a time (a hop goes forward
// whatever it read, by a length of two bytes at most: the row's loop ends).
__device__ __forceinline__ bool hops_in(const HopRows &rows, uint32_t &r, uint32_t bias, uint32_t &pos, uint32_t pos_end,
                                        uint32_t data_len, uint32_t &counted, uint32_t counted_max, uint32_t *result, int lane) {
#pragma unroll
    for (int s = 0; s < kHopRows; s++) {
        uint32_t behind = __shfl_down(rows.q[s].x, 1, 64);
        uint32_t next_row = (uint32_t)__builtin_amdgcn_readlane((int)rows.q[s + 1].x, 0);
        u32x4 low = rows.q[s], high = {low.y, low.z, low.w, lane == kWave - 1 ? next_row : behind};
        uint32_t row_end = (r + 1) * kRowBytes - bias;  // (the record data's position of the next row's first byte)
        uint32_t stop = row_end < pos_end ? row_end : pos_end;
        uint32_t bad = 0;
        while (pos < stop) {
            uint32_t at = pos + bias;
            // (the lane's eight bytes from the position's word on: by eight bytes, then by four)
            bool by8 = (at & 8u) != 0, by4 = (at & 4u) != 0;
            uint32_t a0 = by8 ? low.z : low.x, a1 = by8 ? low.w : low.y, a2 = by8 ? high.w : low.z;
            uint32_t mine = __builtin_amdgcn_alignbyte(by4 ? a2 : a1, by4 ? a1 : a0, at & 3u);
            uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)((at >> 4) & 63u));
            uint32_t two = (head >> 7) & 1u;                        // the length goes on in a second byte
            uint32_t v = (head & 0x7fu) | (two ? (head >> 1) & 0x3f80u : 0u);
            uint32_t size = 1 + two + (v >> 1);
            uint32_t attributes = (head >> (two ? 16 : 8)) & 0xffu;
            // (v < 12, size > data_len - pos: as sign bits, every value here is far below 2^31)
            bad |= (two & (head >> 15)) | (v & 1u) | ((v - 12u) >> 31) | attributes | ((data_len - pos - size) >> 31);
            pos += size;
            counted += 1;
        }
        *result = kSegFailed;
        if (bad || counted > counted_max) return false;
        if (pos >= pos_end) {
            *result = counted;
            return false;
        }
        r += 1;
    }
    return true;
}

__device__ __forceinline__ uint32_t hops_from(const uint8_t *grid, uint32_t limit, uint32_t bias, uint32_t pos, uint32_t pos_end,
                                              uint32_t data_len, uint32_t counted, int lane) {
    // (every lane has the same of each of these: in scalar registers, and the loop's branches scalar ones)
    bias = uni(bias), pos = uni(pos), pos_end = uni(pos_end), data_len = uni(data_len), counted = uni(counted);
    uint32_t r = (pos + bias) / kRowBytes, result = kSegFailed, counted_max = counted + kHopsMax;
    for (;;) {
        HopRows held;
        held.load(grid, limit, r, lane);
        if (!hops_in(held, r, bias, pos, pos_end, data_len, counted, counted_max, &result, lane)) return result;
    }
}

// The same count by 64 chains of hops side by side, as chains_abreast finds the starts of a round:
// the stretch is cut into 64 shares, lane 0's chain starts at pos and every other lane's at a
// candidate, the first place in the first 192 bytes of its share that looks like the head of a
// record (a length of one or two bytes that is not negative, an attributes byte of zero, a byte that
// is not zero two behind it: a delta).  All chains hop in
// step, each through a view of the batch in memory of its own, until a chain lands exactly on the
// start of a chain ahead of it (one it passes without landing on was no record's start: it aims at
// the next).  The count is the sum over the chains on the path from lane 0's, which ends with the
// chain that passes pos_end: nothing of a chain is used unless hops from pos led to where it
// started.  kSegFailed (and the count is hops_from's): shares of less than 200 bytes, a chain on the
// path that is still under way after kChainSteps hops, a hop that does not read as a record.
constexpr uint32_t kChainSteps = 64, kChainScan = 16, kChainWindows = 3;  // hops a chain; dwords a window, windows a lane looks through for its candidate
__device__ __forceinline__ uint32_t chains_from(const uint8_t *grid, uint32_t limit, uint32_t bias, uint32_t pos,
                                                uint32_t pos_end, uint32_t data_len, int lane) {
    typedef uint32_t __attribute__((aligned(4))) u32_al4;
    uint32_t stride = (pos_end - pos) / kWave;
    if (stride < 4 * kChainScan * kChainWindows + 8) return kSegFailed;
    // ---- the candidates
    uint32_t start = lane == 0 ? pos : 0xffffffffu;
#pragma unroll 1
    for (uint32_t window = 0; window < kChainWindows; window++) {  // (64 bytes at a time: a record of 200 bytes starts in one of them)
        if (!__any(start == 0xffffffffu)) break;
        uint32_t q0 = ((pos + (uint32_t)lane * stride + bias) & ~3u) + window * 4 * kChainScan;  // from the batch's grid, a whole word
        bool inside = q0 + 4 * kChainScan + 4 <= limit;
        const u32_al4 *rp = reinterpret_cast<const u32_al4 *>(grid + (inside ? q0 : 0u));
        uint32_t d[kChainScan + 1];
#pragma unroll
        for (uint32_t j = 0; j <= kChainScan; j++) d[j] = rp[j];
        uint32_t first = 0xffffu;
#pragma unroll
        for (int j = (int)kChainScan - 1; j >= 0; j--) {
            uint32_t w = d[j], nx = d[j + 1];
            uint32_t b1 = __builtin_amdgcn_alignbit(nx, w, 8), b2 = __builtin_amdgcn_alignbit(nx, w, 16);
            uint32_t b3 = __builtin_amdgcn_alignbit(nx, w, 24);
            auto not_zero = [](uint32_t v) { return ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v; };  // bit 7 of every byte that is not zero
            // the byte continues, is even (no negative length), the next ends the varint, the one after (the
            // attributes) is zero -- and the byte two behind that is not: the end of a record without headers,
            // "xx xx 00" and the next record's length and attributes, reads the same up to there
            uint32_t cand = w & ~(w << 7) & ~b1 & ~not_zero(b2) & not_zero(nx) & 0x80808080u;
            // or a length of one byte: the byte does not continue, is even and twelve at least (six bytes and
            // more), the next is zero, the byte two behind that is not
            cand |= ~w & ~(w << 7) & ((w & 0x7f7f7f7fu) + 0x74747474u) & ~not_zero(b1) & not_zero(b3) & 0x80808080u;
            uint32_t at = (uint32_t)j * 4 + ((uint32_t)__builtin_ctz(cand | 0x80000000u) >> 3);
            first = cand ? at : first;
        }
        uint32_t found = q0 + first - bias;
        if (start == 0xffffffffu && inside && first != 0xffffu && q0 + first >= bias && found > pos && found + 7 <= pos_end)
            start = found;
    }
    uint64_t exist = __ballot(start != 0xffffffffu);
    auto chain_after = [&](uint32_t t) -> uint32_t {  // the next chain there is after chain t; kWave: none
        uint64_t ahead = t >= kWave - 1 ? 0ull : exist & ~((2ull << t) - 1ull);
        return ahead ? (uint32_t)__builtin_ctzll(ahead) : kWave;
    };
    // ---- the hops
    MemoryView view(grid, limit);
    bool alive = start != 0xffffffffu, landed = false, through = false;  // through: past pos_end, with no chain ahead
    uint32_t pc = alive ? start : pos, cnt = 0;
    uint32_t tgt = chain_after((uint32_t)lane);
    uint32_t tpos = (uint32_t)__shfl((int)start, (int)(tgt < kWave ? tgt : 0u), 64);
    tpos = tgt < kWave ? tpos : 0xffffffffu;
#pragma unroll 1
    for (uint32_t step = 0; step < kChainSteps; step++) {
        bool act = alive && !landed && !through;
        if (!__any(act)) break;
        uint64_t w0 = view.read64((act ? pc : pos) + bias);
        uint32_t used;
        int32_t len;
        bool ok = fast_varint(w0, used, len) && used <= 2 && len >= 6 && ((w0 >> (8 * used)) & 0xffu) == 0;
        uint32_t size = used + (uint32_t)len;
        bool take = act && ok && size <= data_len - pc;
        if (act && !take) alive = false;  // no record here: the chain was on no record's start, or the data is damaged
        cnt += take ? 1u : 0u;
        pc += take ? size : 0u;
#pragma unroll 1
        for (uint32_t t = 0; t < kWave && __any(take && pc > tpos); t++) {  // passed a start without landing on it
            bool passed = take && pc > tpos;
            uint32_t nt = chain_after(passed ? tgt : 0u);
            uint32_t np = (uint32_t)__shfl((int)start, (int)(nt < kWave ? nt : 0u), 64);
            tgt = passed ? nt : tgt;
            tpos = passed ? (nt < kWave ? np : 0xffffffffu) : tpos;
        }
        landed = landed || (take && pc == tpos);
        through = through || (take && !landed && pc >= pos_end);
    }
    // ---- the chains that were landed on, in order from lane 0's
    uint32_t ended = cnt | (tgt << 16) | (landed ? 0x800000u : 0u) | (through ? 0x1000000u : 0u);
    uint32_t total = 0, t = 0;
    for (uint32_t j = 0; j < kWave; j++) {
        uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)ended, (int)t);
        total += e & 0xffffu;
        if (e & 0x1000000u) return total;
        if (!(e & 0x800000u)) return kSegFailed;
        t = (e >> 16) & 0x7fu;
    }
    return kSegFailed;
}

// The records that start in [pos, pos_end) of the record data, the first of them at
// pos (kSegFailed: not counted).
__device__ __forceinline__ uint32_t records_from(const MemoryView &view, uint32_t bias, uint32_t pos, uint32_t pos_end,
                                                 uint32_t data_len, int lane) {
    uint32_t counted = 0, poor = 0;
    for (int tries = 0; tries < kIndexTries; tries++) {
        if (pos >= pos_end) return counted;
        uint32_t size;
        int32_t od;
        if (!plausible_record(view, bias, pos, data_len, size, od)) return kSegFailed;
        size = uni(size);
        // records 0 .. m - 1 of this size from pos on start before the end; record 0 is there
        uint32_t m = (pos_end - pos + size - 1) / size, good_to = 0, bad_at = m;
        while (bad_at - good_to > 1) {
            uint32_t between = bad_at - good_to - 1, step = (between + kWave - 1) / kWave;
            uint32_t i = lane == kWave - 1 ? bad_at - 1 : good_to + 1 + (uint32_t)lane * step;
            bool looked = i < bad_at;
            uint32_t sz;
            bool is = plausible_record(view, bias, looked ? pos + i * size : pos, data_len, sz, od) && sz == size;
            uint64_t bad = __ballot(looked && !is);
            if (!bad) {
                good_to = bad_at - 1;
            } else {
                uint32_t f = (uint32_t)__ffsll((long long)bad) - 1;
                uint64_t before = __ballot(looked) & ((1ull << f) - 1);
                bad_at = (uint32_t)__builtin_amdgcn_readlane((int)i, (int)f);
                if (before) good_to = (uint32_t)__builtin_amdgcn_readlane((int)i, 63 - __builtin_clzll(before));
            }
        }
        counted += good_to + 1;
        pos += (good_to + 1) * size;
        if (good_to + 1 == m) return counted;
        // (two sizes that held for a record or two: these records are not of a size for a stretch)
        poor += good_to < 2 ? 1 : 0;
        if (poor == 2) break;
    }
    // (records of every size: by chains of hops side by side, or failing that one by one)
    uint32_t rest = chains_from(view.grid, view.limit, uni(bias), uni(pos), uni(pos_end), uni(data_len), lane);
    if (rest != kSegFailed) return counted + rest;
    return hops_from(view.grid, view.limit, bias, pos, pos_end, data_len, counted, lane);
}

// Whether the chunks of batch b, whose work items start at `first`, join up.
// *sized: where the last record ends plus one if every chunk found its records written
// the shortest way (the top bit of a chunk's end), else zero.
__device__ __forceinline__ bool segments_join(const Work &w, uint32_t first, uint32_t chunks, uint32_t count, int lane,
                                              uint32_t *sized) {
    bool good = true, minimal = true;
    uint32_t end = 0;
    for (uint32_t j = (uint32_t)lane; j < chunks; j += kWave) {
        u32x4 mine = *reinterpret_cast<const u32x4 *>(w.seg_info + 4 * (uint64_t)(first + j));
        minimal = minimal && (mine.z >> 31);
        mine.z &= 0x7fffffffu;
        if (mine.w == count && mine.z > end) end = mine.z;
        good = good && mine.w != kSegFailed;
        if (j == 0) good = good && mine.x == 0 && mine.y == 0;
        if (j + 1 < chunks) {
            u32x4 next = *reinterpret_cast<const u32x4 *>(w.seg_info + 4 * (uint64_t)(first + j + 1));
            if (mine.w == kSegNone)
                good = good && next.w == kSegNone;
            else if (next.w == kSegNone)
                good = good && mine.w == count;
            else
                good = good && mine.z == next.x && mine.w == next.y;
        } else {
            good = good && (mine.w == kSegNone || mine.w == count);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t other = __shfl_xor(end, d, 64);
        end = other > end ? other : end;
    }
    *sized = __all(minimal) ? end + 1 : 0u;
    return __all(good);
}

// The CRC work items by the decoder's numbering, for a wave that goes through
// some of them: a block of batches from the block sums (every wave adds them up
// itself, or finds its way in the running sums), the batch within the block by
// binary search.  No items, which is the usual case, costs a wave two loads a lane.
struct ChunkItems {
    uint32_t n, n_blocks, total, ex0, ex1;
    bool prefixed;

    __device__ __forceinline__ void init(const Work &w, uint32_t n_, bool prefixed_, int lane) {
        n = n_;
        prefixed = prefixed_;
        n_blocks = (n + kSumBlock - 1) / kSumBlock;
        ex0 = ex1 = 0;
        if (prefixed) {
            total = w.blk_chunks[n_blocks];
        } else {  // n_blocks <= kSumDirect = 128: lane j holds the sums before blocks j and j + 64
            uint32_t s0 = (uint32_t)lane < n_blocks ? w.blk_chunks[lane] : 0;
            uint32_t s1 = (uint32_t)lane + 64 < n_blocks ? w.blk_chunks[lane + 64] : 0;
            uint32_t i0 = s0, i1 = s1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t u0 = __shfl_up(i0, d, 64), u1 = __shfl_up(i1, d, 64);
                if (lane >= d) i0 += u0, i1 += u1;
            }
            i1 += (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
            total = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
            ex0 = i0 - s0;
            ex1 = i1 - s1;
        }
        total = uni(total);
    }

    // item -> (batch, chunk)
    __device__ __forceinline__ void locate(const Work &w, uint32_t item, int lane, uint32_t &b, uint32_t &chunk) const {
        uint32_t blk, base;
        if (prefixed) {
            uint32_t lo_k = 0, hi_k = n_blocks;  // largest block whose running sum is <= item
            while (hi_k - lo_k > 1) {
                uint32_t mid = (lo_k + hi_k) >> 1;
                if (w.blk_chunks[mid] <= item) lo_k = mid; else hi_k = mid;
            }
            blk = lo_k;
            base = w.blk_chunks[blk];
        } else {
            uint64_t m0 = __ballot((uint32_t)lane < n_blocks && ex0 <= item);
            uint64_t m1 = __ballot((uint32_t)lane + 64 < n_blocks && ex1 <= item);
            uint32_t top = m1 ? 63u - (uint32_t)__builtin_clzll(m1) : 63u - (uint32_t)__builtin_clzll(m0);
            blk = m1 ? top + 64 : top;
            base = (uint32_t)__builtin_amdgcn_readlane((int)(m1 ? ex1 : ex0), (int)top);
        }
        uint32_t local = item - base;
        // the largest batch of the block whose first item is <= local: the first items
        // only grow, so that is as many batches in as there are such first items, all
        // looked at at once (four a lane)
        static_assert(kSumBlock == 4 * kWave, "a lane looks at four batches of the block");
        uint32_t lo_b = blk * kSumBlock, at_b = lo_b + 4 * (uint32_t)lane, found = 0;
        uint32_t first[4];
#pragma unroll
        for (int j = 0; j < 4; j++) first[j] = at_b + j < n ? w.chunk_base[at_b + j] : 0xffffffffu;
#pragma unroll
        for (int j = 0; j < 4; j++) found += (uint32_t)__popcll(__ballot(first[j] <= local));
        uint32_t in = uni(found - 1);  // (the block's first batch has first item 0)
        b = lo_b + in;
        uint32_t mine = first[in & 3];
        chunk = uni(local - (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)(in >> 2)));
    }
};

// One unit of decode_fused_kernel's work: batch b, or with `seg` its chunk `chunk`
// (CRC work item `mine`).  `seg` is a constant where this is called.
//
// `phase` (of a batch, not of a chunk) says what the launch is there for:
//   kPhaseAll     every batch; the chunks of large batches are through (chunk_kernel ran before).
//   kPhaseBeside  the chunks are being worked on by other waves of this launch: every
//                 batch but those decoded chunk by chunk; of a batch whose chunks only
//                 give its CRC everything but that CRC.
//   kPhaseAfter   what kPhaseBeside left: the batches decoded chunk by chunk (do the
//                 chunks join up? else the batch is walked here) and the CRCs from chunks.
constexpr uint32_t kPhaseAll = 0, kPhaseBeside = 1, kPhaseAfter = 2;

template <int kRows>
__device__ __forceinline__ void decode_unit(bool seg, uint32_t b, uint32_t chunk, uint32_t mine, const uint8_t *blob,
                                            const uint64_t *batch_off, uint32_t n, const nisshi_columns &c, const Work &w,
                                            uint64_t rec_cap, uint64_t hdr_cap, const CrcTables *t, uint32_t prefixed,
                                            const uint64_t *before, uint32_t *ring, uint32_t *shape_tab, int lane,
                                            uint32_t phase = kPhaseAll, uint32_t tag = 0) {
#ifdef NISSHI_PHASES
    uint64_t tl[6];
    uint64_t wall0 = wall_clock64();
    tl[0] = __builtin_readcyclecounter();
#endif
    // What the unit needs to know of its batch lies at addresses that b alone gives:
    // all of it is asked for in one go and waited for once, so that the rows can be
    // asked for one trip to the L2 later and nothing else is waited for before they come.
    // (The status and the record count are loaded further down, where they are used:
    // placed here with the others they cost the chained walk 4 %.)
    uint32_t chunks_v = w.n_chunks[b], flags_v = w.flags[b];
    asm volatile("" ::: "memory");
    uint64_t off_v = batch_off[b];
    uint32_t length_v = (uint32_t)c.batch_length[b], expect_v = w.hdr_pred[b];
    BasesLoad bases_v = batch_bases_load(w, b, prefixed != 0, before, lane);
    u32x4 guess = u32x4{0, 0, 0, 0};  // the guess at the shape of the block's records (guess_shape)
    if (w.shape_guess != nullptr && (uint32_t)lane < kShapeWords / 4) {
        typedef __attribute__((address_space(1))) const u32x4 gbl_u128;
        guess = ((gbl_u128 *)(uintptr_t)w.shape_guess)[(uint64_t)(b / kSumBlock) * (kShapeWords / 4) + (uint32_t)lane];
    }
    uint32_t chunks = uni(chunks_v);
    uint32_t flags = uni(flags_v);
    if (!seg && phase == kPhaseBeside && (flags & kFlagSeg)) return;
    if (!seg && phase == kPhaseAfter && !(flags & kFlagChunks)) return;
    int status = (int)uni((uint32_t)c.status[b]);
    uint32_t count = uni(c.record_count[b]);
    // The unit's first rows are asked for before anything else is worked out: where
    // the batch's records and headers go takes a few trips to the L2, and the rows
    // come from further away.
#ifdef NISSHI_DIAG_NOFOLD
    BatchStream<kRows, false> stream;
#else
    BatchStream<kRows, true> stream;
#endif
    constexpr int kAhead = RingCfg<kRows>::kAhead;
    u32x4 first_rows[kAhead];
    uint32_t n_first = 0;
    // A batch whose chunks were work items of chunk_kernel, which is through with
    // them: its CRC from their residues; and its records are there, if they were
    // decoded chunk by chunk and the chunks join up.  Otherwise the batch is walked
    // here like any other, but for the CRC.
    bool in_chunks = !seg && (flags & kFlagChunks);
    bool joined_up = false;
    uint32_t sized = 0;
    if (in_chunks && phase != kPhaseBeside) {
        uint32_t first_item = chunk_first(w, b, prefixed != 0, lane);
        uint32_t v = crc_combine_wave(batch_off, c.batch_length, b, chunks, w.partial + first_item, t, lane);
        if (lane == 0) c.crc_computed[b] = v;
        joined_up = (flags & kFlagSeg) && status == NISSHI_OK && segments_join(w, first_item, chunks, count, lane, &sized);
        // (counted: batches whose chunks did not join up, walked whole below; only by the launch behind
        // chunks that were worked on beside the batches -- the kernels that take every batch of a call
        // slow down by 3 % with one more thing to do, rare as it is)
        if (phase == kPhaseAfter && (flags & kFlagSeg) && status == NISSHI_OK && !joined_up && lane == 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 5), 1ull);
#ifdef NISSHI_DIAG_JOINS  // diagnostic builds only: batches whose chunks did not join up, counted as sequential walks
        if ((flags & kFlagSeg) && !joined_up && lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 3), 1ull);
            printf("no join: b=%u chunks=%u count=%u\n", b, chunks, count);
            for (uint32_t j = 0; j < chunks && j < 20; j++) {
                const uint32_t *q = w.seg_info + 4 * (uint64_t)(first_item + j);
                printf("   %u: start %u index %u end %u index %u\n", j, q[0], q[1], q[2], q[3]);
            }
        }
#endif
    }
    if (in_chunks && phase == kPhaseAfter && !(flags & kFlagSeg)) return;  // (walked beside its chunks: only the CRC was left)
    bool streamed = chunks != 0 && !joined_up && !(in_chunks && status != NISSHI_OK);  // (nothing to walk, and the CRC is there)
#ifdef NISSHI_DIAG_NOSTREAM  // diagnostic builds only: what the kernel costs without parts of its work
    streamed = false;
#endif
    uint32_t row0 = chunk * kChunkRows, fold_to = seg ? row0 + kChunkRows : 0xffffffffu;
    uint32_t bias = 0, x = 0;  // x: the chunk's first byte as a position in the record data
    if (streamed) {
        uint64_t lo = (((uint64_t)uni((uint32_t)(off_v >> 32)) << 32) | uni((uint32_t)off_v)) + 21;
        stream.init(blob, lo, lo + (uint64_t)uni(length_v) - 9, (lds_u32 *)ring, lane);
        bias = 40 + stream.lo;
        if (seg) {
            stream.segment(row0, fold_to);
            x = chunk ? row0 * kRowBytes - bias : 0;
            // what the search for the chunk's first record reads
            n_first = stream.row_after(bias + x + kSegScanBytes + 3 * kSegRecordMax + 80) - row0;
        } else {
            if (in_chunks) stream.pass_through();
            n_first = stream.rows;
        }
        n_first = n_first < (uint32_t)kAhead ? n_first : (uint32_t)kAhead;
        stream.load(first_rows, n_first);
    }
    // where the batch's records and headers go
    bool guessed = streamed && w.shape_guess != nullptr;
    uint32_t rec_base, hdr_base;
    batch_bases_sum(bases_v, prefixed != 0, rec_base, hdr_base);
    if (guessed && (uint32_t)lane < kShapeWords / 4) ((lds_u128 *)shape_tab)[lane] = guess;
#ifdef NISSHI_PHASES
    tl[1] = __builtin_readcyclecounter();
#endif
    uint32_t expect = uni(expect_v);
    if (!seg) {
        if (lane == 0) {
            c.rec_base[b] = rec_base;
            c.hdr_base[b] = hdr_base;
            if (b + 1 == n) {  // the last batch: the ends of the record slots and of the expected header slots
                w.totals[0] = (uint64_t)rec_base + w.alloc_count[b];
                w.totals[6] = (uint64_t)hdr_base + expect;
                atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 1), (unsigned long long)hdr_base + expect);
            }
        }
        if (chunks == 0) {
            // done already (a small batch of a call of small batches), or no CRC to take: the header failed early
            if (lane == 0 && status != NISSHI_OK) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), 1ull);
            return;
        }
        if (joined_up) {
            if (lane == 0) {
                c.hdr_count[b] = expect;
                if (c.enc_len != nullptr) c.enc_len[b] = sized;
            }
            return;
        }
    }
#ifdef NISSHI_DIAG_NOSTREAM
    return;
#endif
    bool do_walk = status == NISSHI_OK && !(flags & kFlagExact);
    if (do_walk && (uint64_t)rec_base + count > rec_cap) {
        // (of a batch in chunks the second pass says so)
        if (!seg && lane == 0) c.status[b] = NISSHI_CAPACITY;
        status = NISSHI_CAPACITY;
        do_walk = false;
    }
#ifdef NISSHI_DIAG_NOWALK
    do_walk = false;
#endif
    stream.take_loaded(first_rows, n_first);
    bool exact = !seg && status == NISSHI_OK && (flags & kFlagExact);
#ifdef NISSHI_PHASES
    tl[2] = __builtin_readcyclecounter();
#endif
    // ---- where the unit's walk starts and ends
    uint32_t k0 = 0, pos0 = 0, pos_end = 0xffffffffu, hdr_first = hdr_base, hdr_slots = expect;
    bool none = false;  // seg: no record starts in the chunk
    bool gapped = seg && (flags & kFlagGap) != 0;
    if (seg && (do_walk || gapped)) {
        lds_order();
        uint32_t data_len = uni(c.data_len[b]);
        if (chunk && do_walk) {
            uint32_t lod = gapped ? uni((uint32_t)c.last_offset_delta[b]) : 0u;
            do_walk = x < data_len &&
                      find_record_start(RingView<kRows>{stream.ring}, bias, x, data_len, count, lane, pos0, k0, gapped, lod);
            none = !do_walk;
        }
        pos0 = uni(pos0);
        k0 = uni(k0);
        if (chunk + 1 < chunks) pos_end = fold_to * kRowBytes - bias;
        if (gapped) {
            // A batch that does not number its records 0, 1, 2 ...: the chunk's first index
            // is the number of records that start in the chunks before it.  Every chunk's
            // wave counts its own (records_from) and leaves the count, under the call's tag,
            // for the waves of the chunks behind it, before it waits for those of the chunks
            // before it -- which are further on than itself: their workgroups come first in
            // the grid.  (A wave that waits in vain, kIndexWait, gives up: the chunks then
            // do not join up and the batch is walked whole.)
            uint32_t here = kCountFailed;
            if (none) here = 0;
            if (do_walk) {
                MemoryView view(stream.lane_ptr - (uint32_t)lane * 16, ((stream.hi - 1) & ~15u) + 16);
                uint32_t to = pos_end < data_len ? pos_end : data_len;
                uint32_t n_here = records_from(view, bias, pos0, to, data_len, lane);
                here = n_here < kCountFailed ? n_here : kCountFailed;
            }
            if (lane == 0)
                __hip_atomic_store(w.chunk_done + mine, (tag << 16) | here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t sum = 0;
            bool known = do_walk || none;
            for (uint32_t j0 = 0; known && j0 < chunk; j0 += kWave) {
                uint32_t j = j0 + (uint32_t)lane, v = 0;
                uint64_t since = wall_clock64();
                for (;;) {
                    if (j < chunk) v = __hip_atomic_load(w.chunk_done + (mine - chunk + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all(j >= chunk || (v >> 16) == tag)) break;
                    if (wall_clock64() - since > kIndexWait) {
                        v = kCountFailed;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
                known = __all(j >= chunk || (v & 0xffffu) != kCountFailed);
                sum += j < chunk ? v & 0xffffu : 0u;
            }
            for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
            k0 = uni(sum);
            // (a sum at or past the batch's record count can only be wrong, and would put records in another batch's slots)
            if (!known || (do_walk && k0 >= count)) do_walk = none = false;  // (kSegFailed below)
        }
        // every record is taken to have the header count of the batch's first: the chunk's headers start k0 such records in
        uint32_t hc = count ? expect / count : 0;
        hdr_first += k0 * hc;
        hdr_slots -= k0 * hc;
    }
    uint32_t k = k0, pos = pos0;
    bool joined = false;  // seg: the chunk's records are decoded, as far as the chunk can tell
    bool minimal = none;  // the unit's records are all written the shortest way (none are)
    if (do_walk) {
        HeaderSlots slots;
        slots.init(hdr_first, hdr_slots, hdr_cap);
        lds_order();
        int st = walk_streamed(b, c, rec_base, slots, stream, (lds_u128 *)shape_tab, lane, &k, &pos, seg, pos_end, guessed, &minimal);
        if (st == kWalkChain)
            st = walk_chain(blob, b, c, rec_base, slots, stream, lane, &k, &pos, seg, pos_end, &minimal, (lds_u32 *)shape_tab);
#ifdef NISSHI_PHASES
        tl[3] = __builtin_readcyclecounter();
#endif
        if (seg) {
            joined = st == kWalkDone && !slots.over && !slots.full && slots.count == (k - k0) * (count ? expect / count : 0);
        } else if (st == kWalkExact) {
            exact = true;
        } else if (lane == 0) {
            c.hdr_count[b] = slots.count;
            // every record written the shortest way: they encode to the bytes they came from
            if (kRows != 8 && minimal && k == count && !slots.over && !slots.full && c.enc_len != nullptr) c.enc_len[b] = pos + 1;
            if (uni((uint32_t)c.status[b]) != NISSHI_OK) {  // the columns ended inside the batch
                atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), 1ull);
            } else if (slots.over) {
                note_need(w, b, slots.count);
                w.flags[b] = flags | kFlagMore;
                list_append(w, b);
            }
        }
    } else if (!seg && lane == 0 && status != NISSHI_OK) {
        atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), 1ull);
    }
    if (exact && lane == 0) {
        // The sequential walk decides: first only the outcome and the header count, so
        // that the batch's header slots are known before anything is stored behind the
        // expected ones; the records and headers are stored by second_pass_kernel.
        uint32_t headers = 0;
        int st = exact_walk<false>(blob, b, c, w, rec_base, 0, 0, 0, &headers);
        atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 3), 1ull);
        if (st == NISSHI_OK && (uint64_t)rec_base + w.alloc_count[b] > rec_cap) st = NISSHI_CAPACITY;
        c.status[b] = st;
        c.hdr_count[b] = headers;
        if (st != NISSHI_OK) {
            atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), 1ull);
        } else {
            if (headers > expect) note_need(w, b, headers);
            w.flags[b] = flags | kFlagExact;
            list_append(w, b);
        }
    }
    if (seg) {
        uint32_t r = stream.finish_segment(fold_to, t);
        if (lane == 0) {
            w.partial[mine] = r;
            *reinterpret_cast<u32x4 *>(w.seg_info + 4 * (uint64_t)mine) =
                u32x4{pos0, k0, pos | (minimal ? 0x80000000u : 0u), joined ? k : none ? kSegNone : kSegFailed};
        }
    } else if (!in_chunks) {
        uint32_t v = stream.finish(t);
        if (lane == 0) c.crc_computed[b] = v;
    }
    lds_order();
#ifdef NISSHI_PHASES
    tl[4] = __builtin_readcyclecounter();
    if (lane == 0 && (b % 1024) == 7)
        printf("fused b=%u wall=%lu..%lu bases=%lu init=%lu walk=%lu finish=%lu\n", b, wall0, wall_clock64(), tl[1] - tl[0],
               tl[2] - tl[1], tl[3] - tl[2], tl[4] - tl[3]);
#endif
}

// What a call leaves in page-locked memory for the host's next call on the context:
// how many batches it listed, and how many CRC work items its large batches have.
// By one wave, when the list is complete.
__device__ __forceinline__ void leave_hints(const Work &w, uint32_t n, bool prefixed, int lane, uint32_t entries) {
    if (w.hint == nullptr) return;
    ChunkItems items;
    items.init(w, n, prefixed, lane);
    if (lane == 0) {
        w.hint[0] = entries;
        w.hint[1] = items.total;
    }
}

// The chunks of large batches, a wave each (as many passes as the grid needs),
// ahead of decode_fused_kernel: of a batch of small records a chunk's wave takes
// the chunk's share of the CRC and of the records (see segments_join), of any
// other batch of kBigChunks chunks and more only the CRC (one wave cannot stream
// such a batch fast enough; its records are walked by its wave of
// decode_fused_kernel).  Most calls have no such batch, and every wave is done
// after two loads a lane.
template <int kRows>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void chunk_kernel(
    const uint8_t *blob, const uint64_t *batch_off, uint32_t n, nisshi_columns c, Work w, uint64_t rec_cap,
    uint64_t hdr_cap, const CrcTables *t, uint32_t prefixed, const uint64_t *before, uint32_t tag) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[RingCfg<kRows>::kRingWords];
    __shared__ __attribute__((aligned(16))) uint32_t shape_tab[kShapeWords];
    int lane = threadIdx.x & 63;
    ChunkItems items;
    items.init(w, n, prefixed != 0, lane);
    for (uint32_t item = blockIdx.x; item < items.total; item += gridDim.x) {
        uint32_t b, chunk;
        items.locate(w, item, lane, b, chunk);
        if (uni(w.flags[b]) & kFlagSeg)
            decode_unit<kRows>(true, b, chunk, item, blob, batch_off, n, c, w, rec_cap, hdr_cap, t, prefixed, before, ring,
                               shape_tab, lane, kPhaseAll, tag);
        else
            crc_chunk_one(item, b, chunk, blob, batch_off, c.batch_length, c.crc_computed, w, t, lane);
    }
}

// One wave per batch: CRC-32C of the batch streamed just ahead of the record
// walk through an LDS ring, so the batch is read from HBM once.
// kPhase (see decode_unit): kPhaseAll is the plain kernel; kPhaseBeside and
// kPhaseAfter are the two launches of a call whose chunks are worked on beside its
// batches (kPack 1): the first has chunk_groups workgroups at the front of its grid
// that do chunk_kernel's work.
template <int kRows, int kPack, uint32_t kPhase>
__device__ __forceinline__ void fused_body(const uint8_t *blob, const uint64_t *batch_off, uint32_t n, const nisshi_columns &c,
                                           const Work &w, uint64_t rec_cap, uint64_t hdr_cap, const CrcTables *t,
                                           uint32_t prefixed, const uint64_t *before, uint32_t big_from,
                                           uint32_t chunk_groups) {
    // kPack waves a workgroup, each with a batch, a ring and a shape table of its
    // own: no wave waits for another (there is no barrier below)
    __shared__ __attribute__((aligned(16))) uint32_t rings[kPack][RingCfg<kRows>::kRingWords];
    __shared__ __attribute__((aligned(16))) uint32_t shape_tabs[kPack][kShapeWords];
    int lane = threadIdx.x & 63;
    uint32_t wave = uni(threadIdx.x >> 6);
    static_assert(kPhase == kPhaseAll || kPack == 1, "chunks beside batches: a wave a workgroup");
    // (chunks beside batches: the call's tag for the chunks' counts rides above the number of their workgroups)
    uint32_t tag = kPhase == kPhaseBeside ? chunk_groups >> 16 : 0u;
    if (kPhase == kPhaseBeside) chunk_groups &= 0xffffu;
    if (kPhase == kPhaseBeside && blockIdx.x < chunk_groups) {
        // The chunks of large batches are the largest units of work and start first,
        // the batches fill the machine up behind them.
        ChunkItems items;
        items.init(w, n, prefixed != 0, lane);
        for (uint32_t item = blockIdx.x; item < items.total; item += chunk_groups) {
            uint32_t cb, chunk;
            items.locate(w, item, lane, cb, chunk);
            if (uni(w.flags[cb]) & kFlagSeg)
                decode_unit<kRows>(true, cb, chunk, item, blob, batch_off, n, c, w, rec_cap, hdr_cap, t, prefixed, before,
                                   rings[0], shape_tabs[0], lane, kPhaseAll, tag);
            else
                crc_chunk_one(item, cb, chunk, blob, batch_off, c.batch_length, c.crc_computed, w, t, lane);
        }
        return;
    }
    uint32_t at = blockIdx.x - (kPhase == kPhaseBeside ? chunk_groups : 0u);
    uint32_t b = at * kPack + wave;
    if (big_from) {
        // Two passes over the batches in one grid (2 n workgroups, kPack 1): those of
        // big_from bytes and more first, the others behind them.  A wave takes as long
        // as its batch is large, and the launch ends when the last wave does: large
        // batches that start last would leave the machine waiting for them alone.
        uint32_t second = at >= n ? 1u : 0u;
        b = at - second * n;
        uint64_t size = batch_off[b + 1] - batch_off[b];
        if ((size < big_from ? 1u : 0u) != second) return;
    }
    if (b >= n) return;
#ifdef NISSHI_DIAG_WAVE_TIMES
    uint64_t diag_t0 = wall_clock64();
#endif
    decode_unit<kRows>(false, b, 0, 0, blob, batch_off, n, c, w, rec_cap, hdr_cap, t, prefixed, before, rings[wave],
                       shape_tabs[wave], lane, kPhase);
#ifdef NISSHI_DIAG_WAVE_TIMES
    // (diagnostic build: when this wave began, on the 100 MHz clock, and how long it took, over two header columns)
    if (lane == 0) {
        c.base_sequence[b] = (int32_t)diag_t0;
        c.partition_leader_epoch[b] = (int32_t)(wall_clock64() - diag_t0);
    }
#endif
}

// The columns as a kernel argument without their last pointer, enc_len, which goes
// behind the other arguments: a kernel that does not use it (the one of small
// batches) then has the arguments it had without it.  That kernel holds its
// arguments in scalar registers and is short of them: the pointer in their midst
// had it load sixteen registers more at a time and move others out of the way.
struct ColumnsArg {
    uint64_t words[offsetof(nisshi_columns, enc_len) / 8];
    ColumnsArg(const nisshi_columns &c) { __builtin_memcpy(words, &c, sizeof words); }
    __device__ __forceinline__ nisshi_columns with(uint32_t *enc_len) const {
        nisshi_columns c;
        __builtin_memcpy(&c, words, sizeof words);
        c.enc_len = enc_len;
        return c;
    }
};
static_assert(offsetof(nisshi_columns, enc_len) + 8 == sizeof(nisshi_columns), "enc_len is the last of the columns");

template <int kRows, int kWaves, int kPack, bool kBeside>
__global__ __launch_bounds__(64 * kPack) __attribute__((amdgpu_waves_per_eu(kWaves, kWaves))) void decode_fused_kernel(
    const uint8_t *blob, const uint64_t *batch_off, uint32_t n, ColumnsArg cols, Work w, uint64_t rec_cap, uint64_t hdr_cap,
    const CrcTables *t, uint32_t prefixed, const uint64_t *before, uint32_t big_from, uint32_t chunk_groups,
    uint32_t *enc_len) {
    nisshi_columns c = cols.with(kRows != 8 ? enc_len : nullptr);
    fused_body<kRows, kPack, kBeside ? kPhaseBeside : kPhaseAll>(blob, batch_off, n, c, w, rec_cap, hdr_cap, t, prefixed, before,
                                                                  big_from, chunk_groups);
}

// What a launch with chunks beside batches left: a wave a batch, of which those
// whose chunks were work items have something to do (kPhaseAfter).
template <int kRows, int kWaves>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kWaves, kWaves))) void decode_after_chunks_kernel(
    const uint8_t *blob, const uint64_t *batch_off, uint32_t n, nisshi_columns c, Work w, uint64_t rec_cap, uint64_t hdr_cap,
    const CrcTables *t, uint32_t prefixed, const uint64_t *before) {
    fused_body<kRows, 1, kPhaseAfter>(blob, batch_off, n, c, w, rec_cap, hdr_cap, t, prefixed, before, 0u, 0u);
}

// Batches of a few hundred bytes, a lane each: a workgroup a batch spends more on
// starting than on the batch (a producer that does not linger sends one record a
// batch).  The CRC bit by bit, the records by the sequential walk (exact_walk:
// Record::decode in order, the reference's outcomes by construction); their
// header slots were counted by batch_header_kernel, and the block sums are running
// sums by now.  The batch is then taken off the other kernels' hands (no CRC work item).
constexpr uint32_t kTinyBytes = 1024;

__global__ __launch_bounds__(256) void tiny_batch_kernel(const uint8_t *blob, const uint64_t *batch_off, uint32_t n,
                                                         nisshi_columns c, Work w, uint64_t rec_cap,
                                                         uint64_t hdr_cap, const uint64_t *before) {
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
        uint64_t sum = w.blk_sum[b / kSumBlock];
        uint32_t rec_base = (uint32_t)sum + w.rec_local[b] + (before ? (uint32_t)before[0] : 0u);
        uint32_t hdr_base = (uint32_t)(sum >> 32) + w.hdr_local[b] + (before ? (uint32_t)before[1] : 0u);
        int st = NISSHI_OK;
        uint32_t headers = 0;
        if ((uint64_t)rec_base + w.alloc_count[b] > rec_cap)
            st = NISSHI_CAPACITY;
        else
            st = exact_walk<true>(blob, b, c, w, rec_base, hdr_base, w.hdr_pred[b], hdr_cap, &headers);
        c.status[b] = st;
        c.hdr_count[b] = headers;
        if (flags & kFlagExact) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 3), 1ull);
        // (a failure is counted by decode_fused_kernel, which looks at every batch)
    }
    w.flags[b] = 0;
}

// What decode_fused_kernel left, a wave a batch of its list: batches that need the
// sequential walk's stores, and batches whose headers did not fit their expected
// slots, walked again (passing through the ring without a CRC) with slots behind
// those of all batches.
template <int kRows>
__global__ __launch_bounds__(64) void second_pass_kernel(const uint8_t *blob, const uint64_t *batch_off, uint32_t n,
                                                         nisshi_columns c, Work w, uint64_t rec_cap, uint64_t hdr_cap,
                                                         uint32_t prefixed) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[RingCfg<kRows>::kRingWords];
    __shared__ __attribute__((aligned(16))) uint32_t shape_tab[kShapeWords];
    int lane = threadIdx.x & 63;
    uint32_t entries = (uint32_t)w.totals[7];
    if (blockIdx.x == 0) leave_hints(w, n, prefixed != 0, lane, entries);
    for (uint32_t e = blockIdx.x; e < entries; e += gridDim.x) {
        uint32_t b = uni(w.list[e]);
        uint32_t flags = uni(w.flags[b]);
        uint32_t rec_base = uni(c.rec_base[b]);
        uint32_t expect = uni(w.hdr_pred[b]), need = uni(w.hdr_need[b]);
        uint32_t first = need ? extra_base(w, b, lane) : uni(c.hdr_base[b]);
        uint32_t have = need ? need : expect;
        if (flags & kFlagExact) {
            if (lane == 0) {
                uint32_t headers = 0;
                int st = exact_walk<true>(blob, b, c, w, rec_base, first, have, hdr_cap, &headers);
                c.hdr_count[b] = headers;
                c.hdr_base[b] = first;
                c.status[b] = st;
                if (st != NISSHI_OK) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), 1ull);
            }
            continue;
        }
        // (the batch has record slots: its first walk got through)
        BatchStream<kRows, false> stream;
        uint64_t lo = batch_off[b] + 21;
        stream.init(blob, lo, lo + (uint64_t)uni(c.batch_length[b]) - 9, (lds_u32 *)ring, lane);
        HeaderSlots slots;
        slots.init(first, have, hdr_cap);
        walk_batch(blob, b, c, rec_base, slots, stream, (lds_u128 *)shape_tab, lane);
        if (lane == 0) {
            c.hdr_base[b] = first;
            c.hdr_count[b] = slots.count;
            if (c.status[b] != NISSHI_OK) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), 1ull);
        }
        lds_order();
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
#include "inflate.inc"
namespace nisshi {
#include "deflate.inc"
}
#include "api.inc"
#include "pool.inc"
