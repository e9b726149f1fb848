// MI355X (gfx950) Kafka RecordBatch v2 codec: kernels and the C ABI of
// include/nisshi_hip.h.  Reference behaviour: nisshi-sans-io/src/record.rs,
// src/record/{deflated,inflated,codec,header}.rs, src/primitive/varint.rs.
//
// Decode pipeline (all on one stream):
//   batch_header_kernel   one lane per batch: the 61-byte header, its checks, the record_data span
//   scan_batches_kernel   exclusive scans over batches: record slots, CRC work items
//   crc_chunks_kernel     large batches only: one wave per 64 KiB, CRC-32C by sparse folding
//   crc_finish_kernel     combines the chunks of those batches
//   decode_fused_kernel   one wave per batch: CRC-32C streamed just ahead of the record walk;
//                         record starts by speculation, one lane per record, header spans inline
//   exact_kernel          sequential field walk for batches whose lengths disagree with their fields
//   finish_kernel         totals
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
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
constexpr uint32_t kFlagExact = 1u;  // needs the sequential walk

struct Work {
    uint32_t *alloc_count;  // record slots the batch gets
    uint32_t *n_chunks;     // CRC work items of the batch
    uint32_t *chunk_base;   // exclusive scan of n_chunks, n + 1 entries
    uint32_t *partial;      // one residue per CRC work item
    uint32_t *flags;
    uint64_t *totals;       // [0] record slots [1] header slots [2] failed [3] exact [4] out bytes [5] chunks
};

// ------------------------------------------------------------ batch header

// deflated::Batch::try_from(Bytes), deflated.rs:108-213, in the reference's order
// of checks; then Compression::try_from (lib.rs:1685-1698).
__global__ void batch_header_kernel(const uint8_t *blob, const uint64_t *batch_off, uint32_t n,
                                    int by_value, nisshi_columns c, Work w) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
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

// Exclusive scans over batches by one workgroup.  Batch counts are small next to
// record counts, so a single group walking 1024 entries at a time is enough.
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

__global__ void scan_batches_kernel(uint32_t n, nisshi_columns c, Work w) {
    __shared__ uint32_t lds[16];
    uint32_t rec_run = 0, chunk_run = 0;
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        uint32_t b = base + threadIdx.x;
        uint32_t cnt = b < n ? w.alloc_count[b] : 0;
        uint32_t chk = b < n ? w.n_chunks[b] : 0;
        uint32_t tot;
        uint32_t e = block_exclusive_scan(cnt, lds, tot);
        if (b < n) c.rec_base[b] = rec_run + e;
        rec_run += tot;
        e = block_exclusive_scan(chk, lds, tot);
        if (b < n) w.chunk_base[b] = chunk_run + e;
        chunk_run += tot;
    }
    if (threadIdx.x == 0) {
        w.chunk_base[n] = chunk_run;
        w.totals[0] = rec_run;
        w.totals[5] = chunk_run;
    }
}

// ------------------------------------------------------------ CRC-32C

// One wave per work item: item -> (batch, chunk) by binary search in the scan.
__global__ void crc_chunks_kernel(const uint8_t *blob, const uint64_t *batch_off,
                                  const int32_t *batch_length, uint32_t n, uint32_t *crc_out,
                                  Work w, const CrcTables *t, uint32_t min_chunks) {
    uint32_t item = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    uint32_t total = w.chunk_base[n];
    if (item >= total) return;
    uint32_t lo_b = 0, hi_b = n;  // largest b with chunk_base[b] <= item
    while (hi_b - lo_b > 1) {
        uint32_t mid = (lo_b + hi_b) >> 1;
        if (w.chunk_base[mid] <= item) lo_b = mid; else hi_b = mid;
    }
    uint32_t b = lo_b;
    uint32_t chunk = item - w.chunk_base[b];
    uint32_t chunks = w.n_chunks[b];
    if (chunks < min_chunks) return;  // small batches get their CRC from the fused walk
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

// Batches of more than one chunk: R = sum over chunks of R_c * x^(8 * bytes after chunk c).
__global__ void crc_finish_kernel(const uint64_t *batch_off, const int32_t *batch_length,
                                  uint32_t n, uint32_t *crc_out, Work w, const CrcTables *t,
                                  uint32_t min_chunks) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    uint32_t chunks = w.n_chunks[b];
    if (chunks < 2 || chunks < min_chunks) return;
    uint64_t lo = batch_off[b] + 21;
    uint64_t hi = lo + (uint64_t)batch_length[b] - 9;
    uint64_t grid = lo & ~uint64_t(15);
    uint64_t last_bytes = hi - (grid + (uint64_t)(chunks - 1) * kChunkBytes);
    const uint32_t *part = w.partial + w.chunk_base[b];
    uint32_t r = part[0];
    for (uint32_t k = 1; k + 1 < chunks; k++) r = gf_mulmod(r, t->chunk_weight) ^ part[k];
    r = gf_mulmod(r, gf_xpow8(t, last_bytes)) ^ part[chunks - 1];
    crc_out[b] = ~r;
}

// ------------------------------------------------------------ record walk

__device__ __forceinline__ void store_record(const nisshi_columns &c, uint32_t r,
                                             const RecordFields &f, uint32_t hdr_slot) {
    c.length[r] = f.length;
    c.rattributes[r] = (uint8_t)f.attributes;
    c.timestamp_delta[r] = f.timestamp_delta;
    c.offset_delta[r] = f.offset_delta;
    c.key_off[r] = f.key_off;
    c.key_len[r] = f.key_len;
    c.val_off[r] = f.val_off;
    c.val_len[r] = f.val_len;
    c.rhdr_base[r] = hdr_slot;
    c.rhdr_count[r] = f.hdr_count;
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Header slots are handed out from one cursor in runs, so that a batch's
// headers sit in a few contiguous pieces and a record's headers are always
// contiguous; the order of the pieces between batches is not defined.
constexpr uint32_t kHdrRunMax = 4096;

// The record walk of one batch by one wave.  Returns false when the batch has to
// be walked sequentially (exact_kernel): a length that disagrees with the fields
// it describes, or any decode error.
//
// Record starts are found up to 64 at a time by speculation: lane i assumes that
// the i-th next record starts i * S bytes on, S being the size of the record
// before, and decodes the length varint there.  Lanes up to and including the
// first one whose size differs from S did start on a record, and the walk
// resumes after that lane's record.  Batches of equal-sized records resolve 64
// records per step, ragged ones at least one.  Each resolved record is then
// decoded by its lane and must end where its length said.
//
// With `stream` the wave first pulls the rows it is about to parse through the
// CRC and into its LDS ring, and the parse reads LDS; records that do not fit
// the ring are read from global memory.
__device__ __forceinline__ bool walk_batch(const uint8_t *blob, uint32_t b, const nisshi_columns &c,
                                           const Work &w, uint64_t hdr_cap, BatchStream &stream,
                                           bool streaming, int lane) {
    uint32_t count = uni(c.record_count[b]);
    uint32_t rbase = uni(c.rec_base[b]);
    uint64_t data_off = uni64(c.data_off[b]);
    Reader rd(blob + data_off, uni(c.data_len[b]));
    if (streaming) {
        rd.ring = stream.ring;
        rd.ring_bias = (uint32_t)(data_off - stream.grid);
        rd.ring_mask = kRingBytes - 1;
    }
    unsigned long long *hdr_cursor = reinterpret_cast<unsigned long long *>(w.totals + 1);
    uint64_t run_pos = 0, run_end = 0;  // header slots in hand
    uint32_t hdr_total = 0, first_slot = 0;
    uint32_t pos = 0;
    uint32_t S = count ? rd.len / count : 0;
    // with a ring, speculate only as far as the ring reaches
    constexpr uint32_t kReach = kRingBytes - 2 * kRowBytes;
    for (uint32_t k = 0; k < count;) {
        uint32_t m = count - k < (uint32_t)kWave ? count - k : (uint32_t)kWave;
        uint32_t found = 0, my_start = 0, my_size = 0, my_used = 0;
        int32_t my_len = 0;
        uint32_t round_start = pos;
        while (found < m) {
            uint32_t lanes = m - found;
            if (streaming) {
                if ((uint64_t)lanes * S > kReach) lanes = S ? kReach / S : 1;
                if (lanes == 0) lanes = 1;
                if (pos + (uint64_t)lanes * S + 64 - round_start > kReach && found) break;
                stream.advance(stream.row_after(data_off + pos + (uint64_t)lanes * S + 64));
                __syncthreads();
                uint64_t from = stream.staged_from(), to = stream.staged_to();
                rd.win_lo = from > data_off ? (uint32_t)(from - data_off) : 0;
                rd.win_hi = to > data_off ? (uint32_t)(to - data_off) : 0;
            }
            uint64_t guess64 = (uint64_t)pos + (uint64_t)lane * S;
            bool in = (uint32_t)lane < lanes && guess64 < rd.len;
            uint32_t guess = in ? (uint32_t)guess64 : 0;
            int32_t len = 0;
            uint32_t used = 0;
            bool ok = in && fast_varint(rd.get64(guess), used, len);
            // the whole record lies inside the data (this also catches a varint cut short)
            ok = ok && len >= 0 && (uint64_t)guess + used + (uint32_t)len <= rd.len;
            uint32_t size = ok ? used + (uint32_t)len : 0;
            uint64_t okmask = __ballot(ok);
            if (!(okmask & 1)) return false;
            uint64_t same = __ballot(ok && size == S);
            uint32_t good = ~same ? (uint32_t)__ffsll((long long)~same) - 1 : 64;
            uint32_t v = good;  // lanes below `good` are records of size S
            if (good < 64 && ((okmask >> good) & 1)) v = good + 1;
            if (v > lanes) v = lanes;
            v = uni(v);
            if (found == 0) {  // the usual case: the records stay in the lanes that found them
                if ((uint32_t)lane < v) {
                    my_start = guess;
                    my_size = size;
                    my_used = used;
                    my_len = len;
                }
            } else {
                int src = (lane - (int)found) & 63;
                uint32_t s_start = __shfl(guess, src, 64);
                uint32_t s_size = __shfl(size, src, 64);
                uint32_t s_used = __shfl(used, src, 64);
                int32_t s_len = __shfl(len, src, 64);
                if ((uint32_t)lane >= found && (uint32_t)lane < found + v) {
                    my_start = s_start;
                    my_size = s_size;
                    my_used = s_used;
                    my_len = s_len;
                }
            }
            S = (uint32_t)__builtin_amdgcn_readlane((int)size, (int)v - 1);
            pos = (uint32_t)__builtin_amdgcn_readlane((int)guess, (int)v - 1) + S;
            found += v;
        }
        m = found;
        // start fetching what the next round will parse while this one is decoded
        if (streaming) {
            uint32_t next = count - k - m < (uint32_t)kWave ? count - k - m : (uint32_t)kWave;
            if ((uint64_t)next * S > kReach) next = S ? kReach / S : 1;
            if (next) stream.prefetch(stream.row_after(data_off + pos + (uint64_t)next * S + 64));
        }
        // ---- one lane per record: the fields after the length
        RecordFields f;
        bool live = (uint32_t)lane < m && !(stream.dbg & 4);
        uint32_t end = my_start + my_size;
        uint32_t p = my_start + my_used;
        bool ok = live;
        f.length = my_len;
        f.attributes = 0;
        f.timestamp_delta = 0;
        f.offset_delta = 0;
        f.key_off = f.val_off = 0;
        f.key_len = f.val_len = -1;
        f.hdr_count = 0;
        if (live) {
            uint32_t used;
            uint64_t w0 = rd.get64(p);  // attributes, then the timestamp delta
            f.attributes = (uint32_t)w0 & 0xffu;
            if (fast_varlong((uint32_t)(w0 >> 8), used, f.timestamp_delta)) {
                p += 1 + used;
            } else {
                p += 1;
                ok = rd.varlong(p, f.timestamp_delta) == NISSHI_OK;
            }
            ok = fast_varint(rd.get64(p), used, f.offset_delta) && ok;
            p += used;
            ok = fast_varint(rd.get64(p), used, f.key_len) && ok;
            p += used;
            f.key_off = p;
            ok = ok && f.key_len >= -1 && p <= end && (uint32_t)(f.key_len < 0 ? 0 : f.key_len) <= end - p;
            p += ok && f.key_len > 0 ? (uint32_t)f.key_len : 0u;
            ok = fast_varint(rd.get64(p), used, f.val_len) && ok;
            p += used;
            f.val_off = p;
            ok = ok && f.val_len >= -1 && p <= end && (uint32_t)(f.val_len < 0 ? 0 : f.val_len) <= end - p;
            p += ok && f.val_len > 0 ? (uint32_t)f.val_len : 0u;
            ok = fast_varint(rd.get64(p), used, f.hdr_count) && ok;
            p += used;
            // a header is at least two bytes: a count the record cannot hold is an error
            ok = ok && f.hdr_count >= 0 && p <= end && (uint32_t)f.hdr_count <= (end - p) / 2;
        }
        uint32_t hc = ok && !(stream.dbg & 8) ? (uint32_t)f.hdr_count : 0;
        uint32_t slot = 0;
        uint32_t total = 0;
        if (__any(hc != 0)) {
            // equal header counts in every record (the usual case) need no scan
            uint32_t hc0 = uni(hc);
            uint32_t inc = __all(hc == (live ? hc0 : 0u)) ? (live ? ((uint32_t)lane + 1) * hc0 : m * hc0)
                                                           : wave_inclusive_scan(hc, lane);
            total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            // records whose headers fit in the run in hand use it, the rest a new run
            uint64_t room = run_end - run_pos;
            bool fits = inc <= room;
            uint64_t fit_mask = __ballot(fits);
            uint32_t n_fit = fit_mask == ~0ull ? 64 : (uint32_t)__ffsll((long long)~fit_mask) - 1;
            uint32_t used_old = n_fit ? (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)n_fit - 1) : 0;
            uint32_t rest = total - used_old;
            uint64_t new_base = 0;
            uint32_t grab = 0;
            if (rest) {
                // take enough for the records still to come at this round's density
                uint64_t ahead = (uint64_t)total * (count - k - m) / m;
                grab = rest + (uint32_t)(ahead < kHdrRunMax ? ahead : kHdrRunMax);
                if (lane == 0) new_base = atomicAdd(hdr_cursor, (unsigned long long)grab);
                new_base = uni64(new_base);
                if (new_base + grab > hdr_cap) {
                    if (lane == 0) c.status[b] = NISSHI_CAPACITY;
                    return true;
                }
            }
            slot = (uint32_t)(fits ? run_pos + (inc - hc) : new_base + (inc - hc - used_old));
            if (rest) {
                run_pos = new_base + rest;
                run_end = new_base + grab;
            } else {
                run_pos += total;
            }
            if (hdr_total == 0)
                first_slot = (uint32_t)__builtin_amdgcn_readlane((int)slot, __ffsll((long long)__ballot(hc > 0)) - 1);
            hdr_total += total;
            for (uint32_t j = 0; j < hc && ok; j++) {
                uint32_t used, ko, vo;
                int32_t kl, vl;
                ok = fast_varint(rd.get64(p), used, kl);
                p += used;
                ko = p;
                ok = ok && kl >= -1 && p <= end && (uint32_t)(kl < 0 ? 0 : kl) <= end - p;
                p += ok && kl > 0 ? (uint32_t)kl : 0u;
                ok = fast_varint(rd.get64(p), used, vl) && ok;
                p += used;
                vo = p;
                ok = ok && vl >= -1 && p <= end && (uint32_t)(vl < 0 ? 0 : vl) <= end - p;
                p += ok && vl > 0 ? (uint32_t)vl : 0u;
                if (ok) {
                    c.hkey_off[slot + j] = ko;
                    c.hkey_len[slot + j] = kl;
                    c.hval_off[slot + j] = vo;
                    c.hval_len[slot + j] = vl;
                }
            }
        }
        // the fields must end exactly where the length said
        bool bad = live && (!ok || p != end) && !(stream.dbg & 8);
        if (__any(bad)) return false;
        if (live) store_record(c, rbase + k + lane, f, slot);
        k += m;
        __syncthreads();  // every lane is done with the ring before it moves on
    }
    if (lane == 0) {
        c.hdr_count[b] = hdr_total;
        c.hdr_base[b] = first_slot;
    }
    return true;
}

// One wave per batch: CRC-32C of the batch streamed just ahead of the record
// walk through an LDS ring, so the batch is read from HBM once.  Batches of
// kBigChunks chunks or more get their CRC from crc_chunks_kernel instead (one
// wave cannot stream them fast enough) and are parsed from global memory.
constexpr uint32_t kBigChunks = 8;

__global__ __launch_bounds__(64) void decode_fused_kernel(const uint8_t *blob,
                                                          const uint64_t *batch_off, uint32_t n,
                                                          nisshi_columns c, Work w, uint64_t rec_cap,
                                                          uint64_t hdr_cap, const CrcTables *t, int dbg) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[kRingWords];
    uint32_t b = blockIdx.x;
    int lane = threadIdx.x;
    if (b >= n) return;
    uint32_t chunks = uni(w.n_chunks[b]);
    bool do_crc = chunks >= 1 && chunks < kBigChunks;
    bool do_walk = uni(c.status[b]) == NISSHI_OK && !(uni(w.flags[b]) & kFlagExact) && !(dbg & 1);
    if (do_walk && (uint64_t)c.rec_base[b] + c.record_count[b] > rec_cap) {
        if (lane == 0) c.status[b] = NISSHI_CAPACITY;
        do_walk = false;
    }
    BatchStream stream;
    stream.dbg = dbg;
    uint64_t lo = batch_off[b] + 21;
    stream.init(blob, lo, do_crc ? lo + (uint64_t)uni(c.batch_length[b]) - 9 : lo, (lds_u32 *)ring, lane);
    if (do_walk && !walk_batch(blob, b, c, w, hdr_cap, stream, do_crc, lane)) {
        if (lane == 0) w.flags[b] |= kFlagExact;
    }
    if (do_crc) {
        uint32_t v = stream.finish(t);
        if (lane == 0) c.crc_computed[b] = v;
    }
}

// One lane per flagged batch: Record::decode in sequence, exactly as
// Vec<Record>::try_from(deflated::Batch) does (deflated.rs:431-442).  The first
// error ends the batch with its status; records are stored only while the batch
// has slots (a batch whose count cannot fit has none and only gets its status).
__global__ void exact_kernel(const uint8_t *blob, uint32_t n, nisshi_columns c, Work w,
                             uint64_t rec_cap, uint64_t hdr_cap) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    if (!(w.flags[b] & kFlagExact) || c.status[b] != NISSHI_OK) return;
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
                c.hkey_off[slot + j] = ko;
                c.hkey_len[slot + j] = kl;
                c.hval_off[slot + j] = vo;
                c.hval_len[slot + j] = vl;
            }
        }
        if (st) break;
        if (k < slots) store_record(c, rbase + k, f, (uint32_t)slot);
        hdr_total += (uint32_t)f.hdr_count;
    }
    c.status[b] = st;
    c.hdr_count[b] = st == NISSHI_OK ? hdr_total : 0;
    c.hdr_base[b] = first_slot;
}

// Totals over batches once every status is final.
__global__ void finish_kernel(uint32_t n, nisshi_columns c, Work w) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    bool failed = b < n && c.status[b] != NISSHI_OK;
    bool exact = b < n && (w.flags[b] & kFlagExact);
    uint64_t fm = __ballot(failed), em = __ballot(exact);
    if ((threadIdx.x & 63) == 0) {
        if (fm) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 2), (unsigned long long)__popcll(fm));
        if (em) atomicAdd(reinterpret_cast<unsigned long long *>(w.totals + 3), (unsigned long long)__popcll(em));
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
#include "api.inc"
