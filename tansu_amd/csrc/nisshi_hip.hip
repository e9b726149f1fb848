// MI355X (gfx950) Kafka RecordBatch v2 codec: kernels and the C ABI of
// include/nisshi_hip.h.  Reference behaviour: nisshi-sans-io/src/record.rs,
// src/record/{deflated,inflated,codec,header}.rs, src/primitive/varint.rs.
//
// Decode pipeline (all on one stream):
//   batch_header_kernel   one lane per batch: the 61-byte header, its checks, the record_data span
//   scan_batches_kernel   exclusive scans over batches: record slots, CRC work items
//   crc_chunks_kernel     one wave per 64 KiB of batch: CRC-32C by sparse folding
//   crc_finish_kernel     combines the chunks of batches larger than one chunk
//   walk_kernel           one wave per batch: hop over the length varints, then one lane per record
//   exact_kernel          sequential field walk for batches whose lengths disagree with their fields
//   scan_headers_kernel   exclusive scan of header counts over batches
//   headers_kernel        one wave per batch: header spans into their slots
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

__global__ void scan_headers_kernel(uint32_t n, nisshi_columns c, Work w) {
    __shared__ uint32_t lds[16];
    uint32_t run = 0, failed = 0, exact = 0;
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        uint32_t b = base + threadIdx.x;
        bool ok = b < n && c.status[b] == NISSHI_OK;
        uint32_t cnt = ok ? c.hdr_count[b] : 0;
        uint32_t tot;
        uint32_t e = block_exclusive_scan(cnt, lds, tot);
        if (b < n) c.hdr_base[b] = run + e;
        run += tot;
        block_exclusive_scan(b < n && !ok ? 1u : 0u, lds, tot);
        failed += tot;
        block_exclusive_scan(b < n && (w.flags[b] & kFlagExact) ? 1u : 0u, lds, tot);
        exact += tot;
    }
    if (threadIdx.x == 0) {
        w.totals[1] = run;
        w.totals[2] = failed;
        w.totals[3] = exact;
    }
}

// ------------------------------------------------------------ CRC-32C

// One wave per work item: item -> (batch, chunk) by binary search in the scan.
__global__ void crc_chunks_kernel(const uint8_t *blob, const uint64_t *batch_off,
                                  const int32_t *batch_length, uint32_t n, uint32_t *crc_out,
                                  Work w, const CrcTables *t) {
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
                                  uint32_t n, uint32_t *crc_out, Work w, const CrcTables *t) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    uint32_t chunks = w.n_chunks[b];
    if (chunks < 2) return;
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
                                             const RecordFields &f, uint32_t hdr_rel) {
    c.length[r] = f.length;
    c.rattributes[r] = (uint8_t)f.attributes;
    c.timestamp_delta[r] = f.timestamp_delta;
    c.offset_delta[r] = f.offset_delta;
    c.key_off[r] = f.key_off;
    c.key_len[r] = f.key_len;
    c.val_off[r] = f.val_off;
    c.val_len[r] = f.val_len;
    c.rhdr_base[r] = hdr_rel;  // relative to the batch until headers_kernel adds hdr_base
    c.rhdr_count[r] = f.hdr_count;
}

// One wave per batch.  Phase 1 hops over the length varints to find up to 64
// record starts; phase 2 gives each start to a lane, which decodes the fields
// and checks that they end where the length said.  The reference never uses the
// length to delimit a record (record.rs:259-265), so any disagreement, and any
// decode error, hands the batch to exact_kernel, which walks field by field.
__global__ void walk_kernel(const uint8_t *blob, uint32_t n, nisshi_columns c, Work w,
                            uint64_t rec_cap) {
    uint32_t b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (b >= n) return;
    if (c.status[b] != NISSHI_OK || (w.flags[b] & kFlagExact)) return;
    uint32_t count = c.record_count[b];
    uint32_t rbase = c.rec_base[b];
    if ((uint64_t)rbase + count > rec_cap) {
        if (lane == 0) c.status[b] = NISSHI_CAPACITY;
        return;
    }
    Cursor hop{blob + c.data_off[b], c.data_len[b], 0};
    uint32_t hdr_run = 0;
    for (uint32_t k = 0; k < count; k += kWave) {
        uint32_t m = count - k < (uint32_t)kWave ? count - k : (uint32_t)kWave;
        uint32_t my_start = 0, my_end = 0;
        bool bad = false;
        for (uint32_t i = 0; i < m; i++) {
            uint32_t start = hop.pos;
            int32_t len;
            int st = read_varint(hop, len);
            if (st != NISSHI_OK || len < 0 || (uint32_t)len > hop.len - hop.pos) {
                bad = true;
                break;
            }
            hop.pos += (uint32_t)len;
            if ((uint32_t)lane == i) {
                my_start = start;
                my_end = hop.pos;
            }
        }
        RecordFields f;
        if (!bad && (uint32_t)lane < m) {
            Cursor cur{hop.data, hop.len, my_start};
            int st = read_record(cur, f);
            bad = st != NISSHI_OK || cur.pos != my_end;
        }
        if (__any(bad)) {
            if (lane == 0) w.flags[b] |= kFlagExact;
            return;
        }
        uint32_t hc = (uint32_t)lane < m ? (uint32_t)f.hdr_count : 0;
        uint32_t inc = hc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if ((uint32_t)lane < m) store_record(c, rbase + k + lane, f, hdr_run + inc - hc);
        hdr_run += __shfl(inc, 63, 64);
    }
    if (lane == 0) c.hdr_count[b] = hdr_run;
}

// One lane per flagged batch: Record::decode in sequence, exactly as
// Vec<Record>::try_from(deflated::Batch) does (deflated.rs:431-442).  The first
// error ends the batch with its status; records are stored only while the batch
// has slots (a batch whose count cannot fit has none and only gets its status).
__global__ void exact_kernel(const uint8_t *blob, uint32_t n, nisshi_columns c, Work w,
                             uint64_t rec_cap) {
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
    Cursor cur{blob + c.data_off[b], c.data_len[b], 0};
    uint32_t hdr_run = 0;
    int st = NISSHI_OK;
    for (uint32_t k = 0; k < count; k++) {
        RecordFields f;
        st = read_record(cur, f);
        if (st) break;
        if (k < slots) store_record(c, rbase + k, f, hdr_run);
        hdr_run += (uint32_t)f.hdr_count;
    }
    c.status[b] = st;
    c.hdr_count[b] = st == NISSHI_OK ? hdr_run : 0;
}

// One wave per batch, one lane per record: header spans into the slots
// hdr_base[b] + (the record's offset within the batch).
__global__ void headers_kernel(const uint8_t *blob, uint32_t n, nisshi_columns c, uint64_t hdr_cap) {
    uint32_t b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (b >= n) return;
    if (c.status[b] != NISSHI_OK) return;
    uint32_t hbase = c.hdr_base[b];
    uint32_t count = c.record_count[b];
    uint32_t rbase = c.rec_base[b];
    if ((uint64_t)hbase + c.hdr_count[b] > hdr_cap) {
        if (lane == 0) c.status[b] = NISSHI_CAPACITY;
        return;
    }
    const uint8_t *data = blob + c.data_off[b];
    uint32_t len = c.data_len[b];
    for (uint32_t k = lane; k < count; k += kWave) {
        uint32_t r = rbase + k;
        int32_t hc = c.rhdr_count[r];
        uint32_t slot = hbase + c.rhdr_base[r];
        c.rhdr_base[r] = slot;
        if (hc <= 0) continue;
        int32_t vl = c.val_len[r];
        Cursor cur{data, len, c.val_off[r] + (vl > 0 ? (uint32_t)vl : 0u)};
        int32_t again;
        read_varint(cur, again);  // the count, already validated by the walk
        for (int32_t j = 0; j < hc; j++) {
            uint32_t ko, vo;
            int32_t kl, vlen;
            read_octets(cur, ko, kl);
            read_octets(cur, vo, vlen);
            c.hkey_off[slot + j] = ko;
            c.hkey_len[slot + j] = kl;
            c.hval_off[slot + j] = vo;
            c.hval_len[slot + j] = vlen;
        }
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
