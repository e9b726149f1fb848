/*
 * nisshi_hip.h: C ABI of the MI355X (gfx950) Kafka RecordBatch v2 codec.
 *
 * This is the boundary a `nisshi-hip` FFI crate binds: plain pointers and
 * sizes, no C++ or torch types.  It replaces, for uncompressed batches, the
 * record path of nisshi-sans-io (file:line below are paths in the tansu /
 * nisshi reference tree):
 *
 *   deflated::Batch::try_from(Bytes)                src/record/deflated.rs:108-213
 *   Vec<Record>::try_from(deflated::Batch)          src/record/deflated.rs:421-459
 *   inflated::Batch::try_from(deflated::Batch)      src/record/inflated.rs:114-150
 *   inflated::Batch::try_from(&deflated::Batch)     src/record/inflated.rs:152-187
 *   inflated::Frame::try_from(deflated::Frame)      src/record/inflated.rs:34-62
 *   deflated::Batch::try_from(inflated::Batch)      src/record/deflated.rs:392-409
 *   deflated::Frame::try_from(inflated::Frame)      src/record/deflated.rs:52-63
 *   From<deflated::Batch> for Bytes                 src/record/deflated.rs:84-106
 *   records-blob framing                            src/de.rs:701-720, 1056-1093; src/ser.rs:484-489, 521-574
 *
 * and, as operators on the decoded columns, four callers of that path in
 * nisshi-storage:
 *
 *   the produce seam of the SQL engines             nisshi-storage/src/pg.rs:797-943, lite.rs:703-815
 *   combine (the produce queue)                     nisshi-storage/src/batch.rs:661-709
 *   the fetch seam of the SQL engines               nisshi-storage/src/pg.rs:1875-2052, lite.rs:2765-2980
 *   log compaction                                  nisshi-storage/src/slate/storage.rs:283-312
 *
 * Decoded batches are columns, not Vec<Record>.  Span
 * offsets are relative to the batch's record_data, mirroring the zero-copy
 * Bytes::split_to slices of the reference (src/record/codec.rs:65-76): a null
 * Option<Bytes> has length -1 and the offset where its bytes would start.
 *
 * Threading: a context may be used by one thread at a time; create one per
 * thread (or guard it).  Calls enqueue work on the given HIP stream.
 */
#ifndef NISSHI_HIP_H
#define NISSHI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-batch and per-call status.  The first seven mirror the outcomes of the
 * reference: its Error variants (src/lib.rs:328-366) and its panics. */
enum {
    NISSHI_OK = 0,
    NISSHI_TRY_GET = 1,             /* Error::TryGet: buffer under-run */
    NISSHI_OVERFLOW = 2,            /* Error::Overflow */
    NISSHI_TRY_FROM_INT = 3,        /* Error::TryFromInt: negative batch_length */
    NISSHI_UNKNOWN_COMPRESSION = 4, /* Error::UnknownCompressionType: attributes & 7 in 5..7 */
    NISSHI_MALFORMED = 5,           /* the reference panics on this input */
    NISSHI_COMPRESSED = 6,          /* attributes & 7 in 1..4: not decoded here */
    NISSHI_CAPACITY = 7,            /* an output array is too small */
    NISSHI_DEVICE_ERROR = 8,        /* a HIP call failed; see nisshi_hip_last_error */
    NISSHI_INVALID_ARGUMENT = 9,
    NISSHI_NONE = 10,               /* no batch: combine of no batches (Ok(None), nisshi-storage/src/batch.rs:666-668);
                                       compaction leaves nothing to write for this batch; inflate passes this one through */
    NISSHI_INFLATE = 11             /* the compressed record data does not inflate (the inflator's Error::Snap / Error::Io) */
};

/* Decoded records are columns, grouped so that the fields the decoder produces
 * together are stored together: per record a 16-byte key group, a 16-byte value
 * group, the timestamp delta and the attributes (41 bytes, record.rs:172-198),
 * per header one 16-byte group (header.rs:21-29). */
typedef struct nisshi_record_key {
    int32_t length; /* as on the wire; never used to delimit the record */
    int32_t offset_delta;
    uint32_t key_off;
    int32_t key_len; /* -1: None */
} nisshi_record_key;

typedef struct nisshi_record_value {
    uint32_t val_off;
    int32_t val_len;   /* -1: None */
    uint32_t hdr_base; /* slot of the record's first header; a record's headers are contiguous */
    int32_t hdr_count;
} nisshi_record_value;

typedef struct nisshi_header_span {
    uint32_t key_off;
    int32_t key_len; /* -1: None */
    uint32_t val_off;
    int32_t val_len; /* -1: None */
} nisshi_header_span;

/* One pointer per column.  For the *_dev entry points these are device
 * pointers; for the host entry points host pointers.  Field order is ABI. */
typedef struct nisshi_columns {
    /* per batch: the 13 fields of the 61-byte header (deflated.rs:67-82) */
    int64_t *base_offset;
    int32_t *batch_length;
    int32_t *partition_leader_epoch;
    int8_t *magic;
    uint32_t *crc; /* the wire value, also when it does not match */
    int16_t *attributes;
    int32_t *last_offset_delta;
    int64_t *base_timestamp;
    int64_t *max_timestamp;
    int64_t *producer_id;
    int16_t *producer_epoch;
    int32_t *base_sequence;
    uint32_t *record_count;
    /* per batch: derived */
    uint32_t *crc_computed; /* CRC-32C over attributes..end of batch (deflated.rs:155-166) */
    uint64_t *data_off;     /* offset of record_data in the blob */
    uint32_t *data_len;
    uint32_t *rec_base;  /* index of the batch's first record in the record columns */
    uint32_t *hdr_base;  /* slot of the batch's first header; its headers are one run, in record order */
    uint32_t *hdr_count; /* headers in the batch */
    int32_t *status;
    /* per record */
    nisshi_record_key *rkey;
    nisshi_record_value *rval;
    int64_t *timestamp_delta;
    uint8_t *rattributes;
    /* per header */
    nisshi_header_span *hdr;
    /* per batch, optional (may be NULL): one more than the bytes the batch's records encode to as
     * they stand, 0 when that is not known.  Record::size_in_bytes summed over the batch
     * (nisshi-sans-io/src/record.rs:200-231) is what the encode works out first; a decode leaves it
     * here for the batches whose records it found written the shortest way (every varint minimal, as
     * producers write them: the records then encode to the bytes they were decoded from), and
     * nisshi_hip_encode_dev takes it from here instead of sizing those records again.  It is a
     * hint: the decode notes it in calls whose batches average NISSHI_WIDE_RING_FROM bytes and
     * more (see nisshi_hip_ctx_create), for batches walked by their records' shapes, whole or
     * chunk by chunk; any other batch has 0 and is sized by the encode as before.  The fetch
     * seam, which works out every record's length anyway, leaves the sum for its batches of
     * fewer than 2048 records; combine and compact store 0 for the batches they write.  A caller that changes a
     * batch's records by hand stores 0 too (or passes NULL): the encode trusts the value, and one
     * that is less than the records take makes it write past the batch's place in the output. */
    uint32_t *enc_len;
} nisshi_columns;

typedef struct nisshi_totals {
    uint64_t n_records;  /* record slots used (a failed batch may leave its slots unused) */
    uint64_t n_headers;  /* header slots in use or set aside: at least the number of headers (see nisshi_hip_decode_dev) */
    uint32_t n_failed;   /* batches whose status is not NISSHI_OK */
    uint32_t n_exact;    /* batches that needed the sequential walk (length disagreed with the fields) */
    uint64_t out_bytes;  /* encode: bytes written */
} nisshi_totals;

typedef struct nisshi_hip_ctx nisshi_hip_ctx;

/* The layout of the structs of this header as a number that grows with every
 * change to them (2: nisshi_columns ends in enc_len).  A binding checks the
 * library it loaded against the header it was built from before any other call:
 * a struct one field short makes the library read a pointer that is not there. */
#define NISSHI_HIP_ABI_VERSION 2
int nisshi_hip_abi_version(void);

/* Create a context on a HIP device; fails with NISSHI_DEVICE_ERROR when no
 * usable gfx950 device is present.  There is no CPU fallback.  Five settings are
 * read from the environment here: NISSHI_SLICE_BYTES (the slice of the
 * host-buffer decode, see nisshi_hip_decode), NISSHI_WIDE_RING_FROM (the mean
 * batch size in bytes from which the decode kernel streams a batch through
 * twelve 1 KiB rows of LDS at three waves a SIMD instead of eight at four;
 * default 65536) and NISSHI_SEG_FEW_BATCHES (calls of fewer batches than this
 * decode their batches of 128 KiB and more a wave a 64 KiB chunk, calls of more
 * only those of 512 KiB and more; default 6144), NISSHI_CHUNKS_BESIDE (0: the
 * chunks of large batches get a launch of their own ahead of the batches' in
 * every call; default 1: calls of few large batches work on them inside the
 * batches' launch) and NISSHI_BIG_FIRST (0: the batches start in the order they
 * lie in; default 1: in calls of large batches those above the mean size start
 * first).  None changes any result. */
int nisshi_hip_ctx_create(int device, nisshi_hip_ctx **ctx);
void nisshi_hip_ctx_destroy(nisshi_hip_ctx *ctx);
const char *nisshi_hip_status_str(int status);
const char *nisshi_hip_last_error(const nisshi_hip_ctx *ctx);
/* Of the context's last decode that handed back totals: the batches that were
 * decoded a wave a 64 KiB chunk on guesses that did not hold (see the decode,
 * below) and so were walked once more, whole, by one wave.  Same columns either
 * way; a number that stays high says the call's large batches are not what the
 * chunks' guesses expect.  (Counted only in calls of fewer than
 * NISSHI_SEG_FEW_BATCHES batches of NISSHI_WIDE_RING_FROM bytes and more on
 * average, where the chunks are worked on beside the batches.)  No counterpart
 * in the reference. */
uint64_t nisshi_hip_last_walked_whole(const nisshi_hip_ctx *ctx);

/* Host-side framing of a records blob (de.rs:1056-1093): writes the start of
 * every batch plus the end of the last one (n + 1 entries). */
int nisshi_frame_split(const uint8_t *blob, uint64_t len, uint64_t *batch_off, uint64_t cap,
                       uint64_t *n_batches);
/* records field length prefix: u32 big-endian, or uvarint(len + 1) in flexible
 * versions; length 0 is a null field (de.rs:701-720, ser.rs:484-489, 553-560) */
int nisshi_records_prefix_decode(const uint8_t *buf, uint64_t len, int flexible, uint64_t *pos,
                                 uint32_t *records_len, int *is_null);
uint64_t nisshi_records_prefix_encode(uint32_t records_len, int flexible, uint8_t *out);
/* The records fields of one receive buffer in one call (a produce request holds one
 * a partition: nisshi-sans-io/src/de.rs:701-720 reads the prefix, de.rs:1056-1093
 * the batches).  field_pos[i]: where field i's length prefix starts in buf, which
 * the request parser knows as it walks the request.  Decodes each prefix, splits
 * each field (nisshi_frame_split) and writes the starts of all batches, field after
 * field, as offsets into buf: batch_off gets *n_batches + 1 entries (cap is its
 * size), the last one the end of the last field with batches.  first_batch (or NULL;
 * n_fields + 1 entries): the index of each field's first batch; a null or empty
 * field has none.  buf and batch_off can go to nisshi_hip_decode as they are: the
 * extent of a field's last batch then runs on to the first batch of the next
 * field, and what lies behind a batch's batch_length is never looked at. */
int nisshi_records_fields_split(const uint8_t *buf, uint64_t len, const uint64_t *field_pos, uint32_t n_fields,
                                int flexible, uint64_t *batch_off, uint64_t cap, uint32_t *first_batch,
                                uint64_t *n_batches);

/* Sum of record_count over the batches whose header is well formed and
 * uncompressed: the record capacity a decode of this blob needs.  Device
 * pointers; synchronises the stream. */
int nisshi_hip_count_dev(nisshi_hip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_len,
                         const uint64_t *d_batch_off, uint32_t n_batches, uint64_t *n_records,
                         void *stream);

/* Decode n_batches batches resident in device memory into device columns.
 * d_batch_off has n_batches + 1 entries.  by_value != 0 clears the compression
 * bits of attributes and drops unknown bits as inflated::Batch::try_from(
 * deflated::Batch) does; 0 keeps attributes verbatim (the by-reference
 * conversion).  With totals == NULL the call only enqueues work; otherwise it
 * waits for the stream and fills *totals.  A batch that fails leaves its
 * status and no valid records; the call still returns NISSHI_OK.
 *
 * Where records and headers go does not depend on the order the batches decode
 * in: two decodes of a blob give the same columns, byte for byte.  A batch's
 * records start at the sum of the record counts of the decodable batches before
 * it (rec_base).  A batch is expected to have as many headers a record as its
 * first record has; its headers start at the sum of what the batches before it
 * are expected to have (hdr_base), in record order, one run.  A batch with more
 * headers than that gets all of them in one run behind the expected slots of all
 * batches, again in batch order, and leaves its expected slots unused
 * (totals.n_headers counts both).  So a call whose batches have the same number
 * of headers in every record needs exactly its header count of slots, and
 * blob_len + 1 slots cover every call (a header is at least two bytes).  A batch
 * whose records or headers do not fit rec_cap or hdr_cap gets NISSHI_CAPACITY.
 *
 * d_blob must be 16-byte aligned, as must d_cols->rkey, rval and hdr (they are
 * read and written sixteen bytes at a time) and timestamp_delta 8-byte aligned;
 * the call returns NISSHI_INVALID_ARGUMENT otherwise.
 *
 * A context serves one call at a time: its workspace is the call's.  The *_dev
 * entry points may be given any stream, and with totals == NULL they return
 * while their kernels are queued; the next call on the context is then ordered
 * behind that work on the device, also when it is given another stream (an event
 * the earlier call left is waited for by the later call's stream), and a call that
 * has to grow the workspace waits for the device first.  Calls on one context from
 * several threads must be serialised by the caller; use a context a thread (or
 * a nisshi_hip_pool) to decode concurrently. */
int nisshi_hip_decode_dev(nisshi_hip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_len,
                          const uint64_t *d_batch_off, uint32_t n_batches, int by_value,
                          const nisshi_columns *d_cols, uint64_t rec_cap, uint64_t hdr_cap,
                          nisshi_totals *totals, void *stream);

/* nisshi_hip_decode_dev for callers that cannot allocate device memory: code
 * under HIP graph capture, serving loops that must not call hipMalloc, harnesses
 * that forbid it.  It has no counterpart in the reference.  Same arguments and
 * results as nisshi_hip_decode_dev (whose kernels it shares), but the workspace
 * (per-batch work arrays, the residues of CRC work items, the call's totals) is a
 * fixed array in the library's device image, 1 380 096 bytes, which every GPU the
 * library runs on holds until the process ends, instead of buffers the context
 * grows with hipMalloc.  The call itself makes no hipMalloc, hipFree or other
 * allocating call, and records no profiling events.  A call of more than
 * NISSHI_STATIC_MAX_BATCHES batches or NISSHI_STATIC_MAX_BYTES bytes is refused
 * with NISSHI_CAPACITY (nisshi_hip_last_error says which limit).  All calls on one
 * GPU, from any context, share that GPU's array: calls that overlap in time on
 * one GPU race on it, so keep them on one stream or order them with events.
 * With totals == NULL the call only enqueues work (use this form under graph
 * capture); otherwise it waits for the stream. */
#define NISSHI_STATIC_MAX_BATCHES 16384u
#define NISSHI_STATIC_MAX_BYTES (512ull << 20)
int nisshi_hip_decode_dev_static(nisshi_hip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_len,
                                 const uint64_t *d_batch_off, uint32_t n_batches, int by_value,
                                 const nisshi_columns *d_cols, uint64_t rec_cap, uint64_t hdr_cap,
                                 nisshi_totals *totals, void *stream);

/* Encode batches from device columns (deflated::Batch::try_from(inflated::Batch)
 * over a frame): record_count, batch_length and crc are recomputed, the length
 * of every record is emitted as stored.  Spans of batch b are read from d_src at
 * d_src_data_off[b] + offset.  d_out_off receives n_batches + 1 offsets into
 * d_out.  Batches whose status is not NISSHI_OK are skipped. */
int nisshi_hip_encode_dev(nisshi_hip_ctx *ctx, const nisshi_columns *d_cols, uint32_t n_batches,
                          const uint8_t *d_src, const uint64_t *d_src_data_off, uint8_t *d_out,
                          uint64_t out_cap, uint64_t *d_out_off, nisshi_totals *totals,
                          void *stream);

/* Host-buffer forms: copy the blob to the device, decode, copy the columns
 * back.  The library owns the device buffers.  cols holds host pointers sized
 * rec_cap / hdr_cap.  A large blob is decoded in slices of whole batches (64 MiB,
 * or NISSHI_SLICE_BYTES in the environment when the context is created; 16 at
 * most) so that the copies of one slice overlap the kernels of another; the
 * records are those of one call, and a slice's header slots (the expected ones,
 * then those of its batches with more headers) follow those of the slice before
 * it.  The overlap needs page-locked buffers (below). */
int nisshi_hip_decode(nisshi_hip_ctx *ctx, const uint8_t *blob, uint64_t blob_len,
                      const uint64_t *batch_off, uint32_t n_batches, int by_value,
                      const nisshi_columns *cols, uint64_t rec_cap, uint64_t hdr_cap,
                      nisshi_totals *totals);
int nisshi_hip_encode(nisshi_hip_ctx *ctx, const nisshi_columns *cols, uint32_t n_batches,
                      uint64_t n_records, uint64_t n_headers, const uint8_t *src, uint64_t src_len,
                      const uint64_t *src_data_off, uint8_t *out, uint64_t out_cap,
                      uint64_t *out_off, nisshi_totals *totals);

/* ---- the produce seam of the SQL engines (nisshi-storage/src/pg.rs:797-943,
 * lite.rs:703-815).  The reference inflates the batch into Vec<Record> and walks
 * it twice to write one row a record (offset, attributes, timestamp, key,
 * value) and one row a header (offset, key, value).  Here the rows are columns
 * beside the decoded ones: key, value and header spans stay where the decode
 * put them (rkey, rval, hdr), the row's attributes are the batch's. */
typedef struct nisshi_rows {
    int64_t *offset;     /* per record slot: high + index in the batch (pg.rs:868-869) */
    int64_t *timestamp;  /* per record slot: base_timestamp + timestamp_delta, ms since the epoch (pg.rs:876) */
    uint32_t *batch;     /* per record slot: the owning batch; 0xFFFFFFFF for a slot that holds no record */
    int64_t *hdr_offset; /* per header slot: the offset of the header's record (pg.rs:913-914, 924) */
    int32_t *status;     /* per batch: the decode status, or NISSHI_TRY_FROM_INT (see check_time) */
} nisshi_rows;

/* d_cols: columns filled by nisshi_hip_decode_dev with by_value != 0 for n_batches
 * batches; n_slots: its totals.n_records (or rec_cap); d_high: per batch, the
 * partition's high watermark, 0 where the reference has None
 * (high.unwrap_or_default()).  check_time != 0 fails a batch that holds a negative
 * timestamp with NISSHI_TRY_FROM_INT, as to_system_time does for the PostgreSQL
 * engine (nisshi-sans-io/src/lib.rs:2189-2193); 0 stores the sum as it is, like
 * lite.rs:791.  The rows of a batch whose status is not NISSHI_OK are undefined.
 * Only enqueues work. */
int nisshi_hip_produce_rows_dev(nisshi_hip_ctx *ctx, const nisshi_columns *d_cols, uint32_t n_batches,
                                uint64_t n_slots, uint64_t hdr_cap, const int64_t *d_high,
                                int check_time, const nisshi_rows *d_rows, void *stream);

/* Host-buffer form: decode (by value) and rows in one call, all outputs copied
 * back.  cols and rows hold host pointers sized rec_cap / hdr_cap. */
int nisshi_hip_produce_rows(nisshi_hip_ctx *ctx, const uint8_t *blob, uint64_t blob_len,
                            const uint64_t *batch_off, uint32_t n_batches, const int64_t *high,
                            int check_time, const nisshi_columns *cols, const nisshi_rows *rows,
                            uint64_t rec_cap, uint64_t hdr_cap, nisshi_totals *totals);

/* ---- combine (nisshi-storage/src/batch.rs:661-709): the batches queued for one
 * topic partition become one batch.  Group g is batches d_group_off[g] ..
 * d_group_off[g + 1] of d_cols (n_groups + 1 ascending entries, the last at most
 * n_batches); its first batch is the sink.  In place, for the records of every
 * later member:
 *     offset_delta    += sink.last_offset_delta + 1        (running, batch.rs:689, 697)
 *     timestamp_delta += sink.base_timestamp - base_timestamp            (batch.rs:690-691)
 * length and all other fields stay, and span offsets are rebased to the sink's
 * record data, so the members must lie after the sink in the blob, within
 * 4 GiB (else NISSHI_CAPACITY).  The per-batch arrays of d_out (n_groups entries
 * each) receive the sink: the first member's header with last_offset_delta and
 * max_timestamp combined (batch.rs:697-698), record_count, rec_base, data_off;
 * status = the first member's status that is not NISSHI_OK (the whole combine
 * fails, as at batch.rs:670, 677; such a group's records are left untouched),
 * or NISSHI_NONE for an empty group.  d_out's per-record and per-header pointers
 * are not used: to deflate the sinks (batch.rs:706) pass d_out with d_cols'
 * record and header pointers to nisshi_hip_encode_dev, and d_out->data_off as
 * d_src_data_off.  Sums wrap, as in a release build of the reference.  d_cols
 * must come from a decode with by_value != 0.  Only enqueues work. */
int nisshi_hip_combine_dev(nisshi_hip_ctx *ctx, const nisshi_columns *d_cols, uint32_t n_batches,
                           uint64_t n_slots, uint64_t hdr_cap, const uint32_t *d_group_off,
                           uint32_t n_groups, const nisshi_columns *d_out, void *stream);

/* Host-buffer form: inflate, combine and deflate in one call.  blob holds
 * n_batches deflated batches; out receives one deflated batch a group
 * (out_off: n_groups + 1 offsets; a group that fails or is empty has no bytes),
 * group_status one status a group. */
int nisshi_hip_combine(nisshi_hip_ctx *ctx, const uint8_t *blob, uint64_t blob_len,
                       const uint64_t *batch_off, uint32_t n_batches, const uint32_t *group_off,
                       uint32_t n_groups, uint8_t *out, uint64_t out_cap, uint64_t *out_off,
                       int32_t *group_status, nisshi_totals *totals);

/* ---- the fetch seam of the SQL engines (nisshi-storage/src/pg.rs:1875-2052,
 * lite.rs:2765-2980).  The reference reads the rows of one topic partition in
 * offset order and builds deflated batches from them: a new batch where
 * (attributes, producer_id, producer_epoch) change (pg.rs:1933-1958), each from
 * inflated::Batch::builder() (nisshi-sans-io/src/record/inflated.rs:311-329) and
 * one Record::builder() a row (src/record.rs:298-314).  Here the rows are columns,
 * one entry a row, nullable SQL columns already defaulted as the reference does
 * (attributes 0, producer_id -1, producer_epoch -1). */
typedef struct nisshi_fetch_rows {
    const int64_t *offset;         /* column 0 */
    const int16_t *attributes;     /* column 1 */
    const int64_t *timestamp;      /* column 2, milliseconds since the epoch (to_timestamp) */
    const int64_t *producer_id;    /* column 6 */
    const int16_t *producer_epoch; /* column 7 */
    const uint8_t *completed;      /* column 8 (pg.rs:1928-1931); NULL: no such check (lite.rs) */
    const uint32_t *key_off;       /* column 3 as a span of the payload buffer */
    const int32_t *key_len;        /* negative: None */
    const uint32_t *val_off;       /* column 4 */
    const int32_t *val_len;
    const uint32_t *hdr_base;      /* n_rows + 1 entries: row i owns headers hdr_base[i] .. hdr_base[i + 1] (header_fetch.sql) */
    const nisshi_header_span *hdr; /* spans of the payload buffer */
} nisshi_fetch_rows;

/* Rows to the columns nisshi_hip_encode_dev takes; record r is row r.
 *   a row with completed == 0 that does not follow its predecessor's offset
 *     directly ends the fetch: it and the rows after it give no records, though it
 *     may still open a last, empty batch (pg.rs:1933-1971)
 *   offset_delta    = i32::try_from(offset - base_offset); a value that does not
 *     fit fails the fetch: every batch gets NISSHI_TRY_FROM_INT (pg.rs:1973)
 *   timestamp_delta = timestamp - base_timestamp (pg.rs:1975-1984), wrapping
 *   length          = the size of the record's body (record.rs:298-314, 363-372)
 *   a batch takes base_offset, attributes, base_timestamp, producer_id and
 *     producer_epoch from its first row, last_offset_delta from its last record,
 *     max_timestamp = now (the builder's default, never set by the fetch),
 *     partition_leader_epoch -1, magic 2, base_sequence 0; batch_length and crc
 *     are left 0 (the deflate recomputes them, deflated.rs:392-409)
 *   no rows give one default batch with base_timestamp = now and producer_epoch 0
 *     (pg.rs:2050-2052)
 * The deadline check of the reference (pg.rs:2044) is the caller's choice of
 * n_rows.  d_out: per-batch arrays of batch_cap entries, per-record arrays of
 * n_rows entries; its hdr pointer is not used: encode with d_out's pointers,
 * hdr = d_rows->hdr, d_src = the payload buffer and d_src_data_off = d_out->data_off
 * (all 0).  n_rows < 2^32.  With n_batches == NULL the call only enqueues work;
 * otherwise it waits for the stream, stores the number of batches and returns
 * NISSHI_CAPACITY when that exceeds batch_cap, NISSHI_TRY_FROM_INT when the
 * fetch failed. */
int nisshi_hip_fetch_batches_dev(nisshi_hip_ctx *ctx, const nisshi_fetch_rows *d_rows, uint64_t n_rows,
                                 int64_t now, const nisshi_columns *d_out, uint32_t batch_cap,
                                 uint32_t *n_batches, void *stream);

/* inflated::Builder::build (nisshi-sans-io/src/record/inflated.rs:448-481) for
 * batches built from builders and already deflated: fills d_cols->batch_length and
 * d_cols->crc with what the builder stores in the inflated::Batch it returns, so
 * that an inflated::Batch rebuilt from these columns equals the reference's.
 * d_cols: batches whose records come from Record::builder() (every length is the
 * size of the record's body: the output of nisshi_hip_fetch_batches_dev, or of
 * nisshi_hip_compact_dev for a rewritten batch); d_out, d_out_off: those batches
 * as nisshi_hip_encode_dev deflated them.
 *   crc          = the deflated batch's crc: Builder::crc (inflated.rs:429-446)
 *                  serialises exactly the bytes that crc covers (deflated.rs:318-326)
 *   batch_length = Builder::size_in_bytes (inflated.rs:331-346).  It sizes every
 *                  record's length prefix as an UNSIGNED varint (Sequence::size_in_bytes,
 *                  record/codec.rs:482-491 through UnsignedVarInt::size_inclusive,
 *                  primitive/varint.rs:536-543) while the prefix is written zigzag
 *                  (record.rs:345-351), so it is the wire's batch_length less one for
 *                  every record whose body size s has 64 <= s < 128, 8192 <= s < 16384,
 *                  2^20 <= s < 2^21 or 2^27 <= s < 2^28
 * Batches whose status is not NISSHI_OK are left as they are.  Only enqueues work. */
int nisshi_hip_builder_fields_dev(nisshi_hip_ctx *ctx, const nisshi_columns *d_cols, uint32_t n_batches,
                                  const uint8_t *d_out, const uint64_t *d_out_off, void *stream);

/* Host-buffer form: rows and payload in host memory to deflated batches in out
 * (out_off: batch_cap + 1 entries, *n_batches + 1 of them written). */
int nisshi_hip_fetch(nisshi_hip_ctx *ctx, const nisshi_fetch_rows *rows, uint64_t n_rows,
                     const uint8_t *payload, uint64_t payload_len, int64_t now, uint8_t *out,
                     uint64_t out_cap, uint64_t *out_off, uint32_t batch_cap, uint32_t *n_batches,
                     nisshi_totals *totals);

/* ---- log compaction of a topic partition (nisshi-storage/src/slate/storage.rs:283-312
 * with inflated::Batch::compact and keys, nisshi-sans-io/src/record/inflated.rs:213-275).
 * d_cols: the partition's batches decoded (by value) from d_blob, oldest first.
 * Working from the newest batch to the oldest, a record is dropped when its key
 * appears in a newer batch or later in its own batch; records without a key
 * stay; control batches (attributes bit 5) are neither compacted nor do their
 * keys count (storage.rs:291-293).  As in the reference, retain() goes by the
 * value of offset_delta (inflated.rs:262-264): a keyed record also stays when a
 * record of its batch that stays as the last of its key has the same offset_delta.
 * d_group (or NULL): a partition id per batch; keys are only compared within a
 * group, so several partitions can be compacted by one call.
 *
 * d_out receives the compacted partition: per-batch arrays of n_batches entries,
 * per-record arrays of n_slots entries (the records that stay, packed in order);
 * its hdr pointer is not used: a record's headers stay where they are in d_cols->hdr.
 * Batch headers are kept (the batch structure is preserved, inflated.rs:229-231);
 * record_count and rec_base are new; in a batch that lost records every record's
 * length becomes the size of its body, as into_builder().build() makes it
 * (inflated.rs:266-268, record.rs:298-314).  d_removed[b]: the count the
 * reference reports (Compaction::records).  d_action[b]: 0 nothing removed, the
 * stored batch stays (storage.rs:299-301); 1 rewritten; 2 no record left, the
 * batch is deleted (storage.rs:305-307).  d_out->status[b] is NISSHI_OK for a
 * rewritten batch and NISSHI_NONE for the other two, so that nisshi_hip_encode_dev
 * on d_out (with hdr = d_cols->hdr, d_src = d_blob, d_src_data_off = d_out->data_off)
 * deflates exactly the batches to write back (storage.rs:309-317); a batch that
 * did not decode keeps its status and takes no part (the reference fails the
 * whole compaction on it, storage.rs:289).  With totals == NULL the call only
 * enqueues work; otherwise it waits and totals->n_records is the number of
 * records that stay. */
int nisshi_hip_compact_dev(nisshi_hip_ctx *ctx, const uint8_t *d_blob, const nisshi_columns *d_cols,
                           uint32_t n_batches, uint64_t n_slots, const uint32_t *d_group,
                           const nisshi_columns *d_out, uint32_t *d_removed, int32_t *d_action,
                           nisshi_totals *totals, void *stream);

/* Host-buffer form: inflate, compact and deflate in one call.  out receives the
 * rewritten batches (out_off: n_batches + 1 offsets; a batch that is not
 * rewritten has no bytes); removed, action, status: one entry a batch. */
int nisshi_hip_compact(nisshi_hip_ctx *ctx, const uint8_t *blob, uint64_t blob_len,
                       const uint64_t *batch_off, uint32_t n_batches, const uint32_t *group,
                       uint8_t *out, uint64_t out_cap, uint64_t *out_off, uint32_t *removed,
                       int32_t *action, int32_t *status, nisshi_totals *totals);

/* ---- one host frame over several GPUs.  The reference handles a produce
 * request's batches partition by partition in one loop on one thread
 * (nisshi-storage/src/service/produce.rs:165-191); batches of different topic
 * partitions share nothing.  A pool holds one context, one worker thread and
 * page-locked staging buffers per device; a frame's batch b goes to device
 * partition[b] mod G (G = the number of devices), the devices decode side by side
 * (nisshi_hip_decode each: copy in, kernels, copy out), and the columns come back
 * in input order, as one device decoding the whole frame would leave them: entry
 * b of every per-batch array is batch b, data_off[b] is an offset in the caller's
 * blob, rec_base and hdr_base index the caller's record and header arrays (a
 * device's records are contiguous, devices in ascending order).  No data moves
 * between devices.  devices: n_devices HIP device ids (NULL: 0 .. n_devices - 1;
 * an id may repeat, which gives that GPU several contexts).  One call at a time. */
typedef struct nisshi_hip_pool nisshi_hip_pool;
int nisshi_hip_pool_create(const int *devices, uint32_t n_devices, nisshi_hip_pool **pool);
void nisshi_hip_pool_destroy(nisshi_hip_pool *pool);
uint32_t nisshi_hip_pool_device_count(const nisshi_hip_pool *pool);
uint32_t nisshi_hip_pool_device_of(const nisshi_hip_pool *pool, uint32_t partition);
const char *nisshi_hip_pool_last_error(const nisshi_hip_pool *pool);
/* Where the last nisshi_hip_pool_decode spent its time, in milliseconds of host
 * clock: ms[0] gathering a device's batches into its staging blob, ms[1] in that
 * device's decode (copy in, kernels, copy out), ms[2] putting its columns back
 * into the caller's -- each the slowest device's, the devices work side by side --
 * and ms[3] the whole call.  (Measurement only: no counterpart in the reference.) */
void nisshi_hip_pool_last_times(const nisshi_hip_pool *pool, double *ms);
/* partition: one id a batch.  totals: those of the frame; device_totals (or NULL):
 * n_devices entries, each device's own.  Capacities are those of the caller's
 * columns (header slots with the slack nisshi_hip_decode_dev describes, summed
 * over the devices). */
int nisshi_hip_pool_decode(nisshi_hip_pool *pool, const uint8_t *blob, uint64_t blob_len, const uint64_t *batch_off,
                           uint32_t n_batches, const uint32_t *partition, int by_value, const nisshi_columns *cols,
                           uint64_t rec_cap, uint64_t hdr_cap, nisshi_totals *totals, nisshi_totals *device_totals);
/* A pool over decoders other than this library's contexts: device d decodes by
 * decode(decoder_ctx[d], ...), the arguments those of nisshi_hip_decode; staging
 * buffers are ordinary host memory.  This is how the scheduling (the partition
 * map, the order of the shares, the columns put back in input order) is tested on
 * machines without GPUs, with the tests' CPU checker as the decoder. */
typedef int (*nisshi_pool_decode_fn)(void *decoder_ctx, const uint8_t *blob, uint64_t blob_len,
                                     const uint64_t *batch_off, uint32_t n_batches, int by_value,
                                     const nisshi_columns *cols, uint64_t rec_cap, uint64_t hdr_cap,
                                     nisshi_totals *totals);
int nisshi_hip_pool_create_custom(uint32_t n_devices, nisshi_pool_decode_fn decode, void *const *decoder_ctx,
                                  nisshi_hip_pool **pool);

/* ---- compressed batches: gzip, Snappy, LZ4 and zstd (Compression::inflator, nisshi-sans-io/src/lib.rs:1716-1765,
 * as called by Vec<Record>::try_from(deflated::Batch), nisshi-sans-io/src/record/deflated.rs:444-457).
 * The reference inflates a compressed batch's record data and deserializes the records from the
 * result.  This call rewrites a frame: every batch of d_blob in order; a batch whose header passes
 * the checks of deflated::Batch::try_from(Bytes) (deflated.rs:108-213), whose codec is gzip, Snappy,
 * LZ4 or zstd (attributes & 7 in 1..4) and whose record data inflates becomes an uncompressed batch --
 * its 61 header bytes with batch_length = 49 + inflated length and the codec bits of attributes
 * cleared (inflated::Batch::try_from clears them, inflated.rs:123-140; the crc field stays the
 * compressed batch's, as the reference's inflated batch carries it: it does not cover the
 * rewritten bytes), then the inflated record data -- with d_status[b] = NISSHI_OK; one that does
 * not inflate is passed through as it is, NISSHI_INFLATE: take the reference's own path for it;
 * every other batch (uncompressed, a header that does not parse) is passed through,
 * NISSHI_NONE.  Decoding the output with nisshi_hip_decode_dev gives the records the reference
 * gives for the inflated batches.
 * Snappy: as in the reference the record data is one raw block (the `snap` crate's
 * raw::Decoder::decompress_vec), behind 20 bytes that are skipped when it starts with the xerial
 * magic "\x82SNAPPY\0": a stream of several xerial chunks does not inflate (the reference's own
 * test of one is #[ignore]d, nisshi-sans-io/tests/snappy.rs:1163-1166).
 * gzip: one member as flate2's GzDecoder reads it (RFC 1952 header with its optional extra field,
 * name and comment; a header checksum is refused; RFC 1951 blocks: stored, fixed and dynamic codes),
 * whose length field must be the number of bytes inflated and must end the record data; the
 * trailer's CRC-32 is not compared (the reference does not read that far).
 * LZ4: one frame as liblz4's LZ4F_decompress reads it (the `lz4` crate's Decoder): linked or
 * independent blocks, stored blocks, block checksums, content size and content checksum; a frame
 * with a dictionary id is refused.
 * zstd: the frames of the record data as libzstd's stream decoder reads them (the `zstd` crate's
 * stream::read::Decoder; RFC 8878): stored, repeated-byte and compressed blocks, Huffman-coded
 * literals in one or four streams, predefined, described, repeated and one-symbol code tables,
 * repeat offsets, content size and content checksum (XXH64), skippable frames; a frame with a
 * dictionary id or a window beyond the stream decoder's default limit (128 MiB) is refused.
 * The reference reads a gzip member, an LZ4 frame or a zstd frame only as far as
 * its records need it; here the whole of it must be sound, so one that is damaged behind its last record is
 * NISSHI_INFLATE here and Ok there: the reference's own path, which the status asks for, gives
 * its result.
 *
 * d_out_off receives n_batches + 1 offsets into d_out.  With d_out == NULL the call only says how
 * much room the output needs: totals->out_bytes is a capacity that holds it whatever the blocks
 * turn out to be (each batch counted at the larger of its own size and what its Snappy block or
 * gzip member announces, an LZ4 or zstd batch with what its frames' blocks can hold at most; a block that announces more than 22 bytes
 * a byte of its own cannot inflate and is not counted so); d_status and d_out_off are not final then.  Otherwise a d_out of out_cap bytes (16-byte
 * aligned) receives the frame; NISSHI_CAPACITY when out_cap is below that capacity.  The output
 * is laid out by what the blocks announce and written by one walk over them (LZ4 and zstd frames,
 * which need not announce anything, are walked into a scratch area of the context first and copied from there);
 * only when a block then does not inflate is the output laid out and written a second time, with
 * that batch passed through.
 * totals (not NULL: the call waits) has out_bytes, the bytes of the frame, and n_failed, the
 * number of batches that did not inflate. */
int nisshi_hip_inflate_dev(nisshi_hip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_len,
                           const uint64_t *d_batch_off, uint32_t n_batches, uint8_t *d_out,
                           uint64_t out_cap, uint64_t *d_out_off, int32_t *d_status,
                           nisshi_totals *totals, void *stream);

/* The same for host buffers: copies in, inflates, copies out.  out_cap below the capacity that the
 * device form asks for: NISSHI_CAPACITY, with totals->out_bytes that capacity. */
int nisshi_hip_inflate(nisshi_hip_ctx *ctx, const uint8_t *blob, uint64_t blob_len,
                       const uint64_t *batch_off, uint32_t n_batches, uint8_t *out, uint64_t out_cap,
                       uint64_t *out_off, int32_t *status, nisshi_totals *totals);

/* Batches compressed: into_record_data (nisshi-sans-io/src/record/deflated.rs:341-386), which
 * deflated::Batch::try_from(inflated::Batch) (deflated.rs:392-409) calls with the batch's codec, for
 * its three arms: Lz4 (deflated.rs:362-371: the lz4 crate's Encoder, liblz4's frame format), Gzip
 * (deflated.rs:341-357: flate2's GzEncoder) and Zstd (deflated.rs:373-386: the zstd crate's
 * Encoder).  A frame of uncompressed batches goes in -- what nisshi_hip_encode_dev writes -- and
 * the same frame comes out with every batch whose d_codec[b] is 3, 1 or 4 rewritten: its record
 * data one LZ4 frame (independent blocks of 8 KiB, no checksums, no content size; a block that
 * does not shrink is stored), one gzip member (the header flate2 writes; DEFLATE blocks of 8 KiB
 * each in a code of its own, in the fixed code of RFC 1951 3.2.6, or stored, whichever is
 * smallest; CRC-32 and length in the trailer) or one zstd frame (no content size, no checksum, window 64 KiB; blocks
 * of 8 KiB with raw literals and sequences in the predefined tables of RFC 8878, or raw where
 * that does not shrink them), its batch_length that of the compressed data, the codec in its
 * attributes, its crc that of the new bytes.
 * d_codec[b] 0: the batch is copied.  2 (Snappy), which the reference refuses to write
 * (Error::UnexpectedType): the batch is copied as it came with NISSHI_COMPRESSED, for the
 * caller's own path.  Above 4: copied, NISSHI_UNKNOWN_COMPRESSION.
 * A batch to be compressed whose header does not hold (shorter than 61 bytes, a batch_length
 * that is not its length, codec bits already set): copied, NISSHI_MALFORMED.
 * Which bytes a compressor writes is its own choice (the reference holds no vector of compressed
 * output); what holds is that the data inflates to the record data it was made from -- by the
 * reference's decoders (lib.rs:1716-1765), by nisshi_hip_inflate_dev, by the CPU checker -- and that
 * the batch around it decodes like any compressed batch.
 * With d_out == NULL the call only says how much room it needs: totals->out_bytes, every batch at
 * its worst case (a few bytes a block more than uncompressed, and a KiB a batch of working room:
 * the batches are compressed into a scratch area of that size).  Otherwise NISSHI_CAPACITY when
 * out_cap is below that; d_out_off receives n_batches + 1 offsets into d_out, and
 * totals->out_bytes the bytes of the frame, n_failed the batches whose status is not NISSHI_OK.
 * totals must not be NULL: the call waits for the device twice. */
int nisshi_hip_deflate_dev(nisshi_hip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_len,
                           const uint64_t *d_batch_off, uint32_t n_batches, const uint8_t *d_codec,
                           uint8_t *d_out, uint64_t out_cap, uint64_t *d_out_off, int32_t *d_status,
                           nisshi_totals *totals, void *stream);

/* The same for host buffers: copies in, compresses, copies out; out_cap is compared with the
 * frame that came out (NISSHI_CAPACITY, with totals->out_bytes its size). */
int nisshi_hip_deflate(nisshi_hip_ctx *ctx, const uint8_t *blob, uint64_t blob_len,
                       const uint64_t *batch_off, uint32_t n_batches, const uint8_t *codec, uint8_t *out,
                       uint64_t out_cap, uint64_t *out_off, int32_t *status, nisshi_totals *totals);

/* Page-locked host memory for the host-buffer entry points: blobs and columns
 * in buffers from here move over PCIe by DMA at link speed; any other host
 * memory works too, staged by the HIP runtime and several times slower.  (The
 * reference has no counterpart: its batches never leave host memory.) */
int nisshi_hip_host_alloc(nisshi_hip_ctx *ctx, uint64_t bytes, void **ptr);
void nisshi_hip_host_free(nisshi_hip_ctx *ctx, void *ptr);

/* CRC-32C of a device buffer (crc-fast "Crc32Iscsi", used at deflated.rs:158-161). */
int nisshi_hip_crc32c_dev(nisshi_hip_ctx *ctx, const uint8_t *d_data, uint64_t len, uint32_t *crc,
                          void *stream);

/* Per-kernel timing with HIP events on the stream the kernels run on.  While
 * enabled, every decode/encode call records an event after each kernel; _get
 * waits for the recorded events and returns the summed duration and launch
 * count of one kernel since the last _enable.  _name returns NULL past the last
 * kernel id.  on = 1: every kernel; on = 2 + k: only kernel k of a decode call (an
 * event before it and one after it: the other kernels run as they do unobserved,
 * an event between two kernels keeps them a few microseconds apart), and with
 * (p - 1) << 8 added only in one decode call of p, the others entirely
 * unobserved; 0: off.  Has
 * no counterpart in the reference (it instruments with the tracing crate, e.g.
 * src/record.rs:215, 234, 255). */
int nisshi_hip_profile_enable(nisshi_hip_ctx *ctx, int on);
int nisshi_hip_profile_get(nisshi_hip_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches);
const char *nisshi_hip_profile_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif
