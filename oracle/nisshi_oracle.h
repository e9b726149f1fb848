/*
 * CPU restatement of the Kafka RecordBatch v2 codec of nisshi-sans-io (tansu).
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * codec and the CPU baseline leg of bench.py.  Nothing in the shipped package
 * may include, link or load it.
 *
 * The reference is Rust and cannot be built in this environment (no rustc or
 * cargo), so this file restates its algorithm function by function; every
 * function cites the reference lines it follows (paths relative to the
 * reference tree).  It is pinned by the reference's own golden vectors and
 * known-answer tests, transcribed as fixtures under tests/golden/.
 *
 * Error model.  The reference returns Result<_, Error> and panics on some
 * malformed inputs; both are folded into one status code:
 *   TRY_GET        Error::TryGet        (bytes::TryGetError, buffer under-run)
 *   OVERFLOW       Error::Overflow
 *   TRY_FROM_INT   Error::TryFromInt    (negative batch_length -> usize)
 *   UNKNOWN_COMPRESSION  Error::UnknownCompressionType (attributes & 7 in 5..7)
 *   PANIC          the reference panics (Bytes::split_to past the end,
 *                  Buf::get_u8 on an empty buffer, Vec::with_capacity overflow)
 *   COMPRESSED     attributes & 7 in 1..4: valid, but needs an inflator that
 *                  is outside this path
 */
#ifndef NISSHI_ORACLE_H
#define NISSHI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    NO_OK = 0,
    NO_TRY_GET = 1,
    NO_OVERFLOW = 2,
    NO_TRY_FROM_INT = 3,
    NO_UNKNOWN_COMPRESSION = 4,
    NO_PANIC = 5,
    NO_COMPRESSED = 6,
    NO_CAPACITY = 7,
    NO_INFLATE = 11 /* the compressed record data does not inflate (Error::Snap / Error::Io from the inflator) */
};

/* ---- CRC-32C ("Crc32Iscsi" of crc-fast 1.8.0; call sites deflated.rs:158-161,
 * 322-325, inflated.rs:442-445).  Reflected polynomial 0x82F63B78, init and
 * final xor 0xFFFFFFFF. */
uint32_t no_crc32c(const uint8_t *data, size_t len);

/* ---- varints (primitive/varint.rs) ---- */
int no_varint_decode(const uint8_t *buf, size_t len, size_t *pos, int32_t *out);      /* :49-78 */
int no_varlong_decode(const uint8_t *buf, size_t len, size_t *pos, int64_t *out);     /* :266-295 */
size_t no_varint_size(int32_t v);                                                      /* :186-196 */
size_t no_varlong_size(int64_t v);                                                     /* :379-389 */
size_t no_uvarint_size(uint32_t v);                                                    /* :546-556 */
size_t no_varint_encode(int32_t v, uint8_t *out);                                      /* :80-98 */
size_t no_varlong_encode(int64_t v, uint8_t *out);                                     /* :248-264 */
size_t no_uvarint_encode(uint32_t v, uint8_t *out);                                    /* :435-450 */
int no_uvarint_decode(const uint8_t *buf, size_t len, size_t *pos, uint32_t *out);     /* :476-534 */

/* ---- batch attributes (lib.rs:1643-1850) ---- */
int no_compression(int16_t attributes, int *codec);          /* Compression::try_from :1685-1698 */
int no_attributes_clear_compression(int16_t attributes, int16_t *out); /* BatchAttribute round trip :1814-1850 */

/* ---- record types: the reference's in-memory shape (record.rs:172-198,
 * header.rs:21-29).  Spans borrow the input buffer like Bytes::split_to. */
typedef struct {
    const uint8_t *ptr; /* first byte of the span (where it would start, when None) */
    int64_t len;        /* -1 when the Option<Bytes> is None */
} no_octets;

typedef struct {
    no_octets key;
    no_octets value;
} no_header;

typedef struct {
    int32_t length;
    uint8_t attributes;
    int64_t timestamp_delta;
    int32_t offset_delta;
    no_octets key;
    no_octets value;
    no_header *headers; /* heap vector, one per record, like Vec<Header> */
    size_t n_headers;
} no_record;

void no_record_free(no_record *r);
/* Record::decode, record.rs:254-277 */
int no_record_decode(const uint8_t *buf, size_t len, size_t *pos, no_record *out);
/* Record::size_in_bytes, record.rs:200-212 (uses the stored length verbatim) */
size_t no_record_size(const no_record *r);
/* Record::encode, record.rs:233-252 */
size_t no_record_encode(const no_record *r, uint8_t *out);
/* Octets::size_in_bytes, record/codec.rs:34-45; len < 0 is None */
size_t no_octets_size(int64_t len);
/* record::Builder::size_in_bytes, record.rs:363-372: body size, excluding the
 * length prefix; this is the value Record::try_from(Builder) stores in length */
size_t no_record_body_size(const no_record *r);

/* ---- deflated::Batch (record/deflated.rs:67-82) ---- */
typedef struct {
    int64_t base_offset;
    int32_t batch_length;
    int32_t partition_leader_epoch;
    int8_t magic;
    uint32_t crc;
    int16_t attributes;
    int32_t last_offset_delta;
    int64_t base_timestamp;
    int64_t max_timestamp;
    int64_t producer_id;
    int16_t producer_epoch;
    int32_t base_sequence;
    uint32_t record_count;
    const uint8_t *record_data;
    size_t record_data_len;
    uint32_t crc_computed; /* what the reference computes and only logs on mismatch */
} no_deflated_batch;

/* deflated::Batch::try_from(Bytes), deflated.rs:108-213 */
int no_deflated_parse(const uint8_t *buf, size_t len, no_deflated_batch *out);
/* From<deflated::Batch> for Bytes, deflated.rs:84-106; returns bytes written (61 + data) */
size_t no_deflated_emit(const no_deflated_batch *b, uint8_t *out);
/* Vec<Record>::try_from(deflated::Batch), deflated.rs:421-459 (Compression::None fast path).
 * On success *records is a heap vector of record_count entries. */
int no_inflate_records(const no_deflated_batch *b, no_record **records, size_t *n_records);
void no_records_free(no_record *records, size_t n);
/* (&[Record]).encode, record.rs:214-231; returns total bytes; out may be NULL to size */
size_t no_records_encode(const no_record *records, size_t n, uint8_t *out);
/* deflated::Batch::try_from(inflated::Batch), deflated.rs:392-409 with
 * into_record_data :329-390 (None) and CrcData::into_batch :295-326.  The
 * header fields are taken from hdr; batch_length, crc and record_count are
 * recomputed.  out receives 61 + record_data bytes. */
int no_deflate(const no_deflated_batch *hdr, const no_record *records, size_t n, uint8_t *out,
               size_t out_cap, size_t *out_len);
/* inflated::Builder::build batch_length and crc (inflated.rs:331-346, 429-481),
 * including the unsigned-varint sizing of the per-record length prefix
 * (codec.rs:486-490) */
int no_builder_batch_length_crc(const no_deflated_batch *hdr, const no_record *records, size_t n,
                                int32_t *batch_length, uint32_t *crc);

/* ---- records-blob framing (de.rs:1056-1093): split a records blob into batches.
 * offsets receives n+1 entries. */
int no_frame_split(const uint8_t *blob, size_t len, uint64_t *offsets, size_t cap, size_t *n);
/* records field length prefix (de.rs:701-720, ser.rs:521-574) */
int no_records_prefix_decode(const uint8_t *buf, size_t len, int flexible, size_t *pos,
                             uint32_t *records_len, int *is_null);
size_t no_records_prefix_encode(uint32_t records_len, int flexible, uint8_t *out);

/* ---- column (SoA) form used to compare with the HIP codec ---- */
typedef struct {
    /* per batch, n_batches entries */
    int64_t *base_offset;
    int32_t *batch_length;
    int32_t *partition_leader_epoch;
    int8_t *magic;
    uint32_t *crc;
    int16_t *attributes;
    int32_t *last_offset_delta;
    int64_t *base_timestamp;
    int64_t *max_timestamp;
    int64_t *producer_id;
    int16_t *producer_epoch;
    int32_t *base_sequence;
    uint32_t *record_count;
    uint32_t *crc_computed;
    uint64_t *data_off; /* absolute offset of record_data in the blob */
    uint32_t *data_len;
    uint32_t *rec_base; /* index of the batch's first record in the record columns */
    uint32_t *hdr_base; /* index of the batch's first header in the header columns */
    uint32_t *hdr_count;
    int32_t *status;
    /* per record */
    int32_t *length;
    uint8_t *rattributes;
    int64_t *timestamp_delta;
    int32_t *offset_delta;
    uint32_t *key_off; /* offsets are relative to the batch's record_data */
    int32_t *key_len;  /* -1 = null */
    uint32_t *val_off;
    int32_t *val_len;
    uint32_t *rhdr_base; /* absolute index into the header columns */
    int32_t *rhdr_count;
    /* per header */
    uint32_t *hkey_off;
    int32_t *hkey_len;
    uint32_t *hval_off;
    int32_t *hval_len;
} no_columns;

/* Decode every batch of a blob into columns.  batch_off has n_batches + 1
 * entries.  A batch that fails gets its status and contributes no records.
 * by_value != 0 normalises attributes as inflated::Batch::try_from(deflated::Batch)
 * does (inflated.rs:123-140); 0 keeps them verbatim (inflated.rs:161-177).
 * rec_cap / hdr_cap bound the column arrays; returns NO_CAPACITY when exceeded. */
int no_decode_columns(const uint8_t *blob, const uint64_t *batch_off, size_t n_batches,
                      int by_value, no_columns *cols, size_t rec_cap, size_t hdr_cap,
                      size_t *n_records, size_t *n_headers);
/* the same outcome without the intermediate Vec<Record> (see the definition) */
int no_decode_columns_lean(const uint8_t *blob, const uint64_t *batch_off, size_t n_batches,
                      int by_value, no_columns *cols, size_t rec_cap, size_t hdr_cap,
                      size_t *n_records, size_t *n_headers);

/* Re-encode batches from columns (deflated::Batch::try_from(inflated::Batch) over a
 * whole frame, deflated.rs:52-63).  Payload spans are read from src at
 * src_data_off[b] + offset.  Batches whose status is not OK are skipped (zero
 * bytes).  out_off receives n_batches + 1 entries. */
int no_encode_columns(const no_columns *cols, size_t n_batches, const uint8_t *src,
                      const uint64_t *src_data_off, uint8_t *out, size_t out_cap,
                      uint64_t *out_off);

/* inflated::Builder::build over columns: per batch, the batch_length and crc an
 * inflated::Batch built from these records would carry (inflated.rs:448-481).
 * Each record's length is recomputed from its body, as Record::try_from(Builder)
 * does (record.rs:298-314). */
int no_builder_columns(const no_columns *cols, size_t n_batches, const uint8_t *src,
                       const uint64_t *src_data_off, int32_t *batch_length, uint32_t *crc);

/* ---- callers of the codec in nisshi-storage, restated at the record level so
 * that the column operators of the HIP library can be checked against them ---- */

/* The produce seam of the SQL engines: inflated::Batch::try_from(deflated), then
 * one row per record and one per header (nisshi-storage/src/pg.rs:797-943).
 *   row offset     high.unwrap_or_default() + index                pg.rs:867-869, 912-914
 *   row timestamp  base_timestamp + timestamp_delta (milliseconds) pg.rs:876
 *   row attributes the batch's, after the by-value normalisation   pg.rs:870
 * check_time != 0 follows pg.rs:876, where to_system_time (nisshi-sans-io/src/lib.rs:
 * 2189-2193) turns a negative timestamp into Error::TryFromInt; 0 follows
 * nisshi-storage/src/lite.rs:762-815, which stores the sum as it is.
 * Spans are positions in `batch`.  hdr_* rows are in record order, each with
 * the offset of its record. */
int no_produce_rows(const uint8_t *batch, size_t len, int64_t high, int check_time,
                    size_t row_cap, size_t hdr_cap, int16_t *attributes,
                    int64_t *row_offset, int64_t *row_timestamp,
                    uint64_t *row_key_pos, int64_t *row_key_len,
                    uint64_t *row_val_pos, int64_t *row_val_len,
                    int64_t *hdr_offset, uint64_t *hdr_key_pos, int64_t *hdr_key_len,
                    uint64_t *hdr_val_pos, int64_t *hdr_val_len,
                    size_t *n_rows, size_t *n_hdrs);

/* combine, nisshi-storage/src/batch.rs:661-709: the batches queued for one
 * topic partition become one.  The first batch is the sink; the records of each
 * later batch are appended with
 *   offset_delta    += sink.last_offset_delta + 1
 *   timestamp_delta += sink.base_timestamp - batch.base_timestamp
 * (length and everything else as they were), then
 *   sink.last_offset_delta += batch.last_offset_delta + 1
 *   sink.max_timestamp      = max(sink.max_timestamp, batch.max_timestamp)
 * and the sink is deflated again (deflated.rs:392-409).  Integer sums wrap, as
 * they do in a release build of the reference.  n == 0 gives *is_none = 1
 * (Ok(None), batch.rs:666-668).  The first batch that fails to inflate gives
 * its status. */
int no_combine(const uint8_t *const *batches, const size_t *lens, size_t n, uint8_t *out,
               size_t out_cap, size_t *out_len, int *is_none);

/* The fetch seam of the SQL engines: the rows of one topic partition, in offset
 * order, become deflated batches (nisshi-storage/src/pg.rs:1875-2052; lite.rs:2760-2975
 * is the same without the `completed` column: pass NULL).  One array entry a row,
 * nullable columns already defaulted as the reference does (attributes 0,
 * producer_id -1, producer_epoch -1); timestamps in milliseconds (to_timestamp).
 * A new batch starts where (attributes, producer_id, producer_epoch) change
 * (pg.rs:1933-1958); a row with completed == 0 that does not follow its
 * predecessor's offset directly ends the fetch (pg.rs:1965-1971); each record is
 *   offset_delta    = i32::try_from(offset - base_offset)   (else TRY_FROM_INT for the call)
 *   timestamp_delta = timestamp - base_timestamp
 *   length          = the body size (record.rs:298-314), attributes 0
 * and each batch is inflated::Batch::builder() (inflated.rs:311-329: partition_leader_epoch
 * -1, magic 2, base_sequence 0, max_timestamp = now) with base_offset, attributes,
 * base_timestamp, producer_id, producer_epoch of its first row and last_offset_delta
 * of its last record, deflated (deflated.rs:392-409).  No rows give one empty
 * default batch (pg.rs:2050-2052).  Spans are (offset, length) into payload, a
 * negative length is None; row i's headers are hdr_spans[4 * hdr_base[i] .. 4 * hdr_base[i + 1])
 * as (key_off, key_len, val_off, val_len).  The deadline check of the reference
 * (pg.rs:2044) is the caller's choice of n. */
int no_fetch(size_t n, const int64_t *offset, const int16_t *attributes, const int64_t *timestamp,
             const int64_t *producer_id, const int16_t *producer_epoch, const uint8_t *completed,
             const uint32_t *key_off, const int32_t *key_len, const uint32_t *val_off,
             const int32_t *val_len, const uint32_t *hdr_base, const uint32_t *hdr_spans,
             const uint8_t *payload, int64_t now, uint8_t *out, size_t out_cap, uint64_t *out_off,
             size_t batch_cap, size_t *n_batches);

/* Log compaction of one topic partition (nisshi-storage/src/slate/storage.rs:283-312
 * with inflated::Batch::compact and keys, nisshi-sans-io/src/record/inflated.rs:213-275).
 * batches: the partition's deflated batches, oldest first.  Newest to oldest, a
 * record is dropped when its key appears in a newer batch or later in its own
 * batch; records without a key stay; control batches (attributes bit 5) are
 * neither compacted nor do their keys count.  As in the reference, retain() goes
 * by the VALUE of offset_delta: a keyed record also stays when another record of
 * its batch that stays as the last of its key has the same offset_delta.
 * removed[b] is the count the reference reports (Compaction::records); action[b]:
 *   UNCHANGED  nothing removed (or a control batch): the stored batch stays
 *   DELETED    no record left (storage.rs:305-307)
 *   REWRITTEN  the batch rebuilt by into_builder().build() (every length becomes the
 *              body size, record.rs:298-314) and deflated (deflated.rs:392-409): its
 *              bytes are out[out_off[b] .. out_off[b + 1]) */
enum { NO_COMPACT_UNCHANGED = 0, NO_COMPACT_REWRITTEN = 1, NO_COMPACT_DELETED = 2 };
int no_compact_partition(const uint8_t *const *batches, const size_t *lens, size_t n, uint8_t *out,
                         size_t out_cap, uint64_t *out_off, int64_t *removed, int32_t *action);

/* ---- compressed record data: Snappy ----
 * Compression::inflator, Snappy arm (nisshi-sans-io/src/lib.rs:1717-1751): the whole record data
 * is one raw Snappy block, after 20 bytes that are skipped when it starts with the xerial magic
 * "\x82SNAPPY\0" (magic 8, version 4, compatible version 4 and the first chunk's length 4: a stream
 * of several xerial chunks is not handled by the reference, and so not here).
 * The raw block is decoded by snap::raw::Decoder::decompress_vec of the `snap` crate (1.1.1 in the
 * reference's Cargo.lock; the crate is not vendored, so this follows the published format, Snappy's
 * format_description.txt, with the crate's outcomes: an empty input, a malformed or over-long length
 * varint, a length beyond u32, an element that reads past the input or writes past the announced
 * length, a copy with offset 0 or beyond what is written, and a final length that differs from the
 * announced one all fail).  Returns NO_OK or NO_INFLATE. */
int no_snappy_raw_len(const uint8_t *in, size_t n, size_t *header_len, uint64_t *len);
int no_snappy_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);
int no_snappy_inflate(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);

/* ---- compressed record data: LZ4 ----
 * Compression::inflator, Lz4 arm (nisshi-sans-io/src/lib.rs:1752-1755): lz4::Decoder over the record
 * data, i.e. liblz4's LZ4F_decompress (`lz4` crate 1.28.1 in Cargo.lock, which builds liblz4 from
 * source; neither is vendored in the reference, so this follows the published LZ4 frame and block
 * formats, lz4_Frame_format.md and lz4_Block_format.md, with liblz4's checks).  One frame is read:
 * magic 0x184D2204, FLG (version 01, reserved bit clear), BD (block size 4..7, reserved bits clear),
 * optional content size and dictionary id, the header checksum (second byte of XXH32 of the
 * descriptor); blocks of up to the maximal block size, compressed or stored, until the end mark;
 * block checksums and the content checksum where flagged; the content size where given.  What lies
 * behind the frame is not looked at (the reference never reads it).
 * The reference reads the frame lazily, as far as its records need it, so damage behind the last
 * record's block goes unnoticed there; here the whole frame must be sound (NO_INFLATE otherwise: the
 * caller takes the reference's own path for that batch).  A frame with a dictionary id is refused. */
uint32_t no_xxh32(const uint8_t *data, size_t len, uint32_t seed);
int no_lz4_frame(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);

/* ---- compressed record data: gzip ----
 * Compression::inflator, Gzip arm (nisshi-sans-io/src/lib.rs:1716): flate2's GzDecoder over the record data
 * (`flate2` 1.1 in Cargo.lock with its Rust inflate, not vendored: restated from RFC 1952, the gzip member,
 * and RFC 1951, DEFLATE).  One member is read: the ten-byte header (magic 1f 8b, method 8; the optional
 * extra field, name and comment are skipped; a header checksum or a reserved flag is refused), DEFLATE blocks
 * (stored, fixed and dynamic Huffman codes; a set of code lengths must be complete, but for a distance code
 * of one single code; distances reach no further back than what is written), and the trailer, whose length
 * field must be the number of bytes inflated and must end the record data.  The trailer's CRC-32 is not
 * compared: the reference reads the stream as far as its records need it and does not get to the trailer.
 * As for LZ4: the whole member must be sound here, NO_INFLATE otherwise. */
int no_gzip_member(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);

/* Compression::inflator's Zstd arm (nisshi-sans-io/src/lib.rs:1756-1765): the zstd crate's stream Decoder
 * (libzstd) over the record data, restated from the published format (RFC 8878): every frame of the data
 * (skippable ones skipped) into `out`; a frame with a dictionary id, a window beyond the stream decoder's
 * default limit of 128 MiB, a content checksum or content size that the bytes do not meet, or any damage
 * is NO_INFLATE.  As for LZ4: the whole of the record data must be sound here.  no_zstd_room: a `cap` that
 * suffices (0: the frame and block headers do not chain up to the end of the data).  no_xxh64: XXH64, of
 * which a frame's content checksum is the low half. */
size_t no_zstd_room(const uint8_t *in, size_t n);
int no_zstd_frames(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);
uint64_t no_xxh64(const uint8_t *p, size_t len, uint64_t seed);

/* A frame with its gzip, Snappy and LZ4 batches inflated: every batch of blob in order; one that parses
 * (no_deflated_parse) with codec 1, 2 or 3 (no_compression) and inflates becomes
 *   its 61 header bytes with batch_length = 49 + inflated length and the codec bits of
 *   attributes cleared (inflated::Batch::try_from clears them, inflated.rs:123-140; the crc field
 *   stays the one of the compressed batch, as the inflated batch carries it), then the inflated bytes,
 * status NO_OK; one that does not inflate is passed through, status NO_INFLATE; every other batch
 * (uncompressed, zstd, a header that does not parse) is passed through with status NO_NONE_DONE.  Decoding the output with no_decode_columns gives
 * the records Vec<Record>::try_from(deflated::Batch) gives for the inflated batches
 * (deflated.rs:444-457). */
enum { NO_NONE_DONE = 10 };
int no_inflate_frame(const uint8_t *blob, const uint64_t *batch_off, size_t n_batches, uint8_t *out,
                     size_t out_cap, uint64_t *out_off, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif
