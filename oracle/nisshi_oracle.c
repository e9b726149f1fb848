/*
 * CPU restatement of the nisshi-sans-io RecordBatch v2 codec.
 * TEST INFRASTRUCTURE ONLY: see nisshi_oracle.h.  Citations are paths inside
 * the reference tree.
 */
#include "nisshi_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ CRC-32C */

static uint32_t crc_table[8][256];
static int crc_table_ready;

static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
        crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++)
            crc_table[t][i] = (crc_table[t - 1][i] >> 8) ^ crc_table[0][crc_table[t - 1][i] & 0xff];
    crc_table_ready = 1;
}

uint32_t no_crc32c(const uint8_t *data, size_t len) {
    if (!crc_table_ready) crc_init();
    uint32_t c = 0xFFFFFFFFu;
    while (len >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, data, 4);
        memcpy(&hi, data + 4, 4);
        lo ^= c;
        c = crc_table[7][lo & 0xff] ^ crc_table[6][(lo >> 8) & 0xff] ^
            crc_table[5][(lo >> 16) & 0xff] ^ crc_table[4][lo >> 24] ^ crc_table[3][hi & 0xff] ^
            crc_table[2][(hi >> 8) & 0xff] ^ crc_table[1][(hi >> 16) & 0xff] ^
            crc_table[0][hi >> 24];
        data += 8;
        len -= 8;
    }
    while (len--) c = (c >> 8) ^ crc_table[0][(c ^ *data++) & 0xff];
    return c ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------ varints */

/* VarInt::decode, primitive/varint.rs:49-78.  shift advances 0,7,..,28,35;
 * u32::checked_shl fails only once shift >= 32, so a 5th byte's upper bits are
 * silently dropped and a 6th byte is Error::Overflow.  The byte is fetched
 * (try_get_u8) before the shift is checked, so under-run wins over overflow.
 * checked_add cannot overflow: every addend occupies bits above the sum so far. */
int no_varint_decode(const uint8_t *buf, size_t len, size_t *pos, int32_t *out) {
    uint32_t shift = 0, acc = 0;
    for (;;) {
        if (*pos >= len) return NO_TRY_GET;
        uint8_t byte = buf[(*pos)++];
        if (shift >= 32) return NO_OVERFLOW;
        if (byte & 0x80) {
            acc += (uint32_t)(byte & 0x7f) << shift;
            shift += 7;
        } else {
            acc += (uint32_t)byte << shift;
            break;
        }
    }
    /* de_zigzag :107-109 */
    *out = (int32_t)(acc >> 1) ^ -(int32_t)(acc & 1);
    return NO_OK;
}

/* LongVarInt::decode, varint.rs:266-295: the same with u64; the 11th byte overflows. */
int no_varlong_decode(const uint8_t *buf, size_t len, size_t *pos, int64_t *out) {
    uint32_t shift = 0;
    uint64_t acc = 0;
    for (;;) {
        if (*pos >= len) return NO_TRY_GET;
        uint8_t byte = buf[(*pos)++];
        if (shift >= 64) return NO_OVERFLOW;
        if (byte & 0x80) {
            acc += (uint64_t)(byte & 0x7f) << shift;
            shift += 7;
        } else {
            acc += (uint64_t)byte << shift;
            break;
        }
    }
    *out = (int64_t)(acc >> 1) ^ -(int64_t)(acc & 1); /* :303-306 */
    return NO_OK;
}

/* UnsignedVarInt::deserialize, varint.rs:476-534 (same loop, no zigzag) */
int no_uvarint_decode(const uint8_t *buf, size_t len, size_t *pos, uint32_t *out) {
    uint32_t shift = 0, acc = 0;
    for (;;) {
        if (*pos >= len) return NO_TRY_GET;
        uint8_t byte = buf[(*pos)++];
        if (shift >= 32) return NO_OVERFLOW;
        if (byte & 0x80) {
            acc += (uint32_t)(byte & 0x7f) << shift;
            shift += 7;
        } else {
            acc += (uint32_t)byte << shift;
            break;
        }
    }
    *out = acc;
    return NO_OK;
}

static uint32_t zigzag32(int32_t v) { /* varint.rs:102-104 */
    return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31);
}
static uint64_t zigzag64(int64_t v) { /* varint.rs:299-301 */
    return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
}

size_t no_uvarint_size(uint32_t v) { /* varint.rs:546-556 */
    size_t n = 1;
    while (v >= 0x80) {
        v >>= 7;
        n++;
    }
    return n;
}
size_t no_varint_size(int32_t v) { return no_uvarint_size(zigzag32(v)); } /* :186-196 */
size_t no_varlong_size(int64_t v) {                                       /* :379-389 */
    uint64_t z = zigzag64(v);
    size_t n = 1;
    while (z >= 0x80) {
        z >>= 7;
        n++;
    }
    return n;
}
size_t no_uvarint_encode(uint32_t v, uint8_t *out) { /* :435-450 */
    size_t n = 0;
    while (v >= 0x80) {
        out[n++] = (uint8_t)v | 0x80;
        v >>= 7;
    }
    out[n++] = (uint8_t)v;
    return n;
}
size_t no_varint_encode(int32_t v, uint8_t *out) { return no_uvarint_encode(zigzag32(v), out); }
size_t no_varlong_encode(int64_t v, uint8_t *out) { /* :248-264 */
    uint64_t z = zigzag64(v);
    size_t n = 0;
    while (z >= 0x80) {
        out[n++] = (uint8_t)z | 0x80;
        z >>= 7;
    }
    out[n++] = (uint8_t)z;
    return n;
}

/* --------------------------------------------------------------- attributes */

int no_compression(int16_t attributes, int *codec) { /* lib.rs:1685-1698 */
    int c = attributes & 7;
    if (c > 4) return NO_UNKNOWN_COMPRESSION;
    *codec = c;
    return NO_OK;
}

/* BatchAttribute::try_from(i16) then .compression(None).into(): lib.rs:1814-1850.
 * Only bits 3 (timestamp type), 4, 5 and 6 survive the round trip. */
int no_attributes_clear_compression(int16_t attributes, int16_t *out) {
    int c;
    int st = no_compression(attributes, &c);
    if (st) return st;
    *out = (int16_t)(attributes & (8 | 16 | 32 | 64));
    return NO_OK;
}

/* ------------------------------------------------------------------ records */

/* Octets::decode, record/codec.rs:65-76.  -1 is None; anything else is cast to
 * usize and handed to Bytes::split_to, which panics past the end (so does any
 * other negative length, which casts to a huge usize). */
static int octets_decode(const uint8_t *buf, size_t len, size_t *pos, no_octets *out) {
    int32_t n;
    int st = no_varint_decode(buf, len, pos, &n);
    if (st) return st;
    if (n == -1) {
        out->ptr = buf + *pos; /* None has no bytes; keep where they would start */
        out->len = -1;
        return NO_OK;
    }
    if (n < 0 || (size_t)n > len - *pos) return NO_PANIC;
    out->ptr = buf + *pos;
    out->len = n;
    *pos += (size_t)n;
    return NO_OK;
}

size_t no_octets_size(int64_t len) { /* codec.rs:34-45 */
    if (len < 0) return no_varint_size(-1);
    return no_varint_size((int32_t)len) + (size_t)len;
}
static size_t octets_size(const no_octets *o) { return no_octets_size(o->len); }

static size_t octets_encode(const no_octets *o, uint8_t *out) { /* codec.rs:47-63 */
    if (o->len < 0) return no_varint_encode(-1, out);
    size_t n = no_varint_encode((int32_t)o->len, out);
    memcpy(out + n, o->ptr, (size_t)o->len);
    return n + (size_t)o->len;
}

void no_record_free(no_record *r) {
    free(r->headers);
    r->headers = NULL;
    r->n_headers = 0;
}

/* Record::decode, record.rs:254-277, with VarIntSequence<Header>::decode
 * (codec.rs:283-302) and Header::decode (header.rs:55-65) inlined. */
int no_record_decode(const uint8_t *buf, size_t len, size_t *pos, no_record *out) {
    int st;
    memset(out, 0, sizeof *out);
    if ((st = no_varint_decode(buf, len, pos, &out->length))) return st;
    if (*pos >= len) return NO_PANIC; /* Buf::get_u8 on an empty buffer, record.rs:260 */
    out->attributes = buf[(*pos)++];
    if ((st = no_varlong_decode(buf, len, pos, &out->timestamp_delta))) return st;
    if ((st = no_varint_decode(buf, len, pos, &out->offset_delta))) return st;
    if ((st = octets_decode(buf, len, pos, &out->key))) return st;
    if ((st = octets_decode(buf, len, pos, &out->value))) return st;
    int32_t count;
    if ((st = no_varint_decode(buf, len, pos, &count))) return st;
    /* `length.0 as usize` then Vec::with_capacity: a negative count is a
     * capacity-overflow panic (codec.rs:291-295) */
    if (count < 0) return NO_PANIC;
    /* A header is at least two bytes, so a count beyond the remaining bytes can
     * only end in one of the errors below; walking finds which one first, as
     * the reference does, without reserving count entries. */
    size_t cap = 0;
    for (int32_t i = 0; i < count; i++) {
        no_header h;
        if ((st = octets_decode(buf, len, pos, &h.key))) break;
        if ((st = octets_decode(buf, len, pos, &h.value))) break;
        if (out->n_headers == cap) {
            cap = cap ? cap * 2 : 4;
            out->headers = (no_header *)realloc(out->headers, cap * sizeof(no_header));
        }
        out->headers[out->n_headers++] = h;
    }
    if (st) {
        no_record_free(out);
        return st;
    }
    return NO_OK;
}

static size_t headers_size(const no_record *r) { /* codec.rs:248-262, header.rs:31-41 */
    size_t n = no_varint_size((int32_t)r->n_headers);
    for (size_t i = 0; i < r->n_headers; i++)
        n += octets_size(&r->headers[i].key) + octets_size(&r->headers[i].value);
    return n;
}

size_t no_record_body_size(const no_record *r) { /* record.rs:363-372 */
    return 1 + no_varlong_size(r->timestamp_delta) + no_varint_size(r->offset_delta) +
           octets_size(&r->key) + octets_size(&r->value) + headers_size(r);
}

size_t no_record_size(const no_record *r) { /* record.rs:200-212 */
    return no_varint_size(r->length) + no_record_body_size(r);
}

size_t no_record_encode(const no_record *r, uint8_t *out) { /* record.rs:233-252 */
    size_t n = no_varint_encode(r->length, out); /* the stored length, verbatim */
    out[n++] = r->attributes;
    n += no_varlong_encode(r->timestamp_delta, out + n);
    n += no_varint_encode(r->offset_delta, out + n);
    n += octets_encode(&r->key, out + n);
    n += octets_encode(&r->value, out + n);
    n += no_varint_encode((int32_t)r->n_headers, out + n); /* codec.rs:264-281 */
    for (size_t i = 0; i < r->n_headers; i++) {            /* header.rs:43-53 */
        n += octets_encode(&r->headers[i].key, out + n);
        n += octets_encode(&r->headers[i].value, out + n);
    }
    return n;
}

size_t no_records_encode(const no_record *records, size_t n, uint8_t *out) { /* record.rs:214-231 */
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        if (out)
            total += no_record_encode(&records[i], out + total);
        else
            total += no_record_size(&records[i]);
    }
    return total;
}

/* ----------------------------------------------------------- deflated batch */

static uint64_t be64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
static uint32_t be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
static uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }
static void put64(uint8_t *p, uint64_t v) {
    for (int i = 7; i >= 0; i--) {
        p[i] = (uint8_t)v;
        v >>= 8;
    }
}
static void put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}
static void put16(uint8_t *p, uint16_t v) {
    p[0] = (uint8_t)(v >> 8);
    p[1] = (uint8_t)v;
}

#define FIXED_BATCH_LENGTH 49 /* deflated.rs:486-508 */

/* deflated::Batch::try_from(Bytes), deflated.rs:108-213, in the reference's
 * order of checks. */
int no_deflated_parse(const uint8_t *buf, size_t len, no_deflated_batch *out) {
    memset(out, 0, sizeof *out);
    /* :111-116 five try_get_* calls: 8 + 4 + 4 + 1 + 4 bytes */
    if (len < 8) return NO_TRY_GET;
    out->base_offset = (int64_t)be64(buf);
    if (len < 12) return NO_TRY_GET;
    out->batch_length = (int32_t)be32(buf + 8);
    if (len < 16) return NO_TRY_GET;
    out->partition_leader_epoch = (int32_t)be32(buf + 12);
    if (len < 17) return NO_TRY_GET;
    out->magic = (int8_t)buf[16];
    if (len < 21) return NO_TRY_GET;
    out->crc = be32(buf + 17);
    size_t rem = len - 21;
    /* :120-147 usize::try_from(batch_length) then three checked_sub (4, 1, 4) */
    if (out->batch_length < 0) return NO_TRY_FROM_INT;
    if (out->batch_length < 9) return NO_OVERFLOW;
    size_t crc_data_size = (size_t)out->batch_length - 9;
    if (crc_data_size > rem) return NO_OVERFLOW; /* :151-153 */
    out->crc_computed = no_crc32c(buf + 21, crc_data_size); /* :155-166, mismatch only logs */
    /* :168-175 eight more try_get_*: 2 + 4 + 8 + 8 + 8 + 2 + 4 + 4 = 40 bytes */
    if (rem < 40) return NO_TRY_GET;
    out->attributes = (int16_t)be16(buf + 21);
    out->last_offset_delta = (int32_t)be32(buf + 23);
    out->base_timestamp = (int64_t)be64(buf + 27);
    out->max_timestamp = (int64_t)be64(buf + 35);
    out->producer_id = (int64_t)be64(buf + 43);
    out->producer_epoch = (int16_t)be16(buf + 51);
    out->base_sequence = (int32_t)be32(buf + 53);
    out->record_count = be32(buf + 57);
    /* :177-190 */
    if (out->batch_length < FIXED_BATCH_LENGTH) return NO_OVERFLOW;
    size_t record_data_size = (size_t)out->batch_length - FIXED_BATCH_LENGTH;
    if (record_data_size > len - 61) return NO_OVERFLOW;
    out->record_data = buf + 61;
    out->record_data_len = record_data_size;
    return NO_OK;
}

size_t no_deflated_emit(const no_deflated_batch *b, uint8_t *out) { /* deflated.rs:84-106 */
    put64(out, (uint64_t)b->base_offset);
    put32(out + 8, (uint32_t)b->batch_length);
    put32(out + 12, (uint32_t)b->partition_leader_epoch);
    out[16] = (uint8_t)b->magic;
    put32(out + 17, b->crc);
    put16(out + 21, (uint16_t)b->attributes);
    put32(out + 23, (uint32_t)b->last_offset_delta);
    put64(out + 27, (uint64_t)b->base_timestamp);
    put64(out + 35, (uint64_t)b->max_timestamp);
    put64(out + 43, (uint64_t)b->producer_id);
    put16(out + 51, (uint16_t)b->producer_epoch);
    put32(out + 53, (uint32_t)b->base_sequence);
    put32(out + 57, b->record_count);
    if (b->record_data_len) memcpy(out + 61, b->record_data, b->record_data_len);
    return 61 + b->record_data_len;
}

void no_records_free(no_record *records, size_t n) {
    if (!records) return;
    for (size_t i = 0; i < n; i++) no_record_free(&records[i]);
    free(records);
}

/* Vec<Record>::try_from(deflated::Batch), deflated.rs:421-459.  Bytes after
 * the last record are ignored (:437-442). */
int no_inflate_records(const no_deflated_batch *b, no_record **records, size_t *n_records) {
    int codec;
    int st = no_compression(b->attributes, &codec);
    *records = NULL;
    *n_records = 0;
    if (st) return st; /* :444-446 compression()? on the slow branch */
    if (codec != 0) return NO_COMPRESSED;
    /* Vec::with_capacity(record_count): a record is at least 7 bytes, so a
     * count beyond the data can only fail; reserve what can succeed. */
    size_t cap = b->record_count;
    if (cap > b->record_data_len / 7 + 1) cap = b->record_data_len / 7 + 1;
    no_record *v = (no_record *)calloc(cap ? cap : 1, sizeof(no_record));
    size_t pos = 0, n = 0;
    for (uint32_t i = 0; i < b->record_count; i++) {
        no_record r;
        st = no_record_decode(b->record_data, b->record_data_len, &pos, &r);
        if (st) {
            no_records_free(v, n);
            return st;
        }
        if (n == cap) { /* unreachable for well-formed sizes; keeps the walk honest */
            cap = cap * 2 + 1;
            v = (no_record *)realloc(v, cap * sizeof(no_record));
        }
        v[n++] = r;
    }
    *records = v;
    *n_records = n;
    return NO_OK;
}

/* deflated::Batch::try_from(inflated::Batch), deflated.rs:392-409: record_count
 * = records.len(); record_data = (&[Record]).encode(); batch_length = 49 +
 * len (:302); crc over attributes..record_data (:318-326).  The inflated
 * batch_length and crc are ignored. */
int no_deflate(const no_deflated_batch *hdr, const no_record *records, size_t n, uint8_t *out,
               size_t out_cap, size_t *out_len) {
    int codec;
    int st = no_compression(hdr->attributes, &codec);
    if (st) return st;
    if (codec != 0) return NO_COMPRESSED;
    size_t data_len = no_records_encode(records, n, NULL);
    if (61 + data_len > out_cap) return NO_CAPACITY;
    if (FIXED_BATCH_LENGTH + data_len > 0x7fffffffu) return NO_TRY_FROM_INT; /* :302 */
    no_deflated_batch b = *hdr;
    b.record_count = (uint32_t)n;
    b.batch_length = (int32_t)(FIXED_BATCH_LENGTH + data_len);
    b.record_data = NULL;
    b.record_data_len = 0;
    b.crc = 0;
    no_deflated_emit(&b, out);
    no_records_encode(records, n, out + 61);
    uint32_t crc = no_crc32c(out + 21, 40 + data_len);
    put32(out + 17, crc);
    *out_len = 61 + data_len;
    return NO_OK;
}

/* inflated::Builder::size_in_bytes (inflated.rs:331-346) + Sequence::size_in_bytes
 * (codec.rs:482-491, which sizes each record's prefix as an UNSIGNED varint of
 * the body size via UnsignedVarInt::size_inclusive, varint.rs:536-543), and
 * Builder::crc (inflated.rs:429-446), which serialises the records with the
 * zigzag prefix actually written (record.rs:339-361). */
int no_builder_batch_length_crc(const no_deflated_batch *hdr, const no_record *records, size_t n,
                                int32_t *batch_length, uint32_t *crc) {
    size_t size = 4 + 1 + 4 + 2 + 4 + 8 + 8 + 8 + 2 + 4 + 4;
    size_t wire = 0;
    for (size_t i = 0; i < n; i++) {
        size_t body = no_record_body_size(&records[i]);
        size += no_uvarint_size((uint32_t)body) + body;
        wire += no_varint_size((int32_t)body) + body;
    }
    *batch_length = (int32_t)size;
    uint8_t *buf = (uint8_t *)malloc(40 + wire);
    put16(buf, (uint16_t)hdr->attributes);
    put32(buf + 2, (uint32_t)hdr->last_offset_delta);
    put64(buf + 6, (uint64_t)hdr->base_timestamp);
    put64(buf + 14, (uint64_t)hdr->max_timestamp);
    put64(buf + 22, (uint64_t)hdr->producer_id);
    put16(buf + 30, (uint16_t)hdr->producer_epoch);
    put32(buf + 32, (uint32_t)hdr->base_sequence);
    put32(buf + 36, (uint32_t)n);
    size_t p = 40;
    for (size_t i = 0; i < n; i++) {
        no_record r = records[i];
        r.length = (int32_t)no_record_body_size(&records[i]);
        p += no_record_encode(&r, buf + p);
    }
    *crc = no_crc32c(buf, p);
    free(buf);
    return NO_OK;
}

/* ------------------------------------------------------------------ framing */

/* Batch::next_element_seed, de.rs:1059-1092: while bytes remain, read
 * base_offset (i64) and batch_length (i32), Overflow if batch_length exceeds
 * what is left, and hand 12 + batch_length bytes to the batch decoder.  A
 * negative batch_length casts to a huge usize in BytesMut::with_capacity
 * (:1074) and panics there. */
int no_frame_split(const uint8_t *blob, size_t len, uint64_t *offsets, size_t cap, size_t *n) {
    size_t pos = 0, k = 0;
    while (pos < len) {
        if (len - pos < 8) return NO_TRY_GET;
        if (len - pos < 12) return NO_TRY_GET;
        int32_t batch_length = (int32_t)be32(blob + pos + 8);
        if (batch_length < 0) return NO_PANIC;
        if ((size_t)batch_length > len - pos - 12) return NO_OVERFLOW;
        if (k + 1 >= cap) return NO_CAPACITY;
        offsets[k++] = pos;
        pos += 12 + (size_t)batch_length;
    }
    if (k >= cap) return NO_CAPACITY;
    offsets[k] = pos;
    *n = k;
    return NO_OK;
}

/* de.rs:701-720: flexible versions carry uvarint(len + 1), Overflow when the
 * varint is 0; older versions a big-endian u32.  Length 0 means null. */
int no_records_prefix_decode(const uint8_t *buf, size_t len, int flexible, size_t *pos,
                             uint32_t *records_len, int *is_null) {
    uint32_t v;
    if (flexible) {
        int st = no_uvarint_decode(buf, len, pos, &v);
        if (st) return st;
        if (v == 0) return NO_OVERFLOW;
        v -= 1;
    } else {
        if (len - *pos < 4) return NO_TRY_GET;
        v = be32(buf + *pos);
        *pos += 4;
    }
    *records_len = v;
    *is_null = (v == 0);
    return NO_OK;
}

size_t no_records_prefix_encode(uint32_t records_len, int flexible, uint8_t *out) { /* ser.rs:484-489, 553-560 */
    if (flexible) return no_uvarint_encode(records_len + 1, out);
    put32(out, records_len);
    return 4;
}

/* ------------------------------------------------------------------ columns */

int no_decode_columns(const uint8_t *blob, const uint64_t *batch_off, size_t n_batches,
                      int by_value, no_columns *c, size_t rec_cap, size_t hdr_cap,
                      size_t *n_records, size_t *n_headers) {
    size_t nr = 0, nh = 0;
    for (size_t b = 0; b < n_batches; b++) {
        const uint8_t *buf = blob + batch_off[b];
        size_t len = (size_t)(batch_off[b + 1] - batch_off[b]);
        no_deflated_batch d;
        int st = no_deflated_parse(buf, len, &d);
        c->base_offset[b] = d.base_offset;
        c->batch_length[b] = d.batch_length;
        c->partition_leader_epoch[b] = d.partition_leader_epoch;
        c->magic[b] = d.magic;
        c->crc[b] = d.crc;
        c->attributes[b] = d.attributes;
        c->last_offset_delta[b] = d.last_offset_delta;
        c->base_timestamp[b] = d.base_timestamp;
        c->max_timestamp[b] = d.max_timestamp;
        c->producer_id[b] = d.producer_id;
        c->producer_epoch[b] = d.producer_epoch;
        c->base_sequence[b] = d.base_sequence;
        c->record_count[b] = d.record_count;
        c->crc_computed[b] = d.crc_computed;
        c->data_off[b] = batch_off[b] + 61;
        c->data_len[b] = (uint32_t)d.record_data_len;
        c->rec_base[b] = (uint32_t)nr;
        c->hdr_base[b] = (uint32_t)nh;
        c->hdr_count[b] = 0;
        no_record *recs = NULL;
        size_t n = 0;
        if (st == NO_OK && by_value) {
            /* inflated.rs:123-124: BatchAttribute::try_from before the records */
            int16_t cleared;
            st = no_attributes_clear_compression(d.attributes, &cleared);
            if (st == NO_OK) {
                int codec;
                no_compression(d.attributes, &codec);
                if (codec == 0) c->attributes[b] = cleared;
            }
        }
        if (st == NO_OK) st = no_inflate_records(&d, &recs, &n);
        c->status[b] = st;
        if (st != NO_OK) {
            c->record_count[b] = d.record_count;
            continue;
        }
        size_t h0 = nh;
        for (size_t i = 0; i < n; i++) {
            const no_record *r = &recs[i];
            if (nr >= rec_cap || nh + r->n_headers > hdr_cap) {
                no_records_free(recs, n);
                return NO_CAPACITY;
            }
            c->length[nr] = r->length;
            c->rattributes[nr] = r->attributes;
            c->timestamp_delta[nr] = r->timestamp_delta;
            c->offset_delta[nr] = r->offset_delta;
            /* a null span keeps the position right after its length varint */
            c->key_len[nr] = (int32_t)r->key.len;
            c->val_len[nr] = (int32_t)r->value.len;
            c->key_off[nr] = (uint32_t)(r->key.ptr - d.record_data);
            c->val_off[nr] = (uint32_t)(r->value.ptr - d.record_data);
            c->rhdr_base[nr] = (uint32_t)nh;
            c->rhdr_count[nr] = (int32_t)r->n_headers;
            for (size_t j = 0; j < r->n_headers; j++) {
                const no_header *h = &r->headers[j];
                c->hkey_len[nh] = (int32_t)h->key.len;
                c->hval_len[nh] = (int32_t)h->value.len;
                c->hkey_off[nh] = (uint32_t)(h->key.ptr - d.record_data);
                c->hval_off[nh] = (uint32_t)(h->value.ptr - d.record_data);
                nh++;
            }
            nr++;
        }
        c->hdr_count[b] = (uint32_t)(nh - h0);
        no_records_free(recs, n);
    }
    *n_records = nr;
    *n_headers = nh;
    return NO_OK;
}

/* The same columns without the reference's intermediate Vec<Record>: every field
 * goes from the wire straight into its column, no allocation a record or a header.
 * Not how the reference works: what a CPU decoder built for this output could do,
 * timed beside the faithful form as a second baseline.  Same checks in the same
 * order as no_record_decode, so the same outcome for every input (tests compare
 * the two). */
int no_decode_columns_lean(const uint8_t *blob, const uint64_t *batch_off, size_t n_batches,
                           int by_value, no_columns *c, size_t rec_cap, size_t hdr_cap,
                           size_t *n_records, size_t *n_headers) {
    size_t nr = 0, nh = 0;
    for (size_t b = 0; b < n_batches; b++) {
        const uint8_t *buf = blob + batch_off[b];
        size_t len = (size_t)(batch_off[b + 1] - batch_off[b]);
        no_deflated_batch d;
        int st = no_deflated_parse(buf, len, &d);
        c->base_offset[b] = d.base_offset;
        c->batch_length[b] = d.batch_length;
        c->partition_leader_epoch[b] = d.partition_leader_epoch;
        c->magic[b] = d.magic;
        c->crc[b] = d.crc;
        c->attributes[b] = d.attributes;
        c->last_offset_delta[b] = d.last_offset_delta;
        c->base_timestamp[b] = d.base_timestamp;
        c->max_timestamp[b] = d.max_timestamp;
        c->producer_id[b] = d.producer_id;
        c->producer_epoch[b] = d.producer_epoch;
        c->base_sequence[b] = d.base_sequence;
        c->record_count[b] = d.record_count;
        c->crc_computed[b] = d.crc_computed;
        c->data_off[b] = batch_off[b] + 61;
        c->data_len[b] = (uint32_t)d.record_data_len;
        c->rec_base[b] = (uint32_t)nr;
        c->hdr_base[b] = (uint32_t)nh;
        c->hdr_count[b] = 0;
        int codec = 0;
        if (st == NO_OK && by_value) {
            int16_t cleared;
            st = no_attributes_clear_compression(d.attributes, &cleared);
            if (st == NO_OK) {
                no_compression(d.attributes, &codec);
                if (codec == 0) c->attributes[b] = cleared;
            }
        }
        if (st == NO_OK) st = no_compression(d.attributes, &codec);
        if (st == NO_OK && codec != 0) st = NO_COMPRESSED;
        size_t r0 = nr, h0 = nh, pos = 0;
        const uint8_t *data = d.record_data;
        size_t dl = d.record_data_len;
        for (uint32_t i = 0; st == NO_OK && i < d.record_count; i++) {
            int32_t length, od, count;
            int64_t ts;
            no_octets key, value;
            if ((st = no_varint_decode(data, dl, &pos, &length))) break;
            if (pos >= dl) { st = NO_PANIC; break; }
            uint8_t attributes = data[pos++];
            if ((st = no_varlong_decode(data, dl, &pos, &ts))) break;
            if ((st = no_varint_decode(data, dl, &pos, &od))) break;
            if ((st = octets_decode(data, dl, &pos, &key))) break;
            if ((st = octets_decode(data, dl, &pos, &value))) break;
            if ((st = no_varint_decode(data, dl, &pos, &count))) break;
            if (count < 0) { st = NO_PANIC; break; }
            size_t hfirst = nh;
            for (int32_t j = 0; j < count; j++) {
                no_octets hk, hv;
                if ((st = octets_decode(data, dl, &pos, &hk))) break;
                if ((st = octets_decode(data, dl, &pos, &hv))) break;
                if (nh >= hdr_cap) return NO_CAPACITY;
                c->hkey_len[nh] = (int32_t)hk.len;
                c->hval_len[nh] = (int32_t)hv.len;
                c->hkey_off[nh] = (uint32_t)(hk.ptr - data);
                c->hval_off[nh] = (uint32_t)(hv.ptr - data);
                nh++;
            }
            if (st) break;
            if (nr >= rec_cap) return NO_CAPACITY;
            c->length[nr] = length;
            c->rattributes[nr] = attributes;
            c->timestamp_delta[nr] = ts;
            c->offset_delta[nr] = od;
            c->key_len[nr] = (int32_t)key.len;
            c->val_len[nr] = (int32_t)value.len;
            c->key_off[nr] = (uint32_t)(key.ptr - data);
            c->val_off[nr] = (uint32_t)(value.ptr - data);
            c->rhdr_base[nr] = (uint32_t)hfirst;
            c->rhdr_count[nr] = count;
            nr++;
        }
        c->status[b] = st;
        if (st != NO_OK) { /* a failed batch leaves no records */
            nr = r0;
            nh = h0;
            continue;
        }
        c->hdr_count[b] = (uint32_t)(nh - h0);
    }
    *n_records = nr;
    *n_headers = nh;
    return NO_OK;
}

static no_record *records_from_columns(const no_columns *c, size_t b, const uint8_t *base) {
    size_t n = c->record_count[b];
    no_record *recs = (no_record *)calloc(n ? n : 1, sizeof(no_record));
    for (size_t i = 0; i < n; i++) {
        size_t r = c->rec_base[b] + i;
        no_record *x = &recs[i];
        x->length = c->length[r];
        x->attributes = c->rattributes[r];
        x->timestamp_delta = c->timestamp_delta[r];
        x->offset_delta = c->offset_delta[r];
        x->key.len = c->key_len[r];
        x->key.ptr = base + c->key_off[r];
        x->value.len = c->val_len[r];
        x->value.ptr = base + c->val_off[r];
        x->n_headers = (size_t)c->rhdr_count[r];
        x->headers = (no_header *)calloc(x->n_headers ? x->n_headers : 1, sizeof(no_header));
        for (size_t j = 0; j < x->n_headers; j++) {
            size_t h = c->rhdr_base[r] + j;
            x->headers[j].key.len = c->hkey_len[h];
            x->headers[j].key.ptr = base + c->hkey_off[h];
            x->headers[j].value.len = c->hval_len[h];
            x->headers[j].value.ptr = base + c->hval_off[h];
        }
    }
    return recs;
}

static void header_from_columns(const no_columns *c, size_t b, no_deflated_batch *hdr) {
    memset(hdr, 0, sizeof *hdr);
    hdr->base_offset = c->base_offset[b];
    hdr->partition_leader_epoch = c->partition_leader_epoch[b];
    hdr->magic = c->magic[b];
    hdr->attributes = c->attributes[b];
    hdr->last_offset_delta = c->last_offset_delta[b];
    hdr->base_timestamp = c->base_timestamp[b];
    hdr->max_timestamp = c->max_timestamp[b];
    hdr->producer_id = c->producer_id[b];
    hdr->producer_epoch = c->producer_epoch[b];
    hdr->base_sequence = c->base_sequence[b];
}

int no_encode_columns(const no_columns *c, size_t n_batches, const uint8_t *src,
                      const uint64_t *src_data_off, uint8_t *out, size_t out_cap,
                      uint64_t *out_off) {
    size_t pos = 0;
    for (size_t b = 0; b < n_batches; b++) {
        out_off[b] = pos;
        if (c->status[b] != NO_OK) continue;
        size_t n = c->record_count[b];
        no_record *recs = records_from_columns(c, b, src + src_data_off[b]);
        no_deflated_batch hdr;
        header_from_columns(c, b, &hdr);
        size_t len = 0;
        int st = no_deflate(&hdr, recs, n, out + pos, out_cap - pos, &len);
        no_records_free(recs, n);
        if (st) return st;
        pos += len;
    }
    out_off[n_batches] = pos;
    return NO_OK;
}

int no_builder_columns(const no_columns *c, size_t n_batches, const uint8_t *src,
                       const uint64_t *src_data_off, int32_t *batch_length, uint32_t *crc) {
    for (size_t b = 0; b < n_batches; b++) {
        size_t n = c->record_count[b];
        no_record *recs = records_from_columns(c, b, src + src_data_off[b]);
        no_deflated_batch hdr;
        header_from_columns(c, b, &hdr);
        int st = no_builder_batch_length_crc(&hdr, recs, n, &batch_length[b], &crc[b]);
        no_records_free(recs, n);
        if (st) return st;
    }
    return NO_OK;
}

/* ------------------------------------------------------------------ callers */

/* by-value inflate: deflated parse, BatchAttribute round trip with the
 * compression cleared (inflated.rs:123-140), the records (deflated.rs:421-459) */
static int inflate_by_value(const uint8_t *buf, size_t len, no_deflated_batch *d, no_record **recs,
                            size_t *n) {
    int st = no_deflated_parse(buf, len, d);
    if (st) return st;
    int16_t cleared;
    st = no_attributes_clear_compression(d->attributes, &cleared);
    if (st) return st;
    st = no_inflate_records(d, recs, n);
    if (st) return st;
    d->attributes = cleared; /* only an uncompressed batch gets here */
    return NO_OK;
}

/* pg.rs:797-943 (check_time) and lite.rs:703-815 (!check_time) */
int no_produce_rows(const uint8_t *batch, size_t len, int64_t high, int check_time,
                    size_t row_cap, size_t hdr_cap, int16_t *attributes,
                    int64_t *row_offset, int64_t *row_timestamp,
                    uint64_t *row_key_pos, int64_t *row_key_len,
                    uint64_t *row_val_pos, int64_t *row_val_len,
                    int64_t *hdr_offset, uint64_t *hdr_key_pos, int64_t *hdr_key_len,
                    uint64_t *hdr_val_pos, int64_t *hdr_val_len,
                    size_t *n_rows, size_t *n_hdrs) {
    no_deflated_batch d;
    no_record *recs = NULL;
    size_t n = 0, nh = 0;
    *n_rows = *n_hdrs = 0;
    int st = inflate_by_value(batch, len, &d, &recs, &n); /* pg.rs:797 */
    if (st) return st;
    *attributes = d.attributes;
    for (size_t i = 0; i < n && st == NO_OK; i++) { /* pg.rs:867-897 */
        const no_record *r = &recs[i];
        if (i >= row_cap) { st = NO_CAPACITY; break; }
        row_offset[i] = (int64_t)((uint64_t)high + (uint64_t)i);
        int64_t ts = (int64_t)((uint64_t)d.base_timestamp + (uint64_t)r->timestamp_delta);
        if (check_time && ts < 0) { st = NO_TRY_FROM_INT; break; } /* lib.rs:2189-2193 */
        row_timestamp[i] = ts;
        row_key_pos[i] = (uint64_t)(r->key.ptr - batch);
        row_key_len[i] = r->key.len;
        row_val_pos[i] = (uint64_t)(r->value.ptr - batch);
        row_val_len[i] = r->value.len;
    }
    for (size_t i = 0; i < n && st == NO_OK; i++) { /* pg.rs:912-936 */
        const no_record *r = &recs[i];
        for (size_t j = 0; j < r->n_headers; j++) {
            if (nh >= hdr_cap) { st = NO_CAPACITY; break; }
            hdr_offset[nh] = (int64_t)((uint64_t)high + (uint64_t)i);
            hdr_key_pos[nh] = (uint64_t)(r->headers[j].key.ptr - batch);
            hdr_key_len[nh] = r->headers[j].key.len;
            hdr_val_pos[nh] = (uint64_t)(r->headers[j].value.ptr - batch);
            hdr_val_len[nh] = r->headers[j].value.len;
            nh++;
        }
    }
    no_records_free(recs, n);
    if (st) return st;
    *n_rows = n;
    *n_hdrs = nh;
    return NO_OK;
}

/* batch.rs:661-709 */
int no_combine(const uint8_t *const *batches, const size_t *lens, size_t n, uint8_t *out,
               size_t out_cap, size_t *out_len, int *is_none) {
    *out_len = 0;
    *is_none = 0;
    if (n == 0) { /* :666-668 */
        *is_none = 1;
        return NO_OK;
    }
    no_deflated_batch sink;
    no_record *all = NULL;
    size_t total = 0;
    int st = inflate_by_value(batches[0], lens[0], &sink, &all, &total); /* :670 */
    if (st) return st;
    for (size_t b = 1; b < n; b++) { /* :676-699 */
        no_deflated_batch d;
        no_record *recs = NULL;
        size_t m = 0;
        st = inflate_by_value(batches[b], lens[b], &d, &recs, &m); /* :677 */
        if (st) {
            no_records_free(all, total);
            return st;
        }
        all = (no_record *)realloc(all, (total + m + 1) * sizeof(no_record));
        for (size_t i = 0; i < m; i++) { /* :684-695; the moved record keeps its header vector */
            no_record r = recs[i];
            r.offset_delta = (int32_t)((uint32_t)r.offset_delta + (uint32_t)sink.last_offset_delta + 1u);
            r.timestamp_delta = (int64_t)((uint64_t)r.timestamp_delta +
                                          ((uint64_t)sink.base_timestamp - (uint64_t)d.base_timestamp));
            all[total + i] = r;
        }
        total += m;
        free(recs); /* the records themselves moved into `all` */
        sink.last_offset_delta =
            (int32_t)((uint32_t)sink.last_offset_delta + (uint32_t)d.last_offset_delta + 1u); /* :697 */
        if (d.max_timestamp > sink.max_timestamp) sink.max_timestamp = d.max_timestamp;      /* :698 */
    }
    st = no_deflate(&sink, all, total, out, out_cap, out_len); /* :706 */
    no_records_free(all, total);
    return st;
}

/* ---- the fetch seam of the SQL engines */

/* inflated::Batch::builder() with the setters the fetch uses (inflated.rs:311-329):
 * what is not set keeps its default */
typedef struct {
    no_deflated_batch hdr;
    no_record *records;
    size_t n, cap;
} fetch_builder;

static void fetch_builder_init(fetch_builder *fb, int64_t now) {
    memset(fb, 0, sizeof *fb);
    fb->hdr.base_offset = 0;
    fb->hdr.partition_leader_epoch = -1;
    fb->hdr.magic = 2;
    fb->hdr.attributes = 0;
    fb->hdr.last_offset_delta = 0;
    fb->hdr.base_timestamp = now; /* to_timestamp(&SystemTime::now()) */
    fb->hdr.max_timestamp = now;
    fb->hdr.producer_id = -1;
    fb->hdr.producer_epoch = 0;
    fb->hdr.base_sequence = 0;
}

/* batch_builder.build().and_then(TryInto::try_into): Record::try_from(Builder)
 * stores the body size as length (record.rs:298-314), then deflated.rs:392-409 */
static int fetch_builder_push(fetch_builder *fb, uint8_t *out, size_t out_cap, uint64_t *out_off,
                              size_t batch_cap, size_t *n_batches) {
    if (*n_batches >= batch_cap) return NO_CAPACITY;
    for (size_t i = 0; i < fb->n; i++) fb->records[i].length = (int32_t)no_record_body_size(&fb->records[i]);
    size_t at = (size_t)out_off[*n_batches], len = 0;
    int st = no_deflate(&fb->hdr, fb->records, fb->n, out + at, out_cap - at, &len);
    if (st) return st;
    out_off[++*n_batches] = at + len;
    return NO_OK;
}

static void fetch_builder_clear(fetch_builder *fb) {
    for (size_t i = 0; i < fb->n; i++) free(fb->records[i].headers);
    free(fb->records);
    fb->records = NULL;
    fb->n = fb->cap = 0;
}

/* pg.rs:1875-2052 (completed != NULL) and lite.rs:2760-2975 (completed == NULL) */
int no_fetch(size_t n, const int64_t *offset, const int16_t *attributes, const int64_t *timestamp,
             const int64_t *producer_id, const int16_t *producer_epoch, const uint8_t *completed,
             const uint32_t *key_off, const int32_t *key_len, const uint32_t *val_off,
             const int32_t *val_len, const uint32_t *hdr_base, const uint32_t *hdr_spans,
             const uint8_t *payload, int64_t now, uint8_t *out, size_t out_cap, uint64_t *out_off,
             size_t batch_cap, size_t *n_batches) {
    fetch_builder fb;
    int st = NO_OK;
    *n_batches = 0;
    out_off[0] = 0;
    fetch_builder_init(&fb, now);
    if (n == 0) /* pg.rs:2050-2052: an empty batch */
        return fetch_builder_push(&fb, out, out_cap, out_off, batch_cap, n_batches);
    fb.hdr.base_offset = offset[0]; /* pg.rs:1878-1909 */
    fb.hdr.attributes = attributes[0];
    fb.hdr.base_timestamp = timestamp[0];
    fb.hdr.producer_id = producer_id[0];
    fb.hdr.producer_epoch = producer_epoch[0];
    int have_previous = 0;
    int64_t previous_offset = 0;
    for (size_t i = 0; i < n; i++) {
        if (fb.hdr.attributes != attributes[i] || fb.hdr.producer_id != producer_id[i] ||
            fb.hdr.producer_epoch != producer_epoch[i]) { /* pg.rs:1933-1958 */
            st = fetch_builder_push(&fb, out, out_cap, out_off, batch_cap, n_batches);
            fetch_builder_clear(&fb);
            if (st) return st;
            fetch_builder_init(&fb, now);
            fb.hdr.base_offset = offset[i];
            fb.hdr.base_timestamp = timestamp[i];
            fb.hdr.attributes = attributes[i];
            fb.hdr.producer_id = producer_id[i];
            fb.hdr.producer_epoch = producer_epoch[i];
        }
        /* pg.rs:1965-1971: a record of a transaction still open ends the fetch,
         * unless it follows its predecessor directly */
        if (completed && !completed[i] && (!have_previous || previous_offset + 1 != offset[i])) break;
        int64_t delta = (int64_t)((uint64_t)offset[i] - (uint64_t)fb.hdr.base_offset);
        if (delta < INT32_MIN || delta > INT32_MAX) { /* i32::try_from, pg.rs:1973 */
            fetch_builder_clear(&fb);
            *n_batches = 0;
            return NO_TRY_FROM_INT;
        }
        if (fb.n == fb.cap) {
            fb.cap = fb.cap ? 2 * fb.cap : 16;
            fb.records = (no_record *)realloc(fb.records, fb.cap * sizeof(no_record));
        }
        no_record *r = &fb.records[fb.n++];
        memset(r, 0, sizeof *r);
        r->attributes = 0; /* Record::builder() */
        r->offset_delta = (int32_t)delta;
        r->timestamp_delta = (int64_t)((uint64_t)timestamp[i] - (uint64_t)fb.hdr.base_timestamp); /* pg.rs:1975-1984 */
        r->key.ptr = payload + key_off[i];
        r->key.len = key_len[i] < 0 ? -1 : key_len[i];
        r->value.ptr = payload + val_off[i];
        r->value.len = val_len[i] < 0 ? -1 : val_len[i];
        r->n_headers = hdr_base[i + 1] - hdr_base[i]; /* pg.rs:2004-2036 */
        r->headers = r->n_headers ? (no_header *)calloc(r->n_headers, sizeof(no_header)) : NULL;
        for (size_t j = 0; j < r->n_headers; j++) {
            const uint32_t *h = hdr_spans + 4 * ((size_t)hdr_base[i] + j);
            r->headers[j].key.ptr = payload + h[0];
            r->headers[j].key.len = (int32_t)h[1] < 0 ? -1 : (int32_t)h[1];
            r->headers[j].value.ptr = payload + h[2];
            r->headers[j].value.len = (int32_t)h[3] < 0 ? -1 : (int32_t)h[3];
        }
        have_previous = 1;
        previous_offset = offset[i];
        fb.hdr.last_offset_delta = (int32_t)delta; /* pg.rs:2040-2042 */
    }
    st = fetch_builder_push(&fb, out, out_cap, out_off, batch_cap, n_batches); /* pg.rs:2049 */
    fetch_builder_clear(&fb);
    if (st) *n_batches = 0;
    return st;
}

/* ---- log compaction */

/* BTreeSet<Bytes> / BTreeMap<Bytes, i32> of the reference, as a chained hash map
 * (only membership and the value of a key matter here, not the order) */
typedef struct key_node {
    const uint8_t *ptr;
    size_t len;
    int32_t value;
    struct key_node *next;
} key_node;

typedef struct {
    key_node **slot;
    size_t n_slots;
} key_map;

static void key_map_init(key_map *m, size_t expect) {
    m->n_slots = 64;
    while (m->n_slots < 2 * expect) m->n_slots *= 2;
    m->slot = (key_node **)calloc(m->n_slots, sizeof(key_node *));
}

static void key_map_free(key_map *m) {
    for (size_t i = 0; i < m->n_slots; i++)
        for (key_node *k = m->slot[i]; k;) {
            key_node *next = k->next;
            free(k);
            k = next;
        }
    free(m->slot);
}

static key_node **key_map_find(key_map *m, const uint8_t *ptr, size_t len) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < len; i++) h = (h ^ ptr[i]) * 1099511628211ull;
    key_node **at = &m->slot[h & (m->n_slots - 1)];
    while (*at && !((*at)->len == len && memcmp((*at)->ptr, ptr, len) == 0)) at = &(*at)->next;
    return at;
}

/* insert: returns 1 when the key was there already (its value is replaced) */
static int key_map_insert(key_map *m, const uint8_t *ptr, size_t len, int32_t value) {
    key_node **at = key_map_find(m, ptr, len);
    if (*at) {
        (*at)->value = value;
        return 1;
    }
    key_node *k = (key_node *)calloc(1, sizeof *k);
    k->ptr = ptr;
    k->len = len;
    k->value = value;
    *at = k;
    return 0;
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : x > y;
}

/* inflated::Batch::compact, inflated.rs:235-275.  Returns the number of records
 * the reference counts as removed and marks the records retain() keeps. */
static size_t compact_batch(const no_record *recs, size_t n, key_map *head, uint8_t *keep) {
    key_map last; /* last_delta_offset_for_key */
    key_map_init(&last, n);
    size_t removed = 0;
    for (size_t i = 0; i < n; i++) { /* :239-253 */
        if (recs[i].key.len < 0) continue;
        if (*key_map_find(head, recs[i].key.ptr, (size_t)recs[i].key.len)) {
            removed++;
            continue;
        }
        if (key_map_insert(&last, recs[i].key.ptr, (size_t)recs[i].key.len, recs[i].offset_delta)) removed++;
    }
    for (size_t i = 0; i < n; i++) keep[i] = 1;
    if (removed > 0) { /* :257-264: retained by the VALUE of offset_delta */
        int32_t *vals = (int32_t *)malloc((n + 1) * sizeof(int32_t)); /* delta_offsets_to_retain */
        size_t nv = 0;
        for (size_t s = 0; s < last.n_slots; s++)
            for (key_node *k = last.slot[s]; k; k = k->next) vals[nv++] = k->value;
        qsort(vals, nv, sizeof(int32_t), cmp_i32);
        for (size_t i = 0; i < n; i++)
            if (recs[i].key.len >= 0)
                keep[i] = bsearch(&recs[i].offset_delta, vals, nv, sizeof(int32_t), cmp_i32) != NULL;
        free(vals);
    }
    key_map_free(&last);
    return removed;
}

/* slate/storage.rs:283-312 over the batches of one partition, oldest first */
int no_compact_partition(const uint8_t *const *batches, const size_t *lens, size_t n, uint8_t *out,
                         size_t out_cap, uint64_t *out_off, int64_t *removed, int32_t *action) {
    key_map head;
    size_t total_keys = 0;
    no_deflated_batch *d = (no_deflated_batch *)calloc(n + 1, sizeof *d);
    no_record **recs = (no_record **)calloc(n + 1, sizeof *recs);
    size_t *counts = (size_t *)calloc(n + 1, sizeof *counts);
    int st = NO_OK;
    for (size_t b = 0; b < n && st == NO_OK; b++) { /* InflatedBatch::try_from(deflated.clone())? */
        st = inflate_by_value(batches[b], lens[b], &d[b], &recs[b], &counts[b]);
        total_keys += counts[b];
    }
    key_map_init(&head, total_keys);
    size_t at = 0;
    /* the reference walks newest to oldest; outputs are laid out oldest first, so
     * the bytes of each rewritten batch are produced into a scratch area first */
    uint8_t **piece = (uint8_t **)calloc(n + 1, sizeof *piece);
    size_t *piece_len = (size_t *)calloc(n + 1, sizeof *piece_len);
    for (size_t r = 0; r < n && st == NO_OK; r++) {
        size_t b = n - 1 - r;
        removed[b] = 0;
        action[b] = NO_COMPACT_UNCHANGED;
        if (d[b].attributes & 0x20) continue; /* BatchAttribute.control, storage.rs:291-293 */
        uint8_t *keep = (uint8_t *)malloc(counts[b] + 1);
        size_t gone = compact_batch(recs[b], counts[b], &head, keep); /* :296 */
        for (size_t i = 0; i < counts[b]; i++) /* head.extend(keys), :295, 297 */
            if (recs[b][i].key.len >= 0)
                key_map_insert(&head, recs[b][i].key.ptr, (size_t)recs[b][i].key.len, 0);
        removed[b] = (int64_t)gone;
        if (gone) {
            size_t m = 0;
            no_record *kept = (no_record *)calloc(counts[b] + 1, sizeof *kept);
            for (size_t i = 0; i < counts[b]; i++)
                if (keep[i]) kept[m++] = recs[b][i];
            if (m == 0) {
                action[b] = NO_COMPACT_DELETED; /* storage.rs:305-307 */
            } else {
                /* into_builder().build() (inflated.rs:266-268): lengths become body sizes;
                 * then compaction.batch.try_into() (storage.rs:309) */
                for (size_t i = 0; i < m; i++) kept[i].length = (int32_t)no_record_body_size(&kept[i]);
                size_t cap = 61;
                for (size_t i = 0; i < m; i++) cap += no_record_size(&kept[i]);
                piece[b] = (uint8_t *)malloc(cap);
                st = no_deflate(&d[b], kept, m, piece[b], cap, &piece_len[b]);
                action[b] = NO_COMPACT_REWRITTEN;
            }
            free(kept);
        }
        free(keep);
    }
    out_off[0] = 0;
    for (size_t b = 0; b < n; b++) {
        if (st == NO_OK && piece_len[b]) {
            if (at + piece_len[b] > out_cap) st = NO_CAPACITY;
            else memcpy(out + at, piece[b], piece_len[b]), at += piece_len[b];
        }
        out_off[b + 1] = at;
        free(piece[b]);
        no_records_free(recs[b], counts[b]);
    }
    key_map_free(&head);
    free(piece);
    free(piece_len);
    free(d);
    free(recs);
    free(counts);
    return st;
}

/* ------------------------------------------------------------------ Snappy */

/* bytes::read_varu64 of the snap crate: LEB128, up to ten bytes; a shift of 64 and more fails,
 * bits shifted out below that are lost; no terminating byte fails.  Header::read then refuses a
 * length above u32::MAX (Error::TooBig). */
int no_snappy_raw_len(const uint8_t *in, size_t n, size_t *header_len, uint64_t *len) {
    uint64_t v = 0;
    unsigned shift = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t b = in[i];
        if (shift >= 64) return NO_INFLATE;
        if (b < 0x80) {
            v |= (uint64_t)b << shift;
            if (v > 0xffffffffull) return NO_INFLATE;
            *header_len = i + 1;
            *len = v;
            return NO_OK;
        }
        v |= (uint64_t)(b & 0x7f) << shift;
        shift += 7;
    }
    return NO_INFLATE; /* (Error::Header: no terminator; an empty input is Error::Empty) */
}

/* snap::raw::Decoder::decompress into a buffer of the announced length */
int no_snappy_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    size_t s = 0, d = 0;
    uint64_t want = 0;
    *out_len = 0;
    if (n == 0) return NO_INFLATE;
    if (no_snappy_raw_len(in, n, &s, &want)) return NO_INFLATE;
    if (want > cap) return NO_CAPACITY;
    while (s < n) {
        uint8_t tag = in[s++];
        if ((tag & 3) == 0) {
            uint64_t len = (uint64_t)(tag >> 2) + 1;
            if (len >= 61) {
                size_t extra = (size_t)len - 60;
                if (s + 4 > n) return NO_INFLATE; /* (a literal of 61 bytes and more has them behind it) */
                uint32_t word = (uint32_t)in[s] | (uint32_t)in[s + 1] << 8 | (uint32_t)in[s + 2] << 16 | (uint32_t)in[s + 3] << 24;
                if (extra < 4) word &= (1u << (8 * extra)) - 1;
                len = (uint64_t)word + 1;
                s += extra;
            }
            if ((uint64_t)(n - s) < len || want - d < len) return NO_INFLATE;
            memcpy(out + d, in + s, (size_t)len);
            s += (size_t)len;
            d += (size_t)len;
        } else {
            size_t extra = (tag & 3) == 1 ? 1 : (tag & 3) == 2 ? 2 : 4;
            if (n - s < extra) return NO_INFLATE; /* Error::CopyRead */
            uint64_t offset, len;
            if ((tag & 3) == 1) {
                len = 4 + ((tag >> 2) & 7);
                offset = ((uint64_t)(tag >> 5) << 8) | in[s];
            } else if ((tag & 3) == 2) {
                len = 1 + (tag >> 2);
                offset = (uint64_t)in[s] | (uint64_t)in[s + 1] << 8;
            } else {
                len = 1 + (tag >> 2);
                offset = (uint64_t)in[s] | (uint64_t)in[s + 1] << 8 | (uint64_t)in[s + 2] << 16 | (uint64_t)in[s + 3] << 24;
            }
            s += extra;
            if (offset == 0 || offset > d) return NO_INFLATE; /* Error::Offset */
            if (want - d < len) return NO_INFLATE;             /* Error::CopyWrite */
            for (uint64_t i = 0; i < len; i++, d++) out[d] = out[d - (size_t)offset];
        }
    }
    if (d != want) return NO_INFLATE; /* Error::HeaderMismatch */
    *out_len = d;
    return NO_OK;
}

/* lib.rs:1717-1751 */
int no_snappy_inflate(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    static const uint8_t magic[8] = {0x82, 'S', 'N', 'A', 'P', 'P', 'Y', 0};
    *out_len = 0;
    if (n >= 8 && memcmp(in, magic, 8) == 0) {
        if (n < 20) return NO_INFLATE; /* (split_at past the end: the reference panics) */
        in += 20;
        n -= 20;
    }
    return no_snappy_raw(in, n, out, cap, out_len);
}

/* ------------------------------------------------------------------ LZ4 */

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

/* XXH32 (xxhash's published algorithm) */
uint32_t no_xxh32(const uint8_t *p, size_t len, uint32_t seed) {
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t *end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = rotl32(v1 + le32(p) * P2, 13) * P1;
            v2 = rotl32(v2 + le32(p + 4) * P2, 13) * P1;
            v3 = rotl32(v3 + le32(p + 8) * P2, 13) * P1;
            v4 = rotl32(v4 + le32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p + 16 <= end);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) {
        h = rotl32(h + le32(p) * P3, 17) * P4;
        p += 4;
    }
    while (p < end) {
        h = rotl32(h + (*p) * P5, 11) * P1;
        p++;
    }
    h ^= h >> 15;
    h *= P2;
    h ^= h >> 13;
    h *= P3;
    h ^= h >> 16;
    return h;
}

/* One block of the LZ4 block format into out[d ..), which may copy from out[low .. d): LZ4_decompress_safe's
 * outcomes for a block decoded into room for block_max bytes (what LZ4F_decompress gives it).  A sequence is a
 * token, the literals' length (15: more bytes follow, each added, until one is not 255), the literals, and
 * unless the block ends there an offset of two bytes and the match length (15: likewise; four more than
 * written).  liblz4's checks: a sequence with a match has eight bytes of input and more behind its literals
 * and twelve bytes of room and more; otherwise it is the last, and its literals end the block exactly; the
 * bytes of a match length leave four bytes of input behind them; the offset is neither 0 nor beyond out[low];
 * a match ends five bytes and more before the end of the room. */
static int lz4_block(const uint8_t *in, size_t n, uint8_t *out, size_t low, size_t *d_io, size_t room_end) {
    size_t s = 0, d = *d_io;
    if (n == 0) return NO_INFLATE;
    for (;;) {
        if (s >= n) return NO_INFLATE;
        uint8_t token = in[s++];
        size_t len = token >> 4;
        if (len == 15) {
            uint8_t b;
            do {
                if (s >= n) return NO_INFLATE;
                b = in[s++];
                len += b;
            } while (b == 255);
        }
        if (len > n - s || len > room_end - d) return NO_INFLATE;
        if ((n - s - len) < 8 || (room_end - d - len) < 12) {
            if (s + len != n) return NO_INFLATE; /* (the block's last sequence: literals to its very end) */
            memcpy(out + d, in + s, len);
            d += len;
            break;
        }
        memcpy(out + d, in + s, len);
        s += len;
        d += len;
        size_t offset = (size_t)in[s] | (size_t)in[s + 1] << 8;
        s += 2;
        size_t mlen = token & 15;
        if (mlen == 15) {
            uint8_t b;
            do {
                b = in[s++]; /* (eight bytes and more were left: this one is there) */
                mlen += b;
                if (s + 4 > n) return NO_INFLATE;
            } while (b == 255);
        }
        mlen += 4;
        if (offset == 0 || offset > d - low) return NO_INFLATE;
        if (mlen > room_end - d || room_end - d - mlen < 5) return NO_INFLATE; /* (the last five bytes are literals) */
        for (size_t i = 0; i < mlen; i++, d++) out[d] = out[d - offset];
    }
    *d_io = d;
    return NO_OK;
}

/* The frame's descriptor: sets what the flags say; returns the bytes of the header, 0 when it is refused. */
static size_t lz4_header(const uint8_t *in, size_t n, int *linked, int *block_sums, int *content_sum, int *has_size,
                         uint64_t *content, size_t *block_max) {
    if (n < 7 || le32(in) != 0x184D2204u) return 0;
    uint8_t flg = in[4], bd = in[5];
    if ((flg >> 6) != 1 || (flg & 2) || (bd & 0x8f) || ((bd >> 4) & 7) < 4) return 0;
    if (flg & 1) return 0; /* a dictionary id: not handled */
    *linked = !(flg & 0x20), *block_sums = (flg >> 4) & 1, *has_size = (flg >> 3) & 1, *content_sum = (flg >> 2) & 1;
    *block_max = (size_t)1 << (8 + 2 * ((bd >> 4) & 7)); /* 4: 64 KiB .. 7: 4 MiB */
    size_t s = 6;
    *content = 0;
    if (*has_size) {
        if (n < s + 8) return 0;
        for (int i = 0; i < 8; i++) *content |= (uint64_t)in[s + i] << (8 * i);
        s += 8;
    }
    if (n < s + 1) return 0;
    if (((no_xxh32(in + 4, s - 4, 0) >> 8) & 0xff) != in[s]) return 0;
    return s + 1;
}

/* Room that holds the frame's output whatever its blocks hold: every block at the maximal block size. */
static size_t lz4_frame_room(const uint8_t *in, size_t n) {
    int linked, block_sums, content_sum, has_size;
    uint64_t content;
    size_t block_max, room = 0;
    size_t s = lz4_header(in, n, &linked, &block_sums, &content_sum, &has_size, &content, &block_max);
    if (!s) return 0;
    while (n >= s + 4) {
        uint32_t word = le32(in + s);
        s += 4;
        if (word == 0) break;
        size_t size = word & 0x7fffffffu;
        if (size > block_max || size > n - s) break;
        room += block_max;
        s += size + (block_sums ? 4 : 0);
    }
    return room;
}

/* cap: at least lz4_frame_room */
int no_lz4_frame(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    int linked, block_sums, content_sum, has_size;
    uint64_t content;
    size_t block_max;
    *out_len = 0;
    size_t s = lz4_header(in, n, &linked, &block_sums, &content_sum, &has_size, &content, &block_max);
    if (!s) return NO_INFLATE;
    size_t d = 0;
    for (;;) {
        if (n < s + 4) return NO_INFLATE;
        uint32_t word = le32(in + s);
        s += 4;
        if (word == 0) break; /* the end mark */
        size_t size = word & 0x7fffffffu;
        if (size > block_max || size > n - s) return NO_INFLATE;
        if (cap - d < block_max) return NO_CAPACITY;
        if (word & 0x80000000u) {
            memcpy(out + d, in + s, size);
            d += size;
        } else {
            size_t low = linked ? (d > 65536 ? d - 65536 : 0) : d;
            int st = lz4_block(in + s, size, out, low, &d, d + block_max);
            if (st) return st;
        }
        if (block_sums) {
            if (n < s + size + 4 || no_xxh32(in + s, size, 0) != le32(in + s + size)) return NO_INFLATE;
            s += 4;
        }
        s += size;
    }
    if (content_sum) {
        if (n < s + 4 || no_xxh32(out, d, 0) != le32(in + s)) return NO_INFLATE;
    }
    if (has_size && content != d) return NO_INFLATE;
    *out_len = d;
    return NO_OK;
}

/* ------------------------------------------------------------------ gzip */

/* DEFLATE's bits come lowest bit of a byte first (RFC 1951, 3.1.1) */
typedef struct {
    const uint8_t *in;
    size_t n, pos;
    uint64_t hold;
    int bits;
    uint8_t *out;
    size_t cap, d;
} zstream;

static int zneed(zstream *z, int need) { /* 0 when `need` bits (up to 32) are in hold */
    while (z->bits < need) {
        if (z->pos >= z->n) return -1;
        z->hold |= (uint64_t)z->in[z->pos++] << z->bits;
        z->bits += 8;
    }
    return 0;
}

static int zbits(zstream *z, int need, uint32_t *v) {
    if (zneed(z, need)) return -1;
    *v = (uint32_t)(z->hold & ((1ull << need) - 1));
    z->hold >>= need;
    z->bits -= need;
    return 0;
}

/* A canonical Huffman code (RFC 1951, 3.2.2): count[len] codes of each length, the symbols in the order of
 * their codes.  Returns 0 for a complete set of lengths, a negative number for an over-subscribed one, a
 * positive number for an incomplete one. */
typedef struct {
    uint16_t count[16], symbol[288];
} zcode;

static int zconstruct(zcode *h, const uint16_t *length, int n) {
    uint16_t offs[16];
    for (int len = 0; len <= 15; len++) h->count[len] = 0;
    for (int i = 0; i < n; i++) h->count[length[i]]++;
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + h->count[len]);
    for (int i = 0; i < n; i++)
        if (length[i]) h->symbol[offs[length[i]]++] = (uint16_t)i;
    return left;
}

/* the next symbol: a code's bits come highest first; -1 out of input, -2 no such code */
static int zdecode(zstream *z, const zcode *h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        uint32_t bit;
        if (zbits(z, 1, &bit)) return -1;
        code |= (int)bit;
        int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -2;
}

static int zcodes(zstream *z, const zcode *lencode, const zcode *distcode) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        int sym = zdecode(z, lencode);
        if (sym < 0) return NO_INFLATE;
        if (sym < 256) {
            if (z->d >= z->cap) return NO_INFLATE;
            z->out[z->d++] = (uint8_t)sym;
        } else if (sym == 256) {
            return NO_OK;
        } else {
            sym -= 257;
            if (sym >= 29) return NO_INFLATE;
            uint32_t extra;
            if (zbits(z, lext[sym], &extra)) return NO_INFLATE;
            size_t len = lbase[sym] + extra;
            int ds = zdecode(z, distcode);
            if (ds < 0 || ds >= 30) return NO_INFLATE;
            if (zbits(z, dext[ds], &extra)) return NO_INFLATE;
            size_t dist = dbase[ds] + extra;
            if (dist > z->d) return NO_INFLATE;
            if (len > z->cap - z->d) return NO_INFLATE;
            for (size_t i = 0; i < len; i++, z->d++) z->out[z->d] = z->out[z->d - dist];
        }
    }
}

/* RFC 1951 over in[pos ..): the blocks up to the final one.  cap: the bytes the member says it holds. */
static int zinflate(zstream *z) {
    for (;;) {
        uint32_t last, type;
        if (zbits(z, 1, &last) || zbits(z, 2, &type)) return NO_INFLATE;
        if (type == 0) {
            /* stored: the rest of the byte is dropped, LEN and its complement, LEN bytes */
            z->hold = 0;
            z->bits = 0;
            if (z->n - z->pos < 4) return NO_INFLATE;
            uint32_t len = (uint32_t)z->in[z->pos] | (uint32_t)z->in[z->pos + 1] << 8;
            uint32_t nlen = (uint32_t)z->in[z->pos + 2] | (uint32_t)z->in[z->pos + 3] << 8;
            if ((len ^ 0xffffu) != nlen) return NO_INFLATE;
            z->pos += 4;
            if (z->n - z->pos < len || z->cap - z->d < len) return NO_INFLATE;
            memcpy(z->out + z->d, z->in + z->pos, len);
            z->pos += len;
            z->d += len;
        } else if (type == 1) {
            zcode lencode, distcode;
            uint16_t lengths[288];
            int i = 0;
            for (; i < 144; i++) lengths[i] = 8;
            for (; i < 256; i++) lengths[i] = 9;
            for (; i < 280; i++) lengths[i] = 7;
            for (; i < 288; i++) lengths[i] = 8;
            zconstruct(&lencode, lengths, 288);
            for (i = 0; i < 30; i++) lengths[i] = 5;
            zconstruct(&distcode, lengths, 30);
            int st = zcodes(z, &lencode, &distcode);
            if (st) return st;
        } else if (type == 2) {
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint32_t nlen, ndist, ncode;
            if (zbits(z, 5, &nlen) || zbits(z, 5, &ndist) || zbits(z, 4, &ncode)) return NO_INFLATE;
            nlen += 257, ndist += 1, ncode += 4;
            if (nlen > 286 || ndist > 30) return NO_INFLATE;
            uint16_t lengths[320];
            zcode lencode, distcode;
            uint32_t i;
            for (i = 0; i < ncode; i++) {
                uint32_t v;
                if (zbits(z, 3, &v)) return NO_INFLATE;
                lengths[order[i]] = (uint16_t)v;
            }
            for (; i < 19; i++) lengths[order[i]] = 0;
            if (zconstruct(&lencode, lengths, 19) != 0) return NO_INFLATE; /* (the code of the code lengths: complete) */
            uint32_t index = 0;
            while (index < nlen + ndist) {
                int sym = zdecode(z, &lencode);
                if (sym < 0) return NO_INFLATE;
                if (sym < 16) {
                    lengths[index++] = (uint16_t)sym;
                } else {
                    uint32_t len = 0, rep;
                    if (sym == 16) {
                        if (index == 0) return NO_INFLATE;
                        len = lengths[index - 1];
                        if (zbits(z, 2, &rep)) return NO_INFLATE;
                        rep += 3;
                    } else if (sym == 17) {
                        if (zbits(z, 3, &rep)) return NO_INFLATE;
                        rep += 3;
                    } else {
                        if (zbits(z, 7, &rep)) return NO_INFLATE;
                        rep += 11;
                    }
                    if (index + rep > nlen + ndist) return NO_INFLATE;
                    while (rep--) lengths[index++] = (uint16_t)len;
                }
            }
            if (lengths[256] == 0) return NO_INFLATE; /* no end of block */
            if (zconstruct(&lencode, lengths, (int)nlen) != 0) return NO_INFLATE;
            int left = zconstruct(&distcode, lengths + nlen, (int)ndist);
            /* (a distance code of one single code, of one bit, is the one incomplete set in use) */
            if (left < 0 || (left > 0 && !(distcode.count[1] == 1 && distcode.count[0] == ndist - 1))) return NO_INFLATE;
            int st = zcodes(z, &lencode, &distcode);
            if (st) return st;
        } else {
            return NO_INFLATE;
        }
        if (last) return NO_OK;
    }
}

/* The header of a gzip member (RFC 1952, 2.3): the bytes it takes, 0 when it is refused. */
static size_t gzip_header(const uint8_t *in, size_t n) {
    if (n < 18 || in[0] != 0x1f || in[1] != 0x8b || in[2] != 8) return 0;
    uint8_t flg = in[3];
    if (flg & 0xe2) return 0; /* reserved bits, or a header checksum */
    size_t s = 10;
    if (flg & 4) {
        if (n < s + 2) return 0;
        size_t xlen = (size_t)in[s] | (size_t)in[s + 1] << 8;
        s += 2;
        if (n - s < xlen) return 0;
        s += xlen;
    }
    for (int field = 0; field < 2; field++) {
        if (flg & (field ? 16 : 8)) {
            while (s < n && in[s]) s++;
            if (s >= n) return 0;
            s++;
        }
    }
    return n - s >= 8 ? s : 0;
}

/* cap: at least the member's length field (its last four bytes) */
int no_gzip_member(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    *out_len = 0;
    size_t s = gzip_header(in, n);
    if (!s) return NO_INFLATE;
    uint32_t isize = le32(in + n - 4);
    if (isize > cap) return NO_CAPACITY;
    zstream z = {in, n - 8, s, 0, 0, out, isize, 0};
    int st = zinflate(&z);
    if (st) return st;
    /* the trailer follows the last block's last byte, and ends the data */
    if (z.pos != n - 8 || z.d != isize) return NO_INFLATE;
    *out_len = z.d;
    return NO_OK;
}

/* ------------------------------------------------------------------ zstd
 * Compression::inflator's Zstd arm (nisshi-sans-io/src/lib.rs:1756-1765): zstd::stream::read::Decoder of the
 * zstd crate (0.13.2 in Cargo.lock, which builds libzstd; neither is vendored) over the record data.  Restated
 * from the published format (RFC 8878): frames of blocks; a compressed block is a literals section (raw, one
 * byte repeated, or Huffman coded in one or four streams read backwards) and a sequences section (literal
 * length, match length and offset codes from three FSE tables, interleaved in one stream read backwards).
 * The reference reads the stream as far as its records need it; here, as for the other codecs, every frame
 * of the record data must be sound. */

static uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

/* XXH64, of which a frame's content checksum is the low 32 bits */
uint64_t no_xxh64(const uint8_t *p, size_t len, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                   P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    const uint8_t *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v[4] = {seed + P1 + P2, seed + P2, seed, seed - P1};
        do {
            for (int i = 0; i < 4; i++) {
                v[i] = rotl64(v[i] + le64(p) * P2, 31) * P1;
                p += 8;
            }
        } while (p + 32 <= end);
        h = rotl64(v[0], 1) + rotl64(v[1], 7) + rotl64(v[2], 12) + rotl64(v[3], 18);
        for (int i = 0; i < 4; i++) h = (h ^ (rotl64(v[i] * P2, 31) * P1)) * P1 + P4;
    } else {
        h = seed + P5;
    }
    h += len;
    for (; p + 8 <= end; p += 8) h = rotl64(h ^ (rotl64(le64(p) * P2, 31) * P1), 27) * P1 + P4;
    if (p + 4 <= end) {
        h = rotl64(h ^ (le32(p) * P1), 23) * P2 + P3;
        p += 4;
    }
    for (; p < end; p++) h = rotl64(h ^ (*p * P5), 11) * P1;
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

static int zs_top_bit(uint32_t v) { /* index of the highest set bit; v != 0 */
    int i = 0;
    while (v >>= 1) i++;
    return i;
}

/* bits read forwards, the lowest bit of a byte first (the FSE table descriptions) */
typedef struct {
    const uint8_t *p;
    size_t n, bit;
    int bad;
} zs_fwd;

static uint32_t zs_fwd_bits(zs_fwd *r, int nb) {
    uint32_t v = 0;
    for (int i = 0; i < nb; i++) {
        size_t at = r->bit + (size_t)i;
        if ((at >> 3) >= r->n) {
            r->bad = 1;
            break;
        }
        v |= (uint32_t)((r->p[at >> 3] >> (at & 7)) & 1) << i;
    }
    r->bit += (size_t)nb;
    return v;
}

/* bits read backwards: the stream's last byte holds a marker, its highest set bit, below which the data starts;
 * `bit` counts the bits still unread and goes negative when more are asked for than there are (those read as 0) */
typedef struct {
    const uint8_t *p;
    int64_t bit;
} zs_back;

static int zs_back_init(zs_back *r, const uint8_t *p, size_t n) {
    if (n == 0 || p[n - 1] == 0) return -1;
    r->p = p;
    r->bit = (int64_t)(n - 1) * 8 + zs_top_bit(p[n - 1]);
    return 0;
}

static uint64_t zs_back_bits(zs_back *r, int nb) {
    uint64_t v = 0;
    r->bit -= nb;
    for (int i = nb - 1; i >= 0; i--) {
        int64_t at = r->bit + i;
        v = (v << 1) | (uint64_t)(at >= 0 ? (r->p[at >> 3] >> (at & 7)) & 1 : 0);
    }
    return v;
}

/* an FSE decoding table: the symbol a state stands for, and the state that follows it: base + that many bits */
typedef struct {
    uint8_t sym[512], nbits[512];
    uint16_t base[512];
    int log; /* -1: none yet (a block that asks to repeat it is damaged) */
} zs_fse;

static int zs_fse_build(zs_fse *t, const int16_t *freq, int n_sym, int log) {
    int size = 1 << log, high = size;
    uint16_t next[256];
    for (int s = 0; s < n_sym; s++)
        if (freq[s] == -1) {
            t->sym[--high] = (uint8_t)s;
            next[s] = 1;
        }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < n_sym; s++) {
        if (freq[s] <= 0) continue;
        next[s] = (uint16_t)freq[s];
        for (int i = 0; i < freq[s]; i++) {
            t->sym[pos] = (uint8_t)s;
            do pos = (pos + step) & mask;
            while (pos >= high);
        }
    }
    if (pos != 0) return -1;
    for (int i = 0; i < size; i++) {
        uint16_t x = next[t->sym[i]]++;
        t->nbits[i] = (uint8_t)(log - zs_top_bit(x));
        t->base[i] = (uint16_t)((x << t->nbits[i]) - size);
    }
    t->log = log;
    return 0;
}

/* a table description (RFC 8878, 4.1.1): the accuracy log, then a count a symbol in as many bits as what is
 * left of the table can need, a run of zero counts by two-bit repeat flags; ends on a byte boundary */
static int zs_fse_describe(const uint8_t *p, size_t n, int max_log, int max_sym, zs_fse *t, size_t *used) {
    zs_fwd r = {p, n, 0, 0};
    int log = 5 + (int)zs_fwd_bits(&r, 4);
    if (log > max_log) return -1;
    int left = 1 << log, n_sym = 0;
    int16_t freq[256];
    while (left > 0 && n_sym < max_sym) {
        int bits = zs_top_bit((uint32_t)left + 1) + 1;
        uint32_t v = zs_fwd_bits(&r, bits);
        uint32_t low = (1u << (bits - 1)) - 1, few = (1u << bits) - 1 - ((uint32_t)left + 1);
        if ((v & low) < few) {
            r.bit -= 1;
            v &= low;
        } else if (v > low) {
            v -= few;
        }
        int count = (int)v - 1;
        left -= count < 0 ? 1 : count;
        freq[n_sym++] = (int16_t)count;
        if (count == 0) {
            for (;;) {
                uint32_t more = zs_fwd_bits(&r, 2);
                for (uint32_t i = 0; i < more && n_sym < max_sym; i++) freq[n_sym++] = 0;
                if (more != 3 || r.bad) break;
            }
        }
        if (r.bad) return -1;
    }
    if (left != 0 || r.bad) return -1;
    *used = (r.bit + 7) >> 3;
    if (*used > n) return -1;
    return zs_fse_build(t, freq, n_sym, log);
}

/* the literals' Huffman code as a table over the next max_bits bits */
typedef struct {
    uint8_t sym[2048], nbits[2048];
    int max_bits; /* 0: none yet */
} zs_huf;

static int zs_huf_build(zs_huf *h, uint8_t *weight, int n) { /* n weights; the last symbol's follows from them */
    uint32_t sum = 0;
    for (int i = 0; i < n; i++) {
        if (weight[i] > 11) return -1;
        sum += weight[i] ? 1u << (weight[i] - 1) : 0;
    }
    if (sum == 0 || n >= 256) return -1;
    int max_bits = zs_top_bit(sum) + 1;
    uint32_t left = (1u << max_bits) - sum;
    if (max_bits > 11 || (left & (left - 1))) return -1;
    weight[n++] = (uint8_t)(zs_top_bit(left) + 1);
    uint32_t start[13] = {0}, count[13] = {0};
    for (int i = 0; i < n; i++) count[weight[i]]++;
    for (int w = 1, at = 0; w <= max_bits; w++) {
        start[w] = (uint32_t)at;
        at += (int)(count[w] << (w - 1));
    }
    for (int s = 0; s < n; s++) {
        int w = weight[s];
        if (!w) continue;
        for (uint32_t i = 0; i < (1u << (w - 1)); i++) {
            h->sym[start[w] + i] = (uint8_t)s;
            h->nbits[start[w] + i] = (uint8_t)(max_bits + 1 - w);
        }
        start[w] += 1u << (w - 1);
    }
    h->max_bits = max_bits;
    return 0;
}

/* the tree description (4.2.1): weights as nibbles, or FSE coded with two states taking turns */
static int zs_huf_describe(const uint8_t *p, size_t n, zs_huf *h, size_t *used) {
    uint8_t weight[256];
    int count = 0;
    if (n < 1) return -1;
    int head = p[0];
    if (head >= 128) {
        count = head - 127;
        size_t bytes = ((size_t)count + 1) / 2;
        if (n < 1 + bytes) return -1;
        for (int i = 0; i < count; i++) weight[i] = i & 1 ? p[1 + i / 2] & 15 : p[1 + i / 2] >> 4;
        *used = 1 + bytes;
    } else {
        if (head == 0 || n < 1 + (size_t)head) return -1;
        zs_fse t;
        size_t hdr = 0;
        if (zs_fse_describe(p + 1, (size_t)head, 6, 256, &t, &hdr)) return -1;
        zs_back r;
        if (zs_back_init(&r, p + 1 + hdr, (size_t)head - hdr)) return -1;
        uint32_t s1 = (uint32_t)zs_back_bits(&r, t.log), s2 = (uint32_t)zs_back_bits(&r, t.log);
        if (r.bit < 0) return -1;
        for (;;) {
            if (count > 253) return -1;
            weight[count++] = t.sym[s1];
            s1 = t.base[s1] + (uint32_t)zs_back_bits(&r, t.nbits[s1]);
            if (r.bit < 0) {
                weight[count++] = t.sym[s2];
                break;
            }
            weight[count++] = t.sym[s2];
            s2 = t.base[s2] + (uint32_t)zs_back_bits(&r, t.nbits[s2]);
            if (r.bit < 0) {
                weight[count++] = t.sym[s1];
                break;
            }
        }
        *used = 1 + (size_t)head;
    }
    return zs_huf_build(h, weight, count);
}

static int zs_huf_stream(const zs_huf *h, const uint8_t *p, size_t n, uint8_t *out, size_t want) {
    zs_back r;
    if (zs_back_init(&r, p, n)) return -1;
    uint32_t mask = (1u << h->max_bits) - 1, state = (uint32_t)zs_back_bits(&r, h->max_bits);
    size_t got = 0;
    while (r.bit > -(int64_t)h->max_bits) {
        if (got == want) return -1;
        out[got++] = h->sym[state];
        int nb = h->nbits[state];
        state = ((state << nb) + (uint32_t)zs_back_bits(&r, nb)) & mask;
    }
    return r.bit == -(int64_t)h->max_bits && got == want ? 0 : -1;
}

typedef struct {
    zs_huf huf;
    zs_fse ll, of, ml;
    uint32_t rep[3];
    uint8_t *lit; /* 128 KiB */
} zs_frame;

static const int16_t zs_ll_default[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                          2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t zs_of_default[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const int16_t zs_ml_default[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                          1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const uint32_t zs_ll_base[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,   13,   14,   15,   16,    18,
                                        20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t zs_ll_extra[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t zs_ml_base[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                        21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                        43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t zs_ml_extra[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                        0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* one of the three code tables of a sequences section, by its mode (4.1.1 of the sequences header) */
static int zs_seq_table(int mode, const uint8_t *p, size_t n, size_t *pos, zs_fse *t, const int16_t *dflt, int dflt_n,
                        int dflt_log, int max_log, int max_sym) {
    if (mode == 0) return zs_fse_build(t, dflt, dflt_n, dflt_log);
    if (mode == 1) {
        if (*pos >= n || p[*pos] >= max_sym) return -1;
        t->sym[0] = p[(*pos)++];
        t->nbits[0] = 0;
        t->base[0] = 0;
        t->log = 0;
        return 0;
    }
    if (mode == 2) {
        size_t used = 0;
        if (zs_fse_describe(p + *pos, n - *pos, max_log, max_sym, t, &used)) return -1;
        *pos += used;
        return 0;
    }
    return t->log < 0 ? -1 : 0; /* repeat: the table of the block before */
}

/* a compressed block (3.1.1.2 - 3.1.1.4): `out` is the frame's output so far, d bytes; block_max bounds what it adds */
static int zs_block(zs_frame *f, const uint8_t *p, size_t n, uint8_t *out, size_t *d_io, size_t cap, size_t block_max) {
    if (n < 1) return -1;
    /* ---- literals */
    int kind = p[0] & 3, fmt = (p[0] >> 2) & 3;
    size_t regen, comp = 0, pos, lit_n;
    const uint8_t *lit;
    int streams = 1;
    if (kind < 2) {
        if (fmt == 0 || fmt == 2) regen = p[0] >> 3, pos = 1;
        else if (fmt == 1) {
            if (n < 2) return -1;
            regen = (size_t)((p[0] | p[1] << 8) >> 4), pos = 2;
        } else {
            if (n < 3) return -1;
            regen = (size_t)((p[0] | p[1] << 8 | p[2] << 16) >> 4), pos = 3;
        }
    } else {
        if (n < 5) return -1;
        uint64_t h = (uint64_t)p[0] | (uint64_t)p[1] << 8 | (uint64_t)p[2] << 16 | (uint64_t)p[3] << 24 | (uint64_t)p[4] << 32;
        if (fmt == 0 || fmt == 1) regen = (h >> 4) & 0x3ff, comp = (h >> 14) & 0x3ff, pos = 3;
        else if (fmt == 2) regen = (h >> 4) & 0x3fff, comp = (h >> 18) & 0x3fff, pos = 4;
        else regen = (h >> 4) & 0x3ffff, comp = (h >> 22) & 0x3ffff, pos = 5;
        streams = fmt == 0 ? 1 : 4;
    }
    if (regen > block_max) return -1;
    lit_n = regen;
    if (kind == 0) {
        if (n - pos < regen) return -1;
        lit = p + pos;
        pos += regen;
    } else if (kind == 1) {
        if (n - pos < 1) return -1;
        memset(f->lit, p[pos], regen);
        lit = f->lit;
        pos += 1;
    } else {
        if (n - pos < comp) return -1;
        const uint8_t *q = p + pos;
        size_t qn = comp;
        if (kind == 2) {
            size_t used = 0;
            if (zs_huf_describe(q, qn, &f->huf, &used)) return -1;
            q += used, qn -= used;
        } else if (!f->huf.max_bits) {
            return -1; /* the tree of the block before: there is none */
        }
        if (streams == 1) {
            if (zs_huf_stream(&f->huf, q, qn, f->lit, regen)) return -1;
        } else {
            if (qn < 6) return -1;
            size_t s1 = (size_t)(q[0] | q[1] << 8), s2 = (size_t)(q[2] | q[3] << 8), s3 = (size_t)(q[4] | q[5] << 8);
            if (s1 + s2 + s3 > qn - 6) return -1;
            size_t s4 = qn - 6 - s1 - s2 - s3, each = (regen + 3) / 4;
            if (3 * each > regen) return -1;
            const uint8_t *a = q + 6;
            if (zs_huf_stream(&f->huf, a, s1, f->lit, each) || zs_huf_stream(&f->huf, a + s1, s2, f->lit + each, each) ||
                zs_huf_stream(&f->huf, a + s1 + s2, s3, f->lit + 2 * each, each) ||
                zs_huf_stream(&f->huf, a + s1 + s2 + s3, s4, f->lit + 3 * each, regen - 3 * each))
                return -1;
        }
        lit = f->lit;
        pos += comp;
    }
    /* ---- sequences */
    if (pos >= n) return -1;
    size_t n_seq = p[pos++];
    if (n_seq >= 128) {
        if (n_seq == 255) {
            if (n - pos < 2) return -1;
            n_seq = (size_t)p[pos] + ((size_t)p[pos + 1] << 8) + 0x7f00;
            pos += 2;
        } else {
            if (n - pos < 1) return -1;
            n_seq = ((n_seq - 128) << 8) + p[pos++];
        }
    }
    size_t d = *d_io, d0 = d, lit_at = 0;
    if (n_seq) {
        if (pos >= n) return -1;
        int modes = p[pos++];
        if (modes & 3) return -1;
        if (zs_seq_table(modes >> 6, p, n, &pos, &f->ll, zs_ll_default, 36, 6, 9, 36) ||
            zs_seq_table((modes >> 4) & 3, p, n, &pos, &f->of, zs_of_default, 29, 5, 8, 32) ||
            zs_seq_table((modes >> 2) & 3, p, n, &pos, &f->ml, zs_ml_default, 53, 6, 9, 53))
            return -1;
        zs_back r;
        if (zs_back_init(&r, p + pos, n - pos)) return -1;
        uint32_t sl = (uint32_t)zs_back_bits(&r, f->ll.log), so = (uint32_t)zs_back_bits(&r, f->of.log),
                 sm = (uint32_t)zs_back_bits(&r, f->ml.log);
        for (size_t i = 0; i < n_seq; i++) {
            int oc = f->of.sym[so], mc = f->ml.sym[sm], lc = f->ll.sym[sl];
            if (oc > 31 || mc > 52 || lc > 35) return -1;
            uint64_t ov = ((uint64_t)1 << oc) + zs_back_bits(&r, oc);
            size_t ml = zs_ml_base[mc] + (size_t)zs_back_bits(&r, zs_ml_extra[mc]);
            size_t ll = zs_ll_base[lc] + (size_t)zs_back_bits(&r, zs_ll_extra[lc]);
            uint64_t off;
            if (ov > 3) {
                off = ov - 3;
                f->rep[2] = f->rep[1], f->rep[1] = f->rep[0], f->rep[0] = (uint32_t)off;
            } else {
                uint32_t idx = (uint32_t)ov - 1 + (ll == 0 ? 1u : 0u); /* 0, 1, 2, or 3: the first one less one */
                if (idx == 0) {
                    off = f->rep[0];
                } else {
                    off = idx < 3 ? f->rep[idx] : (uint64_t)f->rep[0] - 1;
                    if (idx > 1) f->rep[2] = f->rep[1];
                    f->rep[1] = f->rep[0];
                    f->rep[0] = (uint32_t)off;
                }
            }
            if (off == 0 || off > 0xffffffffull) return -1;
            if (i + 1 < n_seq) {
                sl = f->ll.base[sl] + (uint32_t)zs_back_bits(&r, f->ll.nbits[sl]);
                sm = f->ml.base[sm] + (uint32_t)zs_back_bits(&r, f->ml.nbits[sm]);
                so = f->of.base[so] + (uint32_t)zs_back_bits(&r, f->of.nbits[so]);
            }
            if (r.bit < 0) return -1;
            if (ll > lit_n - lit_at || ll + ml > block_max - (d - d0) || ll + ml > cap - d) return -1;
            memcpy(out + d, lit + lit_at, ll);
            d += ll, lit_at += ll;
            if (off > d) return -1; /* (no dictionary: nothing lies before the frame's first byte) */
            for (size_t k = 0; k < ml; k++) out[d + k] = out[d + k - off];
            d += ml;
        }
        if (r.bit != 0) return -1;
    } else if (pos != n) {
        return -1;
    }
    size_t rest = lit_n - lit_at;
    if (rest > block_max - (d - d0) || rest > cap - d) return -1;
    memcpy(out + d, lit + lit_at, rest);
    *d_io = d + rest;
    return 0;
}

/* a frame's header (3.1.1.1); returns its bytes, 0 when it is not one this decoder takes */
static size_t zs_header(const uint8_t *in, size_t n, uint64_t *window, int *has_sum, int *has_size, uint64_t *content) {
    if (n < 6 || le32(in) != 0xFD2FB528u) return 0;
    int fhd = in[4], single = (fhd >> 5) & 1, did = fhd & 3, fcs = fhd >> 6;
    if (fhd & 0x08) return 0; /* the reserved bit */
    size_t s = 5;
    *window = 0;
    if (!single) {
        int wd = in[s++];
        uint64_t base = (uint64_t)1 << (10 + (wd >> 3));
        *window = base + (base >> 3) * (uint64_t)(wd & 7);
    }
    static const int did_bytes[4] = {0, 1, 2, 4};
    if (n < s + (size_t)did_bytes[did]) return 0;
    uint32_t dict = 0;
    for (int i = 0; i < did_bytes[did]; i++) dict |= (uint32_t)in[s + (size_t)i] << (8 * i);
    s += (size_t)did_bytes[did];
    if (dict) return 0; /* a dictionary this decoder does not have */
    int fcs_bytes = fcs == 0 ? single : 1 << fcs;
    if (n < s + (size_t)fcs_bytes) return 0;
    *has_size = fcs_bytes != 0;
    *content = 0;
    for (int i = 0; i < fcs_bytes; i++) *content |= (uint64_t)in[s + (size_t)i] << (8 * i);
    if (fcs_bytes == 2) *content += 256;
    s += (size_t)fcs_bytes;
    if (single) *window = *content;
    *has_sum = (fhd >> 2) & 1;
    if (*window > ((uint64_t)1 << 27)) return 0; /* the stream decoder's default limit on the window (ZSTD_WINDOWLOG_LIMIT_DEFAULT) */
    return s;
}

/* Room that holds what the frames of the record data can regenerate: every raw or repeated block at its size, every
 * compressed block at the maximal block size.  0: the headers do not chain up to the end of the data. */
size_t no_zstd_room(const uint8_t *in, size_t n) {
    size_t s = 0, room = 0;
    while (s < n) {
        if (n - s >= 8 && (le32(in + s) & 0xFFFFFFF0u) == 0x184D2A50u) { /* a skippable frame */
            uint64_t skip = le32(in + s + 4);
            if (skip > n - s - 8) return 0;
            s += 8 + (size_t)skip;
            continue;
        }
        uint64_t window, content;
        int has_sum, has_size;
        size_t h = zs_header(in + s, n - s, &window, &has_sum, &has_size, &content);
        if (!h) return 0;
        s += h;
        size_t block_max = window < (128u << 10) ? (size_t)window : (128u << 10);
        for (;;) {
            if (n - s < 3) return 0;
            uint32_t bh = (uint32_t)in[s] | (uint32_t)in[s + 1] << 8 | (uint32_t)in[s + 2] << 16;
            s += 3;
            uint32_t type = (bh >> 1) & 3, size = bh >> 3;
            if (type == 3 || size > block_max) return 0;
            size_t body = type == 1 ? 1 : size;
            if (n - s < body) return 0;
            s += body;
            room += type == 2 ? block_max : size;
            if (bh & 1) break;
        }
        if (has_sum) {
            if (n - s < 4) return 0;
            s += 4;
        }
    }
    return room + 1;
}

/* cap: at least no_zstd_room */
int no_zstd_frames(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    *out_len = 0;
    size_t s = 0, d = 0;
    zs_frame *f = (zs_frame *)malloc(sizeof *f);
    uint8_t *lit = (uint8_t *)malloc(128u << 10);
    int st = NO_OK;
    while (s < n && st == NO_OK) {
        if (n - s >= 8 && (le32(in + s) & 0xFFFFFFF0u) == 0x184D2A50u) {
            uint64_t skip = le32(in + s + 4);
            if (skip > n - s - 8) st = NO_INFLATE;
            s += 8 + (size_t)skip;
            continue;
        }
        uint64_t window, content;
        int has_sum, has_size;
        size_t h = zs_header(in + s, n - s, &window, &has_sum, &has_size, &content);
        if (!h) {
            st = NO_INFLATE;
            break;
        }
        s += h;
        size_t block_max = window < (128u << 10) ? (size_t)window : (128u << 10);
        memset(f, 0, sizeof *f);
        f->lit = lit;
        f->ll.log = f->of.log = f->ml.log = -1;
        f->rep[0] = 1, f->rep[1] = 4, f->rep[2] = 8;
        size_t first = d; /* a frame stands alone: its matches reach back to its own first byte at most */
        for (;;) {
            if (n - s < 3) {
                st = NO_INFLATE;
                break;
            }
            uint32_t bh = (uint32_t)in[s] | (uint32_t)in[s + 1] << 8 | (uint32_t)in[s + 2] << 16;
            s += 3;
            uint32_t type = (bh >> 1) & 3, size = bh >> 3;
            if (type == 3 || size > block_max || n - s < (type == 1 ? 1u : size)) {
                st = NO_INFLATE;
                break;
            }
            if (type == 0 || type == 1) {
                if (size > cap - d) {
                    st = NO_CAPACITY;
                    break;
                }
                if (type == 0) memcpy(out + d, in + s, size);
                else memset(out + d, in[s], size);
                d += size;
                s += type == 0 ? size : 1;
            } else {
                size_t fd = d - first;
                if (cap - d < block_max) {
                    st = NO_CAPACITY;
                    break;
                }
                if (zs_block(f, in + s, size, out + first, &fd, cap - first, block_max)) {
                    st = NO_INFLATE;
                    break;
                }
                d = first + fd;
                s += size;
            }
            if (bh & 1) break;
        }
        if (st != NO_OK) break;
        if (has_sum) {
            if (n - s < 4 || (uint32_t)no_xxh64(out + first, d - first, 0) != le32(in + s)) {
                st = NO_INFLATE;
                break;
            }
            s += 4;
        }
        if (has_size && content != d - first) st = NO_INFLATE;
    }
    free(lit);
    free(f);
    if (st == NO_OK) *out_len = d;
    return st;
}

int no_inflate_frame(const uint8_t *blob, const uint64_t *batch_off, size_t n_batches, uint8_t *out,
                     size_t out_cap, uint64_t *out_off, int32_t *status) {
    size_t at = 0;
    for (size_t b = 0; b < n_batches; b++) {
        const uint8_t *p = blob + batch_off[b];
        size_t len = (size_t)(batch_off[b + 1] - batch_off[b]);
        out_off[b] = at;
        status[b] = NO_NONE_DONE;
        no_deflated_batch h;
        int st = no_deflated_parse(p, len, &h);
        int codec = 0;
        if (st == NO_OK && no_compression(h.attributes, &codec) == NO_OK && codec >= 1 && codec <= 4) {
            size_t hl = 0, got = 0;
            uint64_t want = 0;
            const uint8_t *data = h.record_data;
            size_t n = h.record_data_len;
            st = NO_INFLATE;
            if (codec == 1) {
                /* gzip: the member's last four bytes say how long it is */
                if (n >= 18 && 49 + (uint64_t)le32(data + n - 4) <= 0x7fffffffull && (uint64_t)le32(data + n - 4) <= 1032ull * n + 64) {
                    want = le32(data + n - 4);
                    if (at + 61 + want > out_cap) return NO_CAPACITY;
                    st = no_gzip_member(data, n, out + at + 61, (size_t)want, &got);
                }
            } else if (codec == 4) {
                /* zstd: into room of its own first, as LZ4 (a frame need not say how long it is) */
                size_t room = no_zstd_room(data, n);
                uint8_t *tmp = (uint8_t *)malloc(room ? room : 1);
                st = room ? no_zstd_frames(data, n, tmp, room, &got) : NO_INFLATE;
                if (st == NO_OK && 49 + (uint64_t)got > 0x7fffffffull) st = NO_INFLATE;
                if (st == NO_OK) {
                    if (at + 61 + got > out_cap) {
                        free(tmp);
                        return NO_CAPACITY;
                    }
                    memcpy(out + at + 61, tmp, got);
                }
                free(tmp);
            } else if (codec == 3) {
                /* LZ4: into room of its own first (a frame does not say how long it is) */
                size_t room = lz4_frame_room(data, n);
                uint8_t *tmp = (uint8_t *)malloc(room ? room : 1);
                st = no_lz4_frame(data, n, tmp, room, &got);
                if (st == NO_OK && 49 + (uint64_t)got > 0x7fffffffull) st = NO_INFLATE;
                if (st == NO_OK) {
                    if (at + 61 + got > out_cap) {
                        free(tmp);
                        return NO_CAPACITY;
                    }
                    memcpy(out + at + 61, tmp, got);
                }
                free(tmp);
            } else {
                /* Snappy: the announced length sizes the output before anything is decoded */
                static const uint8_t magic[8] = {0x82, 'S', 'N', 'A', 'P', 'P', 'Y', 0};
                const uint8_t *raw = data;
                size_t rn = n;
                int ok = 1;
                if (n >= 8 && memcmp(data, magic, 8) == 0) {
                    if (n < 20) ok = 0;
                    else raw += 20, rn -= 20;
                }
                if (ok && rn && no_snappy_raw_len(raw, rn, &hl, &want) == NO_OK && 49 + want <= 0x7fffffffull) {
                    if (at + 61 + want > out_cap) return NO_CAPACITY;
                    st = no_snappy_inflate(data, n, out + at + 61, (size_t)want, &got);
                }
            }
            if (st == NO_OK) {
                memcpy(out + at, p, 61);
                uint32_t bl = 49 + (uint32_t)got;
                out[at + 8] = (uint8_t)(bl >> 24), out[at + 9] = (uint8_t)(bl >> 16);
                out[at + 10] = (uint8_t)(bl >> 8), out[at + 11] = (uint8_t)bl;
                out[at + 22] &= (uint8_t)~7u; /* attributes: big-endian i16 at 21 */
                at += 61 + got;
                status[b] = NO_OK;
                continue;
            }
            status[b] = NO_INFLATE;
        }
        if (at + len > out_cap) return NO_CAPACITY;
        memcpy(out + at, p, len);
        at += len;
    }
    out_off[n_batches] = at;
    return NO_OK;
}
