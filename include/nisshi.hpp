// nisshi.hpp: the record path of nisshi-sans-io as C++ types over the C ABI of
// nisshi_hip.h.  Names, fields, argument meaning and error behaviour follow the
// reference (paths in the tansu / nisshi reference tree):
//
//   record::Header                       src/record/header.rs:21-29
//   record::Record                       src/record.rs:172-198
//   record::deflated::{Batch, Frame}     src/record/deflated.rs:28-106
//   record::inflated::{Batch, Frame}     src/record/inflated.rs:29-82
//   Error                                src/lib.rs:328-366
//   storage::combine                     nisshi-storage/src/batch.rs:661-709
//   storage::produce_rows                nisshi-storage/src/pg.rs:797-943, lite.rs:703-815
//
// Conversions the reference writes as TryFrom/From are static `try_from` /
// `from` functions here.  The ones that decode or encode records take the Codec
// (the device context) as their first argument and run on the GPU, a whole frame
// in one call; deflated::Batch::try_from(Bytes) only splits the 61-byte header
// off, as in the reference, and needs no device.  Header only: link with
// libnisshi_hip.so.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "nisshi_hip.h"

namespace nisshi {

// A cheaply cloneable slice of shared bytes, the role bytes::Bytes has in the reference.
class Bytes {
  public:
    Bytes() : off_(0), len_(0) {}
    explicit Bytes(std::vector<uint8_t> v)
        : buf_(std::make_shared<const std::vector<uint8_t>>(std::move(v))), off_(0), len_(buf_->size()) {}
    Bytes(const uint8_t *p, size_t n) : Bytes(std::vector<uint8_t>(p, p + n)) {}
    const uint8_t *data() const { return buf_ ? buf_->data() + off_ : nullptr; }
    size_t len() const { return len_; }
    bool is_empty() const { return len_ == 0; }
    uint8_t operator[](size_t i) const { return data()[i]; }
    // Bytes::slice(off..off + n): shares the buffer
    Bytes slice(size_t off, size_t n) const {
        if (off > len_ || n > len_ - off) throw std::out_of_range("Bytes::slice");
        Bytes b;
        b.buf_ = buf_;
        b.off_ = off_ + off;
        b.len_ = n;
        return b;
    }
    bool operator==(const Bytes &o) const { return len_ == o.len_ && (len_ == 0 || !memcmp(data(), o.data(), len_)); }
    bool operator!=(const Bytes &o) const { return !(*this == o); }

  private:
    std::shared_ptr<const std::vector<uint8_t>> buf_;
    size_t off_, len_;
};

// nisshi_sans_io::Error, the variants this path can produce (src/lib.rs:328-366),
// and what the device library adds.
struct Error {
    enum Kind {
        TryGet,                  // buffer under-run
        Overflow,
        TryFromInt,              // negative batch_length
        UnknownCompressionType,  // detail = attributes & 7
        Malformed,               // the reference panics on this input (codec.rs:70-74, 291-295; record.rs:260)
        Compressed,              // a batch that is still compressed: the decode entry points take uncompressed batches (inflate first)
        Capacity,
        Device,
        InvalidArgument,
        Inflate,                 // gzip, Snappy, LZ4 or zstd record data that does not inflate (the inflator's Error::Snap / Error::Io)
        InflatedRecords,         // record data that inflated but whose records do not decode: the reference reads
                                 // them through serde over the inflator's reader (deflated.rs:444-457) and returns
                                 // Err (Error::Io or Error::Message) -- it never panics there; detail = the decode's status
    } kind;
    int detail;
    std::string message;

    static Error from_status(int status, int detail = 0) {
        static const Kind kinds[] = {TryGet, TryGet, Overflow, TryFromInt, UnknownCompressionType,
                                     Malformed, Compressed, Capacity, Device, InvalidArgument};
        if (status == NISSHI_INFLATE) return Error{Inflate, detail, nisshi_hip_status_str(status)};
        return Error{kinds[status >= 1 && status <= 9 ? status : 8], detail, nisshi_hip_status_str(status)};
    }
};

// Result<T, Error>
template <class T>
class Result {
  public:
    Result(T v) : v_(std::move(v)) {}
    Result(Error e) : v_(std::move(e)) {}
    bool is_ok() const { return v_.index() == 0; }
    bool is_err() const { return !is_ok(); }
    explicit operator bool() const { return is_ok(); }
    T &operator*() { return std::get<0>(v_); }
    const T &operator*() const { return std::get<0>(v_); }
    T *operator->() { return &std::get<0>(v_); }
    const T *operator->() const { return &std::get<0>(v_); }
    const Error &error() const { return std::get<1>(v_); }
    // like Result::unwrap: throws on an error
    T unwrap() && {
        if (is_err()) throw std::runtime_error("unwrap on Err: " + error().message);
        return std::move(std::get<0>(v_));
    }

  private:
    std::variant<T, Error> v_;
};

// The device context.  One per thread (a context runs one call at a time).
class Codec {
  public:
    explicit Codec(int device = 0) {
        if (nisshi_hip_abi_version() != NISSHI_HIP_ABI_VERSION)
            throw std::runtime_error("libnisshi_hip was built from another nisshi_hip.h than this program");
        int st = nisshi_hip_ctx_create(device, &ctx_);
        if (st) throw std::runtime_error(std::string("nisshi_hip_ctx_create: ") + nisshi_hip_status_str(st));
    }
    ~Codec() { nisshi_hip_ctx_destroy(ctx_); }
    Codec(const Codec &) = delete;
    Codec &operator=(const Codec &) = delete;
    nisshi_hip_ctx *ctx() const { return ctx_; }

  private:
    nisshi_hip_ctx *ctx_ = nullptr;
};

namespace record {

// header.rs:21-29
struct Header {
    std::optional<Bytes> key, value;
    bool operator==(const Header &o) const { return key == o.key && value == o.value; }
};

// record.rs:172-198
struct Record {
    int32_t length = 0;
    uint8_t attributes = 0;
    int64_t timestamp_delta = 0;
    int32_t offset_delta = 0;
    std::optional<Bytes> key, value;
    std::vector<Header> headers;
    bool operator==(const Record &o) const {
        return length == o.length && attributes == o.attributes && timestamp_delta == o.timestamp_delta &&
               offset_delta == o.offset_delta && key == o.key && value == o.value && headers == o.headers;
    }
};

// ByteSize of the primitives (primitive/varint.rs:186-196, 379-389, 546-556;
// record/codec.rs:34-45): the bytes of a zigzag varint of 32 and of 64 bits, of an
// unsigned varint, and of a nullable span behind its length.
inline size_t varint_size(int32_t v) {
    uint32_t z = (uint32_t(v) << 1) ^ uint32_t(v >> 31);
    size_t n = 1;
    while (z >= 0x80) z >>= 7, n++;
    return n;
}
inline size_t long_varint_size(int64_t v) {
    uint64_t z = (uint64_t(v) << 1) ^ uint64_t(v >> 63);
    size_t n = 1;
    while (z >= 0x80) z >>= 7, n++;
    return n;
}
inline size_t unsigned_varint_size(uint32_t v) {
    size_t n = 1;
    while (v >= 0x80) v >>= 7, n++;
    return n;
}
inline size_t octets_size(const std::optional<Bytes> &b) {
    return b ? varint_size(int32_t(b->len())) + b->len() : varint_size(-1);
}
// Header::size_in_bytes, header.rs:31-41
inline size_t header_size(const Header &h) { return octets_size(h.key) + octets_size(h.value); }

// record::Builder, record.rs:329-421.  build() is TryFrom<Builder> for Record
// (record.rs:298-314): `length` becomes the size of the record's body, which
// excludes the length prefix itself.
struct Builder {
    uint8_t attributes_ = 0;
    int64_t timestamp_delta_ = 0;
    int32_t offset_delta_ = 0;
    std::optional<Bytes> key_, value_;
    std::vector<Header> headers_;

    Builder attributes(uint8_t v) && { attributes_ = v; return std::move(*this); }
    Builder timestamp_delta(int64_t v) && { timestamp_delta_ = v; return std::move(*this); }
    Builder offset_delta(int32_t v) && { offset_delta_ = v; return std::move(*this); }
    Builder key(std::optional<Bytes> v) && { key_ = std::move(v); return std::move(*this); }
    Builder value(std::optional<Bytes> v) && { value_ = std::move(v); return std::move(*this); }
    Builder header(Header h) && { headers_.push_back(std::move(h)); return std::move(*this); }

    // ByteSize for Builder, record.rs:363-372
    size_t size_in_bytes() const {
        size_t n = 1 + long_varint_size(timestamp_delta_) + varint_size(offset_delta_) + octets_size(key_) +
                   octets_size(value_) + varint_size(int32_t(headers_.size()));
        for (const Header &h : headers_) n += header_size(h);
        return n;
    }
    Result<Record> build() const {
        size_t n = size_in_bytes();
        if (n > 0x7fffffffu) return Error::from_status(NISSHI_TRY_FROM_INT);
        Record r;
        r.length = int32_t(n);
        r.attributes = attributes_;
        r.timestamp_delta = timestamp_delta_;
        r.offset_delta = offset_delta_;
        r.key = key_;
        r.value = value_;
        r.headers = headers_;
        return r;
    }
    // From<Record> for Builder, record.rs:316-327: the stored length is dropped
    static Builder from(const Record &r) {
        Builder b;
        b.attributes_ = r.attributes;
        b.timestamp_delta_ = r.timestamp_delta;
        b.offset_delta_ = r.offset_delta;
        b.key_ = r.key;
        b.value_ = r.value;
        b.headers_ = r.headers;
        return b;
    }
};

// Sequence<T>, codec.rs:386-522: a u32 count, then the items.  Its ByteSize
// (codec.rs:482-491) sizes every item as UnsignedVarInt::size_inclusive
// (varint.rs:536-543): the item's own size behind an UNSIGNED varint of that
// size.  For record builders the prefix actually written is a zigzag varint
// (record.rs:345-351), one byte longer where the size s has 64 <= s < 128,
// 8192 <= s < 16384, 2^20 <= s < 2^21 or 2^27 <= s < 2^28: size_in_bytes is
// short of the serialised length by one for every such item, as in the reference.
template <class T>
struct Sequence {
    std::vector<T> items;
    Sequence() = default;
    Sequence(std::vector<T> v) : items(std::move(v)) {}
    void extend(std::vector<T> more) {
        for (T &t : more) items.push_back(std::move(t));
    }
    size_t size_in_bytes() const {
        size_t n = sizeof(uint32_t);
        for (const T &t : items) {
            size_t sz = t.size_in_bytes();
            n += unsigned_varint_size(uint32_t(sz)) + sz;
        }
        return n;
    }
};

namespace inflated {
struct Batch;
struct Frame;
struct Builder;
}  // namespace inflated

namespace deflated {

// deflated.rs:67-82
struct Batch {
    int64_t base_offset = 0;
    int32_t batch_length = 0;
    int32_t partition_leader_epoch = 0;
    int8_t magic = 0;
    uint32_t crc = 0;
    int16_t attributes = 0;
    int32_t last_offset_delta = 0;
    int64_t base_timestamp = 0;
    int64_t max_timestamp = 0;
    int64_t producer_id = 0;
    int16_t producer_epoch = 0;
    int32_t base_sequence = 0;
    uint32_t record_count = 0;
    Bytes record_data;

    // TryFrom<Bytes> for Batch, deflated.rs:108-213, in its order of checks.  A
    // CRC that does not match is only logged there; it is not checked here.
    static Result<Batch> try_from(const Bytes &encoded);
    // From<Batch> for Bytes, deflated.rs:84-106
    Bytes into_bytes() const;
    // TryFrom<inflated::Batch> for Batch, deflated.rs:392-409 (uncompressed)
    static Result<Batch> try_from(Codec &codec, const inflated::Batch &batch);
    int64_t max_offset() const { return base_offset + int64_t(last_offset_delta); }  // deflated.rs:412-414
    // deflated.rs:215-230: attributes bit 4, bit 5; a producer id and a sequence that are both set
    bool is_transactional() const { return (attributes & 0x10) == 0x10; }
    bool is_control() const { return (attributes & 0x20) == 0x20; }
    bool is_idempotent() const { return producer_id != -1 && base_sequence != -1; }

    bool operator==(const Batch &o) const {
        return base_offset == o.base_offset && batch_length == o.batch_length &&
               partition_leader_epoch == o.partition_leader_epoch && magic == o.magic && crc == o.crc &&
               attributes == o.attributes && last_offset_delta == o.last_offset_delta &&
               base_timestamp == o.base_timestamp && max_timestamp == o.max_timestamp &&
               producer_id == o.producer_id && producer_epoch == o.producer_epoch &&
               base_sequence == o.base_sequence && record_count == o.record_count &&
               record_data == o.record_data;
    }
};

// deflated.rs:28-63
struct Frame {
    std::vector<Batch> batches;
    // TryFrom<inflated::Frame> for Frame, deflated.rs:52-63: all batches in one device call
    static Result<Frame> try_from(Codec &codec, const inflated::Frame &frame);
    // the batches of a records blob (de.rs:1056-1093)
    static Result<Frame> split(const Bytes &blob);
};

}  // namespace deflated

namespace inflated {

// inflated.rs:66-82
struct Batch {
    int64_t base_offset = 0;
    int32_t batch_length = 0;
    int32_t partition_leader_epoch = -1;
    int8_t magic = 2;
    uint32_t crc = 0;
    int16_t attributes = 0;
    int32_t last_offset_delta = 0;
    int64_t base_timestamp = 0;
    int64_t max_timestamp = 0;
    int64_t producer_id = -1;
    int16_t producer_epoch = -1;
    int32_t base_sequence = -1;
    std::vector<Record> records;

    // TryFrom<deflated::Batch> (by value, inflated.rs:114-150: the compression bits
    // of attributes are cleared) and TryFrom<&deflated::Batch> (inflated.rs:152-187:
    // attributes verbatim)
    static Result<Batch> try_from(Codec &codec, deflated::Batch &&batch);
    static Result<Batch> try_from(Codec &codec, const deflated::Batch &batch);
    // inflated.rs:201-212
    static inflated::Builder builder();
    inflated::Builder into_builder() const;
    int64_t max_offset() const { return base_offset + int64_t(last_offset_delta); }

    bool operator==(const Batch &o) const {
        return base_offset == o.base_offset && batch_length == o.batch_length &&
               partition_leader_epoch == o.partition_leader_epoch && magic == o.magic && crc == o.crc &&
               attributes == o.attributes && last_offset_delta == o.last_offset_delta &&
               base_timestamp == o.base_timestamp && max_timestamp == o.max_timestamp &&
               producer_id == o.producer_id && producer_epoch == o.producer_epoch &&
               base_sequence == o.base_sequence && records == o.records;
    }
};

// inflated::Builder, inflated.rs:296-482.  Defaults as in the reference
// (inflated.rs:311-329): producer_epoch 0, base_sequence 0, partition_leader_epoch
// -1, magic 2, both timestamps = now in milliseconds.  Nothing derives
// last_offset_delta or the records' offset deltas.
struct Builder {
    int64_t base_offset = 0;
    int32_t partition_leader_epoch = -1;
    int8_t magic = 2;
    int16_t attributes = 0;
    int32_t last_offset_delta = 0;
    int64_t base_timestamp = now_ms();
    int64_t max_timestamp = base_timestamp;
    int64_t producer_id = -1;
    int16_t producer_epoch = 0;
    int32_t base_sequence = 0;
    Sequence<record::Builder> records;

    static int64_t now_ms() {
        return std::chrono::duration_cast<std::chrono::milliseconds>(
                   std::chrono::system_clock::now().time_since_epoch()).count();
    }
    Builder with_base_offset(int64_t v) && { base_offset = v; return std::move(*this); }
    Builder with_partition_leader_epoch(int32_t v) && { partition_leader_epoch = v; return std::move(*this); }
    Builder with_magic(int8_t v) && { magic = v; return std::move(*this); }
    Builder with_attributes(int16_t v) && { attributes = v; return std::move(*this); }
    Builder with_last_offset_delta(int32_t v) && { last_offset_delta = v; return std::move(*this); }
    Builder with_base_timestamp(int64_t v) && { base_timestamp = v; return std::move(*this); }
    Builder with_max_timestamp(int64_t v) && { max_timestamp = v; return std::move(*this); }
    Builder with_producer_id(int64_t v) && { producer_id = v; return std::move(*this); }
    Builder with_producer_epoch(int16_t v) && { producer_epoch = v; return std::move(*this); }
    Builder with_base_sequence(int32_t v) && { base_sequence = v; return std::move(*this); }
    Builder record(record::Builder r) && { records.extend({std::move(r)}); return std::move(*this); }

    // ByteSize for Builder, inflated.rs:331-346: the 45 bytes of the header that
    // batch_length counts before record_count, then Sequence::size_in_bytes (which
    // under-counts, see Sequence)
    size_t size_in_bytes() const { return 4 + 1 + 4 + 2 + 4 + 8 + 8 + 8 + 2 + 4 + records.size_in_bytes(); }

    // Builder::build, inflated.rs:448-481: batch_length = size_in_bytes(), crc =
    // CRC-32C over attributes .. records as RecordBatchEncoder writes them
    // (inflated.rs:429-446), which are the bytes the deflated batch's crc covers
    // (deflated.rs:318-326): the records go through the device encoder once.
    Result<Batch> build(Codec &codec) const;
};

// inflated.rs:29-62
struct Frame {
    std::vector<Batch> batches;
    static Result<Frame> try_from(Codec &codec, deflated::Frame &&frame);
    static Result<Frame> try_from(Codec &codec, const deflated::Frame &frame);
};

}  // namespace inflated

// ---------------------------------------------------------------- implementation

namespace detail {

struct Cursor {  // bytes::Buf::try_get_* on a Bytes: big-endian, TryGet on under-run
    const uint8_t *p;
    size_t left;
    template <class T>
    bool get(T &out) {
        if (left < sizeof(T)) return false;
        uint64_t v = 0;
        for (size_t i = 0; i < sizeof(T); i++) v = (v << 8) | p[i];
        out = static_cast<T>(v);
        p += sizeof(T);
        left -= sizeof(T);
        return true;
    }
};

template <class T>
inline void put_be(std::vector<uint8_t> &out, T v) {
    for (int i = int(sizeof(T)) - 1; i >= 0; i--) out.push_back(uint8_t(uint64_t(v) >> (8 * i)));
}

// Host arrays for the columns of one call.
struct HostColumns {
    std::vector<int64_t> base_offset, base_timestamp, max_timestamp, producer_id, timestamp_delta;
    std::vector<int32_t> batch_length, partition_leader_epoch, last_offset_delta, base_sequence, status;
    std::vector<int8_t> magic;
    std::vector<uint32_t> crc, record_count, crc_computed, data_len, rec_base, hdr_base, hdr_count, enc_len;
    std::vector<int16_t> attributes, producer_epoch;
    std::vector<uint64_t> data_off;
    std::vector<nisshi_record_key> rkey;
    std::vector<nisshi_record_value> rval;
    std::vector<uint8_t> rattributes;
    std::vector<nisshi_header_span> hdr;

    HostColumns(size_t nb, size_t nr, size_t nh) {
        nb = nb ? nb : 1;
        nr = nr ? nr : 1;
        nh = nh ? nh : 1;
        base_offset.resize(nb); base_timestamp.resize(nb); max_timestamp.resize(nb); producer_id.resize(nb);
        batch_length.resize(nb); partition_leader_epoch.resize(nb); last_offset_delta.resize(nb);
        base_sequence.resize(nb); status.resize(nb); magic.resize(nb); crc.resize(nb); record_count.resize(nb);
        crc_computed.resize(nb); data_len.resize(nb); rec_base.resize(nb); hdr_base.resize(nb);
        hdr_count.resize(nb); enc_len.resize(nb); attributes.resize(nb); producer_epoch.resize(nb); data_off.resize(nb);
        rkey.resize(nr); rval.resize(nr); timestamp_delta.resize(nr); rattributes.resize(nr);
        hdr.resize(nh);
    }
    nisshi_columns c() {
        nisshi_columns c;
        c.base_offset = base_offset.data(); c.batch_length = batch_length.data();
        c.partition_leader_epoch = partition_leader_epoch.data(); c.magic = magic.data(); c.crc = crc.data();
        c.attributes = attributes.data(); c.last_offset_delta = last_offset_delta.data();
        c.base_timestamp = base_timestamp.data(); c.max_timestamp = max_timestamp.data();
        c.producer_id = producer_id.data(); c.producer_epoch = producer_epoch.data();
        c.base_sequence = base_sequence.data(); c.record_count = record_count.data();
        c.crc_computed = crc_computed.data(); c.data_off = data_off.data(); c.data_len = data_len.data();
        c.rec_base = rec_base.data(); c.hdr_base = hdr_base.data(); c.hdr_count = hdr_count.data();
        c.status = status.data(); c.rkey = rkey.data(); c.rval = rval.data();
        c.timestamp_delta = timestamp_delta.data(); c.rattributes = rattributes.data(); c.hdr = hdr.data();
        c.enc_len = enc_len.data();
        return c;
    }
};

inline std::optional<Bytes> span(const Bytes &data, uint32_t off, int32_t len) {
    if (len < 0) return std::nullopt;
    return data.slice(off, size_t(len));
}

// inflated::Frame::try_from over a whole frame: one blob, one device call
inline Result<inflated::Frame> inflate(Codec &codec, const deflated::Frame &frame, bool by_value) {
    size_t nb = frame.batches.size();
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off(nb + 1, 0);
    uint64_t rec_cap = 0;
    for (size_t b = 0; b < nb; b++) {
        Bytes e = frame.batches[b].into_bytes();
        blob.insert(blob.end(), e.data(), e.data() + e.len());
        off[b + 1] = blob.size();
        rec_cap += frame.batches[b].record_count;
    }
    // gzip, Snappy, LZ4 and zstd batches: the frame with them rewritten as uncompressed batches first
    // (Compression::inflator, lib.rs:1716-1765, as deflated.rs:444-457 calls it)
    std::vector<int32_t> inflated_status(nb ? nb : 1, NISSHI_NONE);
    bool compressed = false;
    for (const auto &d : frame.batches) compressed = compressed || ((d.attributes & 7) >= 1 && (d.attributes & 7) <= 4);
    if (compressed) {
        std::vector<uint64_t> out_off(nb + 1, 0);
        nisshi_totals room{};
        int st = nisshi_hip_inflate(codec.ctx(), blob.data(), blob.size(), off.data(), uint32_t(nb), nullptr, 0,
                                    out_off.data(), inflated_status.data(), &room);
        if (st && st != NISSHI_CAPACITY) return Error::from_status(st);
        std::vector<uint8_t> rewritten(room.out_bytes ? room.out_bytes : 1);
        st = nisshi_hip_inflate(codec.ctx(), blob.data(), blob.size(), off.data(), uint32_t(nb), rewritten.data(),
                                room.out_bytes, out_off.data(), inflated_status.data(), &room);
        if (st) return Error::from_status(st);
        rewritten.resize(room.out_bytes);
        blob.swap(rewritten);
        off.swap(out_off);
    }
    // a record is at least seven bytes, a header at least two
    if (rec_cap > blob.size() / 7 + 1) rec_cap = blob.size() / 7 + 1;
    // (a batch that ends on the sequential walk takes its header slots twice)
    uint64_t hdr_cap = blob.size() + 4096 * nb + 1;
    HostColumns cols(nb, rec_cap, hdr_cap);
    nisshi_columns c = cols.c();
    nisshi_totals totals{};
    int st = nisshi_hip_decode(codec.ctx(), blob.data(), blob.size(), off.data(), uint32_t(nb), by_value ? 1 : 0,
                               &c, rec_cap, hdr_cap, &totals);
    if (st) return Error::from_status(st);
    inflated::Frame out;
    out.batches.reserve(nb);
    for (size_t b = 0; b < nb; b++) {
        const deflated::Batch &d = frame.batches[b];
        // the first failing batch is the frame's error (collect::<Result<Vec<_>>>)
        // (this mirror has no CPU path of its own: where the Rust patch hands such a batch to the existing
        // code, which also accepts a member or frame damaged only behind its last record, the mirror reports)
        if (inflated_status[b] == NISSHI_INFLATE) return Error::from_status(NISSHI_INFLATE, d.attributes & 7);
        if (cols.status[b] && inflated_status[b] == NISSHI_OK)
            return Error{Error::InflatedRecords, cols.status[b],
                         std::string("records of an inflated batch do not decode: ") + nisshi_hip_status_str(cols.status[b])};
        if (cols.status[b]) return Error::from_status(cols.status[b], d.attributes & 7);
        // an inflated batch: its records lie in the rewritten frame; batch_length and, by reference,
        // attributes are the compressed batch's (inflated.rs:114-187 copies them from the deflated batch)
        bool was_inflated = inflated_status[b] == NISSHI_OK;
        Bytes record_data = was_inflated ? Bytes(blob.data() + off[b] + 61, size_t(off[b + 1] - off[b] - 61)) : d.record_data;
        inflated::Batch ib;
        ib.base_offset = cols.base_offset[b];
        ib.batch_length = was_inflated ? d.batch_length : cols.batch_length[b];
        ib.partition_leader_epoch = cols.partition_leader_epoch[b];
        ib.magic = cols.magic[b];
        ib.crc = cols.crc[b];
        ib.attributes = was_inflated && !by_value ? d.attributes : cols.attributes[b];
        ib.last_offset_delta = cols.last_offset_delta[b];
        ib.base_timestamp = cols.base_timestamp[b];
        ib.max_timestamp = cols.max_timestamp[b];
        ib.producer_id = cols.producer_id[b];
        ib.producer_epoch = cols.producer_epoch[b];
        ib.base_sequence = cols.base_sequence[b];
        ib.records.reserve(cols.record_count[b]);
        for (uint32_t r = cols.rec_base[b]; r < cols.rec_base[b] + cols.record_count[b]; r++) {
            Record rec;
            rec.length = cols.rkey[r].length;
            rec.attributes = cols.rattributes[r];
            rec.timestamp_delta = cols.timestamp_delta[r];
            rec.offset_delta = cols.rkey[r].offset_delta;
            rec.key = span(record_data, cols.rkey[r].key_off, cols.rkey[r].key_len);
            rec.value = span(record_data, cols.rval[r].val_off, cols.rval[r].val_len);
            rec.headers.reserve(size_t(cols.rval[r].hdr_count));
            for (uint32_t h = cols.rval[r].hdr_base; h < cols.rval[r].hdr_base + uint32_t(cols.rval[r].hdr_count); h++)
                rec.headers.push_back(Header{span(record_data, cols.hdr[h].key_off, cols.hdr[h].key_len),
                                             span(record_data, cols.hdr[h].val_off, cols.hdr[h].val_len)});
            ib.records.push_back(std::move(rec));
        }
        out.batches.push_back(std::move(ib));
    }
    return out;
}

// deflated::Frame::try_from(inflated::Frame): columns and one payload, one device call
inline Result<deflated::Frame> deflate(Codec &codec, const inflated::Frame &frame) {
    size_t nb = frame.batches.size(), nr = 0, nh = 0;
    for (const auto &b : frame.batches) {
        nr += b.records.size();
        for (const auto &r : b.records) nh += r.headers.size();
    }
    HostColumns cols(nb, nr, nh);
    std::vector<uint8_t> payload;
    std::vector<uint64_t> src_off(nb ? nb : 1, 0);
    // into_record_data (deflated.rs:341-386) compresses a batch's record data with the batch's codec: the
    // batches are encoded uncompressed first and those with a codec compressed behind that, in one call more
    std::vector<uint8_t> want(nb ? nb : 1, 0);
    bool any_codec = false;
    size_t r = 0, h = 0;
    for (size_t b = 0; b < nb; b++) {
        const inflated::Batch &ib = frame.batches[b];
        want[b] = uint8_t(ib.attributes & 7);
        any_codec = any_codec || want[b] != 0;
        cols.base_offset[b] = ib.base_offset;
        cols.batch_length[b] = ib.batch_length;
        cols.partition_leader_epoch[b] = ib.partition_leader_epoch;
        cols.magic[b] = ib.magic;
        cols.crc[b] = ib.crc;
        cols.attributes[b] = int16_t(ib.attributes & ~7);
        cols.last_offset_delta[b] = ib.last_offset_delta;
        cols.base_timestamp[b] = ib.base_timestamp;
        cols.max_timestamp[b] = ib.max_timestamp;
        cols.producer_id[b] = ib.producer_id;
        cols.producer_epoch[b] = ib.producer_epoch;
        cols.base_sequence[b] = ib.base_sequence;
        cols.record_count[b] = uint32_t(ib.records.size());
        cols.rec_base[b] = uint32_t(r);
        cols.hdr_base[b] = uint32_t(h);
        cols.status[b] = NISSHI_OK;
        size_t base = payload.size();
        src_off[b] = base;
        size_t h0 = h;
        auto put = [&](const std::optional<Bytes> &s, uint32_t &off, int32_t &len) {
            off = uint32_t(payload.size() - base);
            len = s ? int32_t(s->len()) : -1;
            if (s) payload.insert(payload.end(), s->data(), s->data() + s->len());
        };
        for (const Record &rec : ib.records) {
            cols.rkey[r].length = rec.length;
            cols.rkey[r].offset_delta = rec.offset_delta;
            cols.rattributes[r] = rec.attributes;
            cols.timestamp_delta[r] = rec.timestamp_delta;
            put(rec.key, cols.rkey[r].key_off, cols.rkey[r].key_len);
            put(rec.value, cols.rval[r].val_off, cols.rval[r].val_len);
            cols.rval[r].hdr_base = uint32_t(h);
            cols.rval[r].hdr_count = int32_t(rec.headers.size());
            for (const Header &hd : rec.headers) {
                put(hd.key, cols.hdr[h].key_off, cols.hdr[h].key_len);
                put(hd.value, cols.hdr[h].val_off, cols.hdr[h].val_len);
                h++;
            }
            r++;
        }
        cols.hdr_count[b] = uint32_t(h - h0);
        cols.data_off[b] = base;
        cols.data_len[b] = uint32_t(payload.size() - base);
    }
    uint64_t out_cap = 2 * payload.size() + 128 * nb + 64 * nr + 1024;
    std::vector<uint8_t> out(out_cap);
    std::vector<uint64_t> out_off(nb + 1, 0);
    nisshi_columns c = cols.c();
    nisshi_totals totals{};
    int st = nisshi_hip_encode(codec.ctx(), &c, uint32_t(nb), nr, nh, payload.data(), payload.size(),
                               src_off.data(), out.data(), out_cap, out_off.data(), &totals);
    if (st) return Error::from_status(st);
    if (any_codec) {
        // LZ4 on the device (nisshi_hip_deflate); gzip and zstd, which the reference compresses too, are not
        // produced there, and Snappy the reference refuses itself (Error::UnexpectedType): Error::Compressed
        std::vector<uint8_t> packed(size_t(totals.out_bytes) + 64 * nb + 64);
        std::vector<uint64_t> packed_off(nb + 1, 0);
        std::vector<int32_t> status(nb ? nb : 1, 0);
        nisshi_totals t2{};
        st = nisshi_hip_deflate(codec.ctx(), out.data(), totals.out_bytes, out_off.data(), uint32_t(nb), want.data(),
                                packed.data(), packed.size(), packed_off.data(), status.data(), &t2);
        if (st) return Error::from_status(st);
        for (size_t b = 0; b < nb; b++)
            if (status[b] != NISSHI_OK) return Error::from_status(status[b], want[b]);
        out.swap(packed);
        out_off.swap(packed_off);
        totals.out_bytes = t2.out_bytes;
    }
    Bytes all(out.data(), size_t(totals.out_bytes));
    deflated::Frame f;
    f.batches.reserve(nb);
    for (size_t b = 0; b < nb; b++) {
        auto d = deflated::Batch::try_from(all.slice(size_t(out_off[b]), size_t(out_off[b + 1] - out_off[b])));
        if (d.is_err()) return d.error();
        f.batches.push_back(*d);
    }
    return f;
}

}  // namespace detail

inline Result<deflated::Batch> deflated::Batch::try_from(const Bytes &encoded) {
    detail::Cursor cur{encoded.data(), encoded.len()};
    Batch b;
    if (!cur.get(b.base_offset) || !cur.get(b.batch_length) || !cur.get(b.partition_leader_epoch) ||
        !cur.get(b.magic) || !cur.get(b.crc))
        return Error::from_status(NISSHI_TRY_GET);
    if (b.batch_length < 0) return Error::from_status(NISSHI_TRY_FROM_INT);
    // batch_length - 4 - 1 - 4 bytes are covered by the CRC and must be present
    if (b.batch_length < 9 || size_t(b.batch_length) - 9 > cur.left) return Error::from_status(NISSHI_OVERFLOW);
    if (!cur.get(b.attributes) || !cur.get(b.last_offset_delta) || !cur.get(b.base_timestamp) ||
        !cur.get(b.max_timestamp) || !cur.get(b.producer_id) || !cur.get(b.producer_epoch) ||
        !cur.get(b.base_sequence) || !cur.get(b.record_count))
        return Error::from_status(NISSHI_TRY_GET);
    if (b.batch_length < 49 || size_t(b.batch_length) - 49 > cur.left) return Error::from_status(NISSHI_OVERFLOW);
    b.record_data = encoded.slice(encoded.len() - cur.left, size_t(b.batch_length) - 49);
    return b;
}

inline Bytes deflated::Batch::into_bytes() const {
    std::vector<uint8_t> out;
    out.reserve(61 + record_data.len());
    detail::put_be(out, base_offset);
    detail::put_be(out, batch_length);
    detail::put_be(out, partition_leader_epoch);
    detail::put_be(out, magic);
    detail::put_be(out, crc);
    detail::put_be(out, attributes);
    detail::put_be(out, last_offset_delta);
    detail::put_be(out, base_timestamp);
    detail::put_be(out, max_timestamp);
    detail::put_be(out, producer_id);
    detail::put_be(out, producer_epoch);
    detail::put_be(out, base_sequence);
    detail::put_be(out, record_count);
    out.insert(out.end(), record_data.data(), record_data.data() + record_data.len());
    return Bytes(std::move(out));
}

inline Result<deflated::Frame> deflated::Frame::split(const Bytes &blob) {
    std::vector<uint64_t> off(blob.len() / 61 + 2);
    uint64_t n = 0;
    int st = nisshi_frame_split(blob.data(), blob.len(), off.data(), off.size(), &n);
    if (st) return Error::from_status(st);
    Frame f;
    for (uint64_t b = 0; b < n; b++) {
        auto d = Batch::try_from(blob.slice(size_t(off[b]), size_t(off[b + 1] - off[b])));
        if (d.is_err()) return d.error();
        f.batches.push_back(*d);
    }
    return f;
}

inline Result<inflated::Frame> inflated::Frame::try_from(Codec &codec, deflated::Frame &&frame) {
    return detail::inflate(codec, frame, true);
}
inline Result<inflated::Frame> inflated::Frame::try_from(Codec &codec, const deflated::Frame &frame) {
    return detail::inflate(codec, frame, false);
}
inline Result<inflated::Batch> inflated::Batch::try_from(Codec &codec, deflated::Batch &&batch) {
    deflated::Frame f;
    f.batches.push_back(std::move(batch));
    auto r = detail::inflate(codec, f, true);
    if (r.is_err()) return r.error();
    return std::move(r->batches[0]);
}
inline Result<inflated::Batch> inflated::Batch::try_from(Codec &codec, const deflated::Batch &batch) {
    deflated::Frame f;
    f.batches.push_back(batch);
    auto r = detail::inflate(codec, f, false);
    if (r.is_err()) return r.error();
    return std::move(r->batches[0]);
}
inline Result<deflated::Frame> deflated::Frame::try_from(Codec &codec, const inflated::Frame &frame) {
    return detail::deflate(codec, frame);
}
inline Result<deflated::Batch> deflated::Batch::try_from(Codec &codec, const inflated::Batch &batch) {
    inflated::Frame f;
    f.batches.push_back(batch);
    auto r = detail::deflate(codec, f);
    if (r.is_err()) return r.error();
    return std::move(r->batches[0]);
}

inline inflated::Builder inflated::Batch::builder() { return inflated::Builder(); }
// From<Batch> for Builder, inflated.rs:278-294
inline inflated::Builder inflated::Batch::into_builder() const {
    inflated::Builder b;
    b.base_offset = base_offset;
    b.partition_leader_epoch = partition_leader_epoch;
    b.magic = magic;
    b.attributes = attributes;
    b.last_offset_delta = last_offset_delta;
    b.base_timestamp = base_timestamp;
    b.max_timestamp = max_timestamp;
    b.producer_id = producer_id;
    b.producer_epoch = producer_epoch;
    b.base_sequence = base_sequence;
    for (const Record &r : records) b.records.items.push_back(record::Builder::from(r));
    return b;
}
inline Result<inflated::Batch> inflated::Builder::build(Codec &codec) const {
    size_t size = size_in_bytes();
    if (size > 0x7fffffffu) return Error::from_status(NISSHI_TRY_FROM_INT);
    inflated::Batch b;
    b.base_offset = base_offset;
    b.batch_length = int32_t(size);
    b.partition_leader_epoch = partition_leader_epoch;
    b.magic = magic;
    b.attributes = attributes;
    b.last_offset_delta = last_offset_delta;
    b.base_timestamp = base_timestamp;
    b.max_timestamp = max_timestamp;
    b.producer_id = producer_id;
    b.producer_epoch = producer_epoch;
    b.base_sequence = base_sequence;
    for (const record::Builder &rb : records.items) {
        auto r = rb.build();
        if (r.is_err()) return r.error();
        b.records.push_back(*r);
    }
    // compressed batches are not handled on this path (as everywhere in this library)
    if (attributes & 7) return Error::from_status((attributes & 7) > 4 ? NISSHI_UNKNOWN_COMPRESSION : NISSHI_COMPRESSED, attributes & 7);
    auto d = deflated::Batch::try_from(codec, b);
    if (d.is_err()) return d.error();
    b.crc = d->crc;
    return b;
}

}  // namespace record

// ---------------------------------------------------------------- callers in nisshi-storage

namespace storage {

// combine, batch.rs:661-709: the batches queued for one topic partition become
// one; the first is the sink, the records of the others are appended with their
// offset and timestamp deltas moved to the sink's terms, and the sink is
// deflated again.  No batches give Ok(None) (batch.rs:666-668); the first batch
// that does not inflate gives its error.  One device call.
inline Result<std::optional<record::deflated::Batch>> combine(Codec &codec,
                                                              const std::vector<record::deflated::Batch> &batches) {
    using record::deflated::Batch;
    if (batches.empty()) return std::optional<Batch>();
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off(batches.size() + 1, 0);
    for (size_t b = 0; b < batches.size(); b++) {
        Bytes e = batches[b].into_bytes();
        blob.insert(blob.end(), e.data(), e.data() + e.len());
        off[b + 1] = blob.size();
    }
    const uint32_t group_off[2] = {0, uint32_t(batches.size())};
    // moved deltas may need longer varints: at most 13 bytes more on a record of 7
    std::vector<uint8_t> out(3 * blob.size() + 64);
    uint64_t out_off[2] = {0, 0};
    int32_t status = 0;
    nisshi_totals totals{};
    int st = nisshi_hip_combine(codec.ctx(), blob.data(), blob.size(), off.data(), uint32_t(batches.size()),
                                group_off, 1, out.data(), out.size(), out_off, &status, &totals);
    if (st) return Error::from_status(st);
    if (status) return Error::from_status(status);
    auto d = Batch::try_from(Bytes(out.data() + out_off[0], size_t(out_off[1] - out_off[0])));
    if (d.is_err()) return d.error();
    return std::optional<Batch>(std::move(*d));
}

// The rows the SQL engines write for a produced batch: record_copy.sql's columns
// (pg.rs:853-862, without the topic partition id the caller holds) and
// header_copy.sql's (pg.rs:908).
struct RecordRow {
    int64_t offset;      // high.unwrap_or_default() + index, pg.rs:868-869
    int16_t attributes;  // the batch's, pg.rs:870
    int64_t timestamp;   // base_timestamp + timestamp_delta, milliseconds since the epoch, pg.rs:876
    std::optional<Bytes> key, value;
};
struct HeaderRow {
    int64_t offset;  // of the header's record, pg.rs:913-914
    std::optional<Bytes> key, value;
};
struct Rows {
    std::vector<RecordRow> records;
    std::vector<HeaderRow> headers;
};

// pg.rs:797-943: Batch::try_from(deflated) and the two row loops, as one device
// call.  check_time follows the PostgreSQL engine, where a negative timestamp is
// Error::TryFromInt (to_system_time, nisshi-sans-io/src/lib.rs:2189-2193); false
// follows lite.rs:791, which stores the sum as it is.
inline Result<Rows> produce_rows(Codec &codec, const record::deflated::Batch &deflated,
                                 std::optional<int64_t> high, bool check_time = true) {
    Bytes e = deflated.into_bytes();
    const uint64_t off[2] = {0, e.len()};
    uint64_t rec_cap = deflated.record_count;
    if (rec_cap > e.len() / 7 + 1) rec_cap = e.len() / 7 + 1;
    uint64_t hdr_cap = e.len() + 4096 + 1;
    record::detail::HostColumns cols(1, rec_cap, hdr_cap);
    std::vector<int64_t> row_offset(rec_cap + 1), row_ts(rec_cap + 1), hdr_offset(hdr_cap + 1);
    std::vector<uint32_t> row_batch(rec_cap + 1);
    int32_t status = 0;
    nisshi_rows rows{row_offset.data(), row_ts.data(), row_batch.data(), hdr_offset.data(), &status};
    nisshi_columns c = cols.c();
    nisshi_totals totals{};
    const int64_t h = high.value_or(0);
    int st = nisshi_hip_produce_rows(codec.ctx(), e.data(), e.len(), off, 1, &h, check_time ? 1 : 0, &c, &rows,
                                     rec_cap, hdr_cap, &totals);
    if (st) return Error::from_status(st);
    if (status) return Error::from_status(status, deflated.attributes & 7);
    Rows out;
    out.records.reserve(cols.record_count[0]);
    const Bytes &data = deflated.record_data;
    for (uint32_t r = cols.rec_base[0]; r < cols.rec_base[0] + cols.record_count[0]; r++) {
        out.records.push_back(RecordRow{row_offset[r], cols.attributes[0], row_ts[r],
                                        record::detail::span(data, cols.rkey[r].key_off, cols.rkey[r].key_len),
                                        record::detail::span(data, cols.rval[r].val_off, cols.rval[r].val_len)});
    }
    for (uint32_t r = cols.rec_base[0]; r < cols.rec_base[0] + cols.record_count[0]; r++)
        for (uint32_t s = cols.rval[r].hdr_base; s < cols.rval[r].hdr_base + uint32_t(cols.rval[r].hdr_count); s++)
            out.headers.push_back(HeaderRow{hdr_offset[s],
                                            record::detail::span(data, cols.hdr[s].key_off, cols.hdr[s].key_len),
                                            record::detail::span(data, cols.hdr[s].val_off, cols.hdr[s].val_len)});
    return out;
}

// One row of record_fetch_pg.sql as the fetch reads it (pg.rs:1877-1931, 1960-1996)
// with the rows header_fetch.sql gives for its offset (pg.rs:2004-2036).
struct FetchRow {
    int64_t offset = 0;                    // column 0
    std::optional<int16_t> attributes;     // column 1, None -> 0
    int64_t timestamp = 0;                 // column 2 through to_timestamp: milliseconds since the epoch
    std::optional<Bytes> key, value;       // columns 3, 4
    std::optional<int64_t> producer_id;    // column 6, None -> -1
    std::optional<int16_t> producer_epoch; // column 7, None -> -1
    bool completed = true;                 // column 8
    std::vector<record::Header> headers;
};

// pg.rs:1875-2052 (with_completed) and lite.rs:2765-2980 (without): the rows of
// one topic partition, in offset order, to deflated batches.  `now` stands for the
// SystemTime::now() of inflated::Batch::builder() (inflated.rs:311-329).  An offset
// delta that does not fit i32 is Error::TryFromInt (pg.rs:1973).  One device call.
inline Result<std::vector<record::deflated::Batch>> fetch(Codec &codec, const std::vector<FetchRow> &rows,
                                                          int64_t now, bool with_completed = true) {
    using record::deflated::Batch;
    const size_t n = rows.size();
    std::vector<int64_t> offset(n + 1), timestamp(n + 1), producer_id(n + 1);
    std::vector<int16_t> attributes(n + 1), producer_epoch(n + 1);
    std::vector<uint8_t> completed(n + 1), payload;
    std::vector<uint32_t> key_off(n + 1), val_off(n + 1), hdr_base(n + 2, 0);
    std::vector<int32_t> key_len(n + 1), val_len(n + 1);
    std::vector<nisshi_header_span> hdr;
    auto put = [&payload](const std::optional<Bytes> &b, uint32_t &off, int32_t &len) {
        off = uint32_t(payload.size());
        len = b ? int32_t(b->len()) : -1;
        if (b) payload.insert(payload.end(), b->data(), b->data() + b->len());
    };
    size_t n_batches = 1;
    for (size_t i = 0; i < n; i++) {
        const FetchRow &r = rows[i];
        offset[i] = r.offset;
        attributes[i] = r.attributes.value_or(0);
        timestamp[i] = r.timestamp;
        producer_id[i] = r.producer_id.value_or(-1);
        producer_epoch[i] = r.producer_epoch.value_or(-1);
        completed[i] = r.completed ? 1 : 0;
        put(r.key, key_off[i], key_len[i]);
        put(r.value, val_off[i], val_len[i]);
        for (const record::Header &h : r.headers) {
            nisshi_header_span sp{};
            put(h.key, sp.key_off, sp.key_len);
            put(h.value, sp.val_off, sp.val_len);
            hdr.push_back(sp);
        }
        hdr_base[i + 1] = uint32_t(hdr.size());
        if (i && (attributes[i] != attributes[i - 1] || producer_id[i] != producer_id[i - 1] ||
                  producer_epoch[i] != producer_epoch[i - 1]))
            n_batches++;
    }
    hdr.push_back(nisshi_header_span{});
    nisshi_fetch_rows fr{offset.data(), attributes.data(), timestamp.data(), producer_id.data(),
                         producer_epoch.data(), with_completed ? completed.data() : nullptr,
                         key_off.data(), key_len.data(), val_off.data(), val_len.data(), hdr_base.data(), hdr.data()};
    std::vector<uint8_t> out(payload.size() + 61 * n_batches + 64 * n + 32 * hdr.size() + 1024);
    std::vector<uint64_t> out_off(n_batches + 1, 0);
    uint32_t got = 0;
    nisshi_totals totals{};
    int st = nisshi_hip_fetch(codec.ctx(), &fr, n, payload.data(), payload.size(), now, out.data(), out.size(),
                              out_off.data(), uint32_t(n_batches), &got, &totals);
    if (st) return Error::from_status(st);
    std::vector<Batch> batches;
    for (uint32_t b = 0; b < got; b++) {
        auto d = Batch::try_from(Bytes(out.data() + out_off[b], size_t(out_off[b + 1] - out_off[b])));
        if (d.is_err()) return d.error();
        batches.push_back(std::move(*d));
    }
    return batches;
}

// What log compaction does with one stored batch (slate/storage.rs:299-317).
struct Compaction {
    enum Action { Unchanged, Rewritten, Deleted } action = Unchanged;
    size_t records = 0;                           // Compaction::records: what the reference counts as removed
    std::optional<record::deflated::Batch> batch; // Rewritten: the batch to put back
};

// slate/storage.rs:283-312 with inflated::Batch::compact and keys (inflated.rs:213-275)
// over the batches of one topic partition, oldest first: a record goes when its
// key appears in a newer batch or later in its own; records without a key and
// control batches stay.  A batch that does not inflate gives its error.  One device call.
inline Result<std::vector<Compaction>> compact(Codec &codec, const std::vector<record::deflated::Batch> &batches) {
    using record::deflated::Batch;
    const size_t n = batches.size();
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off(n + 1, 0);
    for (size_t b = 0; b < n; b++) {
        Bytes e = batches[b].into_bytes();
        blob.insert(blob.end(), e.data(), e.data() + e.len());
        off[b + 1] = blob.size();
    }
    // a rebuilt length may take up to four bytes more than the stored one, on records of seven and more
    std::vector<uint8_t> out(blob.size() + blob.size() / 7 * 4 + 64);
    std::vector<uint64_t> out_off(n + 1, 0);
    std::vector<uint32_t> removed(n + 1, 0);
    std::vector<int32_t> action(n + 1, 0), status(n + 1, 0);
    nisshi_totals totals{};
    int st = nisshi_hip_compact(codec.ctx(), blob.data(), blob.size(), off.data(), uint32_t(n), nullptr, out.data(),
                                out.size(), out_off.data(), removed.data(), action.data(), status.data(), &totals);
    if (st) return Error::from_status(st);
    std::vector<Compaction> result(n);
    for (size_t b = 0; b < n; b++) {
        if (status[b] != NISSHI_OK && status[b] != NISSHI_NONE) return Error::from_status(status[b], batches[b].attributes & 7);
        result[b].records = removed[b];
        result[b].action = action[b] == 1 ? Compaction::Rewritten : action[b] == 2 ? Compaction::Deleted : Compaction::Unchanged;
        if (action[b] == 1) {
            auto d = Batch::try_from(Bytes(out.data() + out_off[b], size_t(out_off[b + 1] - out_off[b])));
            if (d.is_err()) return d.error();
            result[b].batch = std::move(*d);
        }
    }
    return result;
}

}  // namespace storage
}  // namespace nisshi
