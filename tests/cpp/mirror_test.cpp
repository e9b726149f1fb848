// Tests of include/nisshi.hpp written after the reference's own unit tests
// (nisshi-sans-io/src/record/inflated.rs:533-618, src/record.rs:478-523,
// src/record/deflated.rs tests, nisshi-storage/src/batch.rs:1198-1349): the same golden bytes and the same expectations,
// through the C++ types that mirror the reference's.  `mirror_test cpu` runs the
// cases that need no device, `mirror_test gpu` all of them.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "compressed_vectors.h"
#include "nisshi.hpp"

using namespace nisshi;
using namespace nisshi::record;

static int checks = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        checks++;                                                           \
        if (!(cond)) {                                                      \
            fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                        \
        }                                                                   \
    } while (0)

// inflated.rs:560-564: one record, value "def"
static const uint8_t kBatch71[] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 59, 255, 255, 255, 255, 2, 67, 41, 231, 61, 0, 0, 0, 0, 0, 0, 0,
    0, 1, 141, 116, 152, 137, 53, 0, 0, 1, 141, 116, 152, 137, 53, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0,
    0, 0, 1, 0, 0, 0, 1, 18, 0, 0, 0, 1, 6, 100, 101, 102, 0};

static Bytes bytes_of(const char *s) { return Bytes(reinterpret_cast<const uint8_t *>(s), strlen(s)); }

static void deflated_batch_from_bytes() {
    Bytes encoded(kBatch71, sizeof(kBatch71));
    auto r = deflated::Batch::try_from(encoded);
    CHECK(r.is_ok());
    const deflated::Batch &b = *r;
    CHECK(b.base_offset == 0 && b.batch_length == 59 && b.partition_leader_epoch == -1 && b.magic == 2);
    CHECK(b.crc == 1126819645u && b.attributes == 0 && b.last_offset_delta == 0);
    CHECK(b.base_timestamp == 1707058170165 && b.max_timestamp == 1707058170165);
    CHECK(b.producer_id == 1 && b.producer_epoch == 0 && b.base_sequence == 1 && b.record_count == 1);
    CHECK(b.record_data.len() == 10 && b.record_data[0] == 18);
    CHECK(b.max_offset() == 0);
    // From<Batch> for Bytes gives the input back
    CHECK(b.into_bytes() == encoded);
}

static void deflated_batch_errors() {
    Bytes encoded(kBatch71, sizeof(kBatch71));
    // under-run inside the fixed fields: Error::TryGet
    for (size_t n : {0, 7, 8, 11, 16, 20}) {
        auto r = deflated::Batch::try_from(encoded.slice(0, n));
        CHECK(r.is_err() && r.error().kind == Error::TryGet);
    }
    // the bytes the CRC covers are missing: Error::Overflow
    auto r = deflated::Batch::try_from(encoded.slice(0, 40));
    CHECK(r.is_err() && r.error().kind == Error::Overflow);
    std::vector<uint8_t> v(kBatch71, kBatch71 + sizeof(kBatch71));
    v[8] = 0x80;  // batch_length negative: usize::try_from fails
    r = deflated::Batch::try_from(Bytes(v));
    CHECK(r.is_err() && r.error().kind == Error::TryFromInt);
    v[8] = 0;
    v[11] = 5;  // batch_length smaller than the fields it must cover
    r = deflated::Batch::try_from(Bytes(v));
    CHECK(r.is_err() && r.error().kind == Error::Overflow);
    v[11] = 48;  // covers the CRC'd fields but not the fixed 49 bytes
    r = deflated::Batch::try_from(Bytes(v));
    CHECK(r.is_err() && r.error().kind == Error::Overflow);
}

static void frame_split() {
    std::vector<uint8_t> two(kBatch71, kBatch71 + sizeof(kBatch71));
    two.insert(two.end(), kBatch71, kBatch71 + sizeof(kBatch71));
    auto f = deflated::Frame::split(Bytes(two));
    CHECK(f.is_ok() && f->batches.size() == 2 && f->batches[0] == f->batches[1]);
}

static void inflate_batch(Codec &codec) {
    auto d = deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap();
    auto r = inflated::Batch::try_from(codec, d);  // by reference
    CHECK(r.is_ok());
    const inflated::Batch &b = *r;
    CHECK(b.base_offset == 0 && b.batch_length == 59 && b.crc == 1126819645u && b.producer_id == 1);
    CHECK(b.records.size() == 1);
    const Record &rec = b.records[0];
    CHECK(rec.length == 9 && rec.attributes == 0 && rec.timestamp_delta == 0 && rec.offset_delta == 0);
    CHECK(!rec.key.has_value() && rec.value.has_value() && *rec.value == bytes_of("def") && rec.headers.empty());
    // back to the wire: the same bytes
    auto back = deflated::Batch::try_from(codec, b);
    CHECK(back.is_ok() && *back == d && back->into_bytes() == Bytes(kBatch71, sizeof(kBatch71)));
}

static void attributes_by_value_and_by_reference(Codec &codec) {
    auto d = deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap();
    d.attributes = 0x0190;  // transactional, plus two bits BatchAttribute does not know
    auto by_ref = inflated::Batch::try_from(codec, d);
    CHECK(by_ref.is_ok() && by_ref->attributes == 0x0190);
    auto by_val = inflated::Batch::try_from(codec, deflated::Batch(d));
    CHECK(by_val.is_ok() && by_val->attributes == 0x0010);
    d.attributes = 5;  // no such compression
    auto bad = inflated::Batch::try_from(codec, d);
    CHECK(bad.is_err() && bad.error().kind == Error::UnknownCompressionType && bad.error().detail == 5);
    d.attributes = 4;  // zstd, which these ten bytes are not a frame of
    bad = inflated::Batch::try_from(codec, d);
    CHECK(bad.is_err() && bad.error().kind == Error::Inflate);
    d.attributes = 3;  // lz4, which these ten bytes are not a frame of
    bad = inflated::Batch::try_from(codec, d);
    CHECK(bad.is_err() && bad.error().kind == Error::Inflate);
}

// deflated.rs:589-674, 762-905 (decode_gzip, decode_lz4, decode_snappy): a compressed batch of one record whose
// value is LOREM; deflated.rs:676-760 (decode_zstd) likewise.
static void inflate_compressed_batches(Codec &codec) {
    struct Case { const uint8_t *bytes; size_t len; int codec; };
    // (the reference's arrays run a few bytes past their batch: Batch::try_from takes what batch_length says)
    const Case cases[] = {{kGzipBatch, sizeof(kGzipBatch), 1}, {kSnappyBatch, sizeof(kSnappyBatch), 2}, {kLz4Batch, sizeof(kLz4Batch), 3},
                          {kZstdBatch, sizeof(kZstdBatch), 4}};
    for (const Case &k : cases) {
        auto d = deflated::Batch::try_from(Bytes(k.bytes, k.len)).unwrap();
        CHECK((d.attributes & 7) == k.codec && d.record_count == 1);
        auto by_val = inflated::Batch::try_from(codec, deflated::Batch(d));
        CHECK(by_val.is_ok());
        CHECK((by_val->attributes & 7) == 0 && by_val->records.size() == 1);
        // (the inflated batch carries the compressed batch's batch_length and crc, inflated.rs:114-150)
        CHECK(by_val->batch_length == d.batch_length && by_val->crc == d.crc && by_val->base_timestamp == d.base_timestamp);
        const Record &rec = by_val->records[0];
        CHECK(rec.length == kLoremRecordLength && rec.attributes == 0 && rec.timestamp_delta == 0 && rec.offset_delta == 0);
        CHECK(!rec.key.has_value() && rec.value.has_value() && *rec.value == Bytes(kLorem, sizeof(kLorem)) && rec.headers.empty());
        auto by_ref = inflated::Batch::try_from(codec, d);
        CHECK(by_ref.is_ok() && by_ref->attributes == d.attributes && by_ref->records == by_val->records);
        // a frame of the compressed batch between two plain ones: the others are untouched
        deflated::Frame frame;
        frame.batches.push_back(deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap());
        frame.batches.push_back(d);
        frame.batches.push_back(frame.batches[0]);
        auto all = inflated::Frame::try_from(codec, frame);
        CHECK(all.is_ok() && all->batches.size() == 3 && all->batches[1].records == by_val->records);
        CHECK(all->batches[0].records.size() == 1 && all->batches[0] == all->batches[2]);
        // the stream cut short: the inflator's error
        deflated::Batch cut = d;
        cut.record_data = d.record_data.slice(0, d.record_data.len() - 9);
        cut.batch_length -= 9;
        auto bad = inflated::Batch::try_from(codec, cut);
        CHECK(bad.is_err() && bad.error().kind == Error::Inflate && bad.error().detail == k.codec);
    }
    auto bad = inflated::Batch::try_from(codec, deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap());
    // A Snappy block that inflates, to six bytes that are no record (a key of length -3).  The reference
    // reads an inflated batch's records through serde over the inflator's reader (deflated.rs:444-457)
    // and returns Err there, where Record::decode on an uncompressed batch would panic: the mirror
    // reports it as an error of its own kind, never as Malformed.
    const uint8_t block[] = {6, 5 << 2, 0x0A, 0x00, 0x00, 0x00, 0x05, 0x00};
    deflated::Batch odd = deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap();
    odd.attributes = 2;
    odd.record_data = Bytes(block, sizeof(block));
    odd.batch_length = int32_t(49 + sizeof(block));
    bad = inflated::Batch::try_from(codec, odd);
    CHECK(bad.is_err() && bad.error().kind == Error::InflatedRecords && bad.error().detail == NISSHI_MALFORMED);
    odd.attributes = 0;  // the same six bytes as an uncompressed batch's record data: the reference panics
    odd.record_data = Bytes(block + 2, 6);
    odd.batch_length = 49 + 6;
    bad = inflated::Batch::try_from(codec, odd);
    CHECK(bad.is_err() && bad.error().kind == Error::Malformed);
}

// deflated::Batch::try_from(inflated::Batch) of a batch whose attributes name a codec (deflated.rs:392-409 through
// into_record_data, deflated.rs:341-386): LZ4 is compressed on the device and inflates to the same records; gzip and
// zstd (which the reference compresses) and Snappy (which it refuses) are Error::Compressed here.
static void deflate_with_a_codec(Codec &codec) {
    auto d = deflated::Batch::try_from(Bytes(kLz4Batch, sizeof(kLz4Batch))).unwrap();
    auto by_ref = inflated::Batch::try_from(codec, d).unwrap();  // attributes verbatim: codec 3
    CHECK((by_ref.attributes & 7) == 3 && by_ref.records.size() == 1);
    auto packed = deflated::Batch::try_from(codec, by_ref);
    CHECK(packed.is_ok());
    CHECK((packed->attributes & 7) == 3 && packed->record_count == 1 && packed->base_timestamp == d.base_timestamp);
    CHECK(packed->batch_length == int32_t(49 + packed->record_data.len()));
    CHECK(packed->record_data.len() < size_t(kLoremRecordLength) && packed->record_data[0] == 0x04 && packed->record_data[1] == 0x22);
    auto back = inflated::Batch::try_from(codec, *packed);
    CHECK(back.is_ok() && back->records == by_ref.records && back->attributes == by_ref.attributes);
    // a frame of it between plain batches
    inflated::Frame frame;
    frame.batches.push_back(inflated::Batch::try_from(codec, deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap()).unwrap());
    frame.batches.push_back(by_ref);
    frame.batches.push_back(frame.batches[0]);
    auto all = deflated::Frame::try_from(codec, frame);
    CHECK(all.is_ok() && all->batches.size() == 3 && all->batches[1] == *packed);
    CHECK(all->batches[0].into_bytes() == Bytes(kBatch71, sizeof(kBatch71)) && all->batches[2] == all->batches[0]);
    // the same records as a gzip batch (a member: magic 1f 8b, CM 8) and as a zstd batch (a frame: magic 28 b5 2f fd), and back
    for (int other : {1, 4}) {
        inflated::Batch b = by_ref;
        b.attributes = int16_t((b.attributes & ~7) | other);
        auto z = deflated::Batch::try_from(codec, b);
        CHECK(z.is_ok() && (z->attributes & 7) == other && z->batch_length == int32_t(49 + z->record_data.len()));
        CHECK(z->record_data.len() < size_t(kLoremRecordLength));
        if (other == 1) CHECK(z->record_data[0] == 0x1f && z->record_data[1] == 0x8b && z->record_data[2] == 8);
        if (other == 4) CHECK(z->record_data[0] == 0x28 && z->record_data[1] == 0xb5 && z->record_data[2] == 0x2f && z->record_data[3] == 0xfd);
        auto again = inflated::Batch::try_from(codec, *z);
        CHECK(again.is_ok() && again->records == by_ref.records && again->attributes == b.attributes);
    }
    for (int other : {2}) {
        inflated::Batch b = by_ref;
        b.attributes = int16_t((b.attributes & ~7) | other);
        auto r = deflated::Batch::try_from(codec, b);
        CHECK(r.is_err() && r.error().kind == Error::Compressed && r.error().detail == other);
    }
}

static void records_with_keys_and_headers_round_trip(Codec &codec) {
    inflated::Frame frame;
    for (int b = 0; b < 3; b++) {
        inflated::Batch ib;
        ib.base_offset = 100 * b;
        ib.base_timestamp = ib.max_timestamp = 1700000000000 + b;
        ib.last_offset_delta = 4;
        for (int i = 0; i < 5; i++) {
            Record r;
            r.attributes = 0;
            r.timestamp_delta = i * 1000003LL * (b + 1);  // several varint widths
            r.offset_delta = i;
            if (i != 1) r.key = bytes_of(i % 2 ? "k" : "a-longer-key");
            if (i != 3) r.value = Bytes(std::vector<uint8_t>(size_t(40 * i + b), uint8_t(i)));
            for (int h = 0; h < i % 3; h++)
                r.headers.push_back(Header{h ? std::optional<Bytes>(bytes_of("hk")) : std::nullopt,
                                           std::optional<Bytes>(bytes_of("header-value"))});
            ib.records.push_back(r);
        }
        frame.batches.push_back(ib);
    }
    auto wire = deflated::Frame::try_from(codec, frame);
    CHECK(wire.is_ok() && wire->batches.size() == 3);
    for (const auto &d : wire->batches) CHECK(d.record_count == 5 && d.batch_length == int32_t(49 + d.record_data.len()));
    auto back = inflated::Frame::try_from(codec, *wire);
    CHECK(back.is_ok() && back->batches.size() == 3);
    for (size_t b = 0; b < 3; b++) {
        const auto &got = back->batches[b];
        const auto &want = frame.batches[b];
        CHECK(got.records.size() == want.records.size());
        for (size_t i = 0; i < want.records.size(); i++) {
            Record w = want.records[i];
            // the builder's length is the size of the record body (record.rs:329-421); the
            // encoder emits the stored length verbatim, so compare everything but it
            w.length = got.records[i].length;
            CHECK(got.records[i] == w);
        }
        CHECK(got.crc == wire->batches[b].crc && got.batch_length == wire->batches[b].batch_length);
    }
}

static void truncated_record_data_is_an_error(Codec &codec) {
    auto d = deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap();
    d.record_data = d.record_data.slice(0, 6);  // the value's bytes are gone
    d.batch_length -= 4;
    auto r = inflated::Batch::try_from(codec, d);
    // Octets::decode panics on a length past the end (codec.rs:70-74)
    CHECK(r.is_err() && r.error().kind == Error::Malformed);
    d.record_data = d.record_data.slice(0, 0);
    d.batch_length = 49;
    r = inflated::Batch::try_from(codec, d);
    CHECK(r.is_err() && r.error().kind == Error::TryGet);
}

// batch.rs:1204-1242: one batch per list of values
static std::vector<deflated::Batch> into_batches(Codec &codec, int16_t attributes, int64_t producer_id,
                                                 int16_t producer_epoch, int64_t base_offset,
                                                 const std::vector<std::vector<std::string>> &batches) {
    std::vector<deflated::Batch> split;
    int32_t base_sequence = 0;
    int64_t ts = 1700000000000;
    for (const auto &batch : batches) {
        inflated::Batch ib;
        ib.attributes = attributes;
        ib.producer_id = producer_id;
        ib.producer_epoch = producer_epoch;
        ib.base_offset = base_offset;
        ib.last_offset_delta = int32_t(batch.size()) - 1;
        ib.base_sequence = base_sequence;
        ib.base_timestamp = ib.max_timestamp = ts;
        ts += 250;  // Builder::default takes the time of day: later batches are later
        int32_t offset_delta = 0;
        for (const auto &value : batch) {
            Record r;
            r.value = bytes_of(value.c_str());
            r.offset_delta = offset_delta++;
            r.length = int32_t(6 + value.size());  // the body: five one-byte fields, the value, no headers
            ib.records.push_back(r);
        }
        split.push_back(deflated::Batch::try_from(codec, ib).unwrap());
        base_sequence += int32_t(batch.size());
    }
    return split;
}

// batch.rs:1198-1202
static void combine_empty(Codec &codec) {
    auto r = storage::combine(codec, {});
    CHECK(r.is_ok() && !r->has_value());
}

// batch.rs:1244-1349
static void combine_batches(Codec &codec) {
    const std::vector<std::vector<std::string>> batches = {{"a", "b", "c"}, {"f", "g"}, {"i"}, {"j"}, {"k"},
                                                            {"p", "q", "r", "s"}};
    const int64_t producer_id = 54345, base_offset = 0;
    const int16_t producer_epoch = 32123, attributes = 0;
    auto split = into_batches(codec, attributes, producer_id, producer_epoch, base_offset, batches);
    auto deflated = storage::combine(codec, split);
    CHECK(deflated.is_ok() && deflated->has_value());
    auto combined = inflated::Batch::try_from(codec, deflated::Batch(**deflated));
    CHECK(combined.is_ok());
    CHECK(combined->producer_id == producer_id && combined->producer_epoch == producer_epoch);
    CHECK(combined->base_sequence == 0 && combined->base_offset == base_offset && combined->attributes == attributes);
    size_t index = 0;
    for (const auto &batch : batches)
        for (const auto &value : batch) {
            CHECK(index < combined->records.size());
            const Record &rec = combined->records[index];
            CHECK(!rec.key.has_value());
            CHECK(rec.value.has_value() && *rec.value == bytes_of(value.c_str()));
            CHECK(size_t(rec.offset_delta) == index);
            index++;
        }
    CHECK(index == 12 && combined->records.size() == 12);
    // batch.rs:690-691, 697-698
    CHECK(combined->last_offset_delta == 11);
    CHECK(combined->max_timestamp == 1700000000000 + 5 * 250);
    CHECK(combined->records[3].timestamp_delta == -250 && combined->records[11].timestamp_delta == -1250);
    // a member that does not inflate fails the whole combine with its error
    split[2].attributes = 6;
    auto bad = storage::combine(codec, split);
    CHECK(bad.is_err() && bad.error().kind == Error::UnknownCompressionType);
}

// pg.rs:867-936 on the golden batch and on one with keys, nulls and headers
static void produce_rows(Codec &codec) {
    auto d = deflated::Batch::try_from(Bytes(kBatch71, sizeof(kBatch71))).unwrap();
    auto rows = storage::produce_rows(codec, d, std::nullopt);
    CHECK(rows.is_ok() && rows->records.size() == 1 && rows->headers.empty());
    CHECK(rows->records[0].offset == 0 && rows->records[0].timestamp == 1707058170165);
    CHECK(!rows->records[0].key.has_value() && *rows->records[0].value == bytes_of("def"));
    inflated::Batch ib;
    ib.attributes = 0x0110;  // transactional and an unknown bit: the rows carry 0x10
    ib.base_timestamp = ib.max_timestamp = 5000;
    ib.last_offset_delta = 2;
    for (int i = 0; i < 3; i++) {
        Record r;
        r.offset_delta = i;
        r.timestamp_delta = 7 * i - 3;
        if (i != 1) r.key = bytes_of("key");
        if (i != 2) r.value = bytes_of(i ? "v" : "value-0");
        for (int h = 0; h < i; h++) r.headers.push_back(Header{bytes_of("h"), h ? std::nullopt : std::optional<Bytes>(bytes_of("x"))});
        ib.records.push_back(r);
    }
    auto wire = deflated::Batch::try_from(codec, ib).unwrap();
    rows = storage::produce_rows(codec, wire, int64_t(1000));
    CHECK(rows.is_ok() && rows->records.size() == 3 && rows->headers.size() == 3);
    for (int i = 0; i < 3; i++) {
        const auto &row = rows->records[size_t(i)];
        CHECK(row.offset == 1000 + i && row.timestamp == 5000 + 7 * i - 3 && row.attributes == 0x10);
        CHECK(row.key == ib.records[size_t(i)].key && row.value == ib.records[size_t(i)].value);
    }
    CHECK(rows->headers[0].offset == 1001 && rows->headers[1].offset == 1002 && rows->headers[2].offset == 1002);
    CHECK(*rows->headers[1].key == bytes_of("h") && *rows->headers[1].value == bytes_of("x") &&
          !rows->headers[2].value.has_value());
    // a timestamp before the epoch: Error::TryFromInt in the PostgreSQL engine, stored by the SQLite one
    ib.base_timestamp = 2;
    wire = deflated::Batch::try_from(codec, ib).unwrap();
    auto neg = storage::produce_rows(codec, wire, std::nullopt, true);
    CHECK(neg.is_err() && neg.error().kind == Error::TryFromInt);
    neg = storage::produce_rows(codec, wire, std::nullopt, false);
    CHECK(neg.is_ok() && neg->records[0].timestamp == -1);
}

// pg.rs:1875-2052: rows to batches; the record of record.rs:478-523 (value "def", length 9)
static void fetch_rows(Codec &codec) {
    const int64_t now = 1700000123456;
    auto none = storage::fetch(codec, {}, now);
    CHECK(none.is_ok() && none->size() == 1);  // pg.rs:2050-2052: the default batch
    CHECK((*none)[0].record_count == 0 && (*none)[0].base_timestamp == now && (*none)[0].max_timestamp == now &&
          (*none)[0].producer_id == -1 && (*none)[0].producer_epoch == 0 && (*none)[0].base_sequence == 0 &&
          (*none)[0].partition_leader_epoch == -1 && (*none)[0].magic == 2 && (*none)[0].batch_length == 49);
    std::vector<storage::FetchRow> rows(5);
    for (int i = 0; i < 5; i++) {
        rows[size_t(i)].offset = 40 + i;
        rows[size_t(i)].timestamp = 9000 + 10 * i;
        rows[size_t(i)].value = bytes_of("def");
        if (i >= 3) rows[size_t(i)].producer_id = 77, rows[size_t(i)].producer_epoch = int16_t(2);
    }
    rows[1].key = bytes_of("k");
    rows[1].headers.push_back(Header{bytes_of("h"), std::nullopt});
    auto got = storage::fetch(codec, rows, now);
    CHECK(got.is_ok() && got->size() == 2);
    const deflated::Batch &a = (*got)[0], &b = (*got)[1];
    CHECK(a.base_offset == 40 && a.record_count == 3 && a.last_offset_delta == 2 && a.base_timestamp == 9000 &&
          a.producer_id == -1 && a.producer_epoch == -1 && a.max_timestamp == now);
    CHECK(b.base_offset == 43 && b.record_count == 2 && b.last_offset_delta == 1 && b.base_timestamp == 9030 &&
          b.producer_id == 77 && b.producer_epoch == 2);
    const uint8_t first[] = {18, 0, 0, 0, 1, 6, 100, 101, 102, 0};  // record.rs:478-523
    CHECK(a.record_data.len() > sizeof(first) && !memcmp(a.record_data.data(), first, sizeof(first)));
    auto ia = inflated::Batch::try_from(codec, deflated::Batch(a)).unwrap();
    CHECK(ia.records[1].key == rows[1].key && ia.records[1].headers.size() == 1 && ia.records[1].offset_delta == 1 &&
          ia.records[1].timestamp_delta == 10 && ia.records[2].length == 9);
    // pg.rs:1965-1971: an open transaction's row that does not follow its predecessor ends the fetch
    rows[2].completed = false;
    got = storage::fetch(codec, rows, now);
    CHECK(got.is_ok() && got->size() == 2 && (*got)[0].record_count == 3);  // adjacent offsets: it continues
    rows[2].offset = 50;
    got = storage::fetch(codec, rows, now);
    CHECK(got.is_ok() && got->size() == 1 && (*got)[0].record_count == 2 && (*got)[0].last_offset_delta == 1);
    got = storage::fetch(codec, rows, now, false);  // lite.rs: no such check
    CHECK(got.is_ok() && got->size() == 2 && (*got)[0].record_count == 3);
    rows[2].completed = true;
    rows[2].offset = 40 + (int64_t(1) << 32);  // pg.rs:1973
    auto bad = storage::fetch(codec, rows, now);
    CHECK(bad.is_err() && bad.error().kind == Error::TryFromInt);
}

static deflated::Batch keyed_batch(Codec &codec, int64_t base_offset,
                                   const std::vector<std::pair<const char *, const char *>> &kv) {
    inflated::Batch ib;
    ib.base_offset = base_offset;
    ib.last_offset_delta = int32_t(kv.size()) - 1;
    int32_t od = 0;
    for (const auto &p : kv) {
        Record r;
        r.offset_delta = od++;
        if (p.first) r.key = bytes_of(p.first);
        r.value = bytes_of(p.second);
        ib.records.push_back(r);
    }
    return deflated::Batch::try_from(codec, ib).unwrap();
}

// inflated.rs:664-790, the head given as a newer batch holding those keys
static void compact_batches(Codec &codec) {
    auto newer = keyed_batch(codec, 100, {{"k1", "x"}, {"k2", "x"}});
    // compact_noop_without_duplicates
    auto r = storage::compact(codec, {keyed_batch(codec, 0, {{"k1", "v1"}, {"k2", "v2"}})});
    CHECK(r.is_ok() && r->size() == 1 && (*r)[0].records == 0 && (*r)[0].action == storage::Compaction::Unchanged);
    // compact_within_batch_keeps_last_occurrence_per_key
    r = storage::compact(codec, {keyed_batch(codec, 5, {{"k1", "v1"}, {"k2", "v2"}, {"k1", "v3"}})});
    CHECK(r.is_ok() && (*r)[0].records == 1 && (*r)[0].action == storage::Compaction::Rewritten);
    auto kept = inflated::Batch::try_from(codec, deflated::Batch(*(*r)[0].batch)).unwrap();
    CHECK(kept.records.size() == 2 && kept.records[0].offset_delta == 1 && *kept.records[0].key == bytes_of("k2") &&
          kept.records[1].offset_delta == 2 && *kept.records[1].value == bytes_of("v3"));
    CHECK(kept.base_offset == 5 && kept.last_offset_delta == 2);
    // compact_drops_keys_in_head
    r = storage::compact(codec, {keyed_batch(codec, 0, {{"k1", "v1"}, {"k2", "v2"}}), newer});
    CHECK(r.is_ok() && (*r)[0].records == 2 && (*r)[0].action == storage::Compaction::Deleted && (*r)[1].records == 0);
    // compact_retains_keyless_records
    r = storage::compact(codec, {keyed_batch(codec, 0, {{nullptr, "v0"}, {"k1", "v1"}, {"k1", "v2"}})});
    CHECK(r.is_ok() && (*r)[0].records == 1);
    kept = inflated::Batch::try_from(codec, deflated::Batch(*(*r)[0].batch)).unwrap();
    CHECK(kept.records.size() == 2 && kept.records[0].offset_delta == 0 && !kept.records[0].key.has_value() &&
          kept.records[1].offset_delta == 2 && *kept.records[1].key == bytes_of("k1"));
    // compact_retains_keyless_when_all_keyed_records_drop
    r = storage::compact(codec, {keyed_batch(codec, 0, {{nullptr, "v0"}, {"k1", "v1"}}), newer});
    CHECK(r.is_ok() && (*r)[0].records == 1 && (*r)[0].action == storage::Compaction::Rewritten);
    kept = inflated::Batch::try_from(codec, deflated::Batch(*(*r)[0].batch)).unwrap();
    CHECK(kept.records.size() == 1 && kept.records[0].offset_delta == 0 && !kept.records[0].key.has_value());
}

// record.rs:457-475, 525-553; record/codec.rs:578-623; record/inflated.rs:533-554 (the sizes)
static void builder_sizes() {
    std::vector<uint8_t> def = {100, 101, 102};
    CHECK(octets_size(Bytes(def.data(), def.size())) == 4);
    CHECK(octets_size(std::nullopt) == 1);
    record::Builder rb = record::Builder().value(Bytes(def.data(), def.size()));
    CHECK(rb.size_in_bytes() == 9);
    auto rec = rb.build();
    CHECK(rec.is_ok() && rec->length == 9 && rec->attributes == 0 && !rec->key.has_value());
    Sequence<record::Builder> records({rb});
    CHECK(records.size_in_bytes() == 14);
    inflated::Builder b = inflated::Batch::builder();
    CHECK(b.producer_epoch == 0 && b.base_sequence == 0 && b.partition_leader_epoch == -1 && b.magic == 2 &&
          b.producer_id == -1 && b.max_timestamp == b.base_timestamp && b.base_timestamp > 1600000000000);
    CHECK(std::move(b).record(rb).size_in_bytes() == 59);
    // a body of 64..127 bytes: two bytes of zigzag prefix on the wire, one counted
    // (Sequence::size_in_bytes uses the unsigned varint's size, codec.rs:482-491)
    std::vector<uint8_t> v70(70, 7);
    record::Builder wide = record::Builder().value(Bytes(v70.data(), v70.size()));
    CHECK(wide.size_in_bytes() == 1 + 1 + 1 + 1 + 2 + 70 + 1);
    CHECK(Sequence<record::Builder>({wide}).size_in_bytes() == 4 + 1 + 77);
    CHECK(varint_size(77) == 2 && unsigned_varint_size(77) == 1 && long_varint_size(-1) == 1 && varint_size(63) == 1);
}

// deflated.rs:909-956: the transactional and the control batch of the reference's tests
static void batch_predicates() {
    deflated::Batch b;
    b.attributes = 16;
    b.producer_id = 5;
    b.base_sequence = 0;
    CHECK(b.is_transactional() && !b.is_control() && b.is_idempotent());
    b.attributes = 48;
    b.base_sequence = -1;
    b.base_offset = 1;
    CHECK(b.is_transactional() && b.is_control() && !b.is_idempotent() && b.max_offset() == 1);
    b.attributes = 0;
    b.producer_id = -1;
    b.base_sequence = 3;
    b.last_offset_delta = 9;
    CHECK(!b.is_transactional() && !b.is_control() && !b.is_idempotent() && b.max_offset() == 10);
}

// inflated.rs:533-554: Batch::builder() ... .build() gives batch_length 59 and crc 1126819645;
// then a batch whose records' sizes fall where the builder under-counts
static void builder_build(Codec &codec) {
    std::vector<uint8_t> def = {100, 101, 102};
    auto built = inflated::Batch::builder()
                     .with_base_offset(0)
                     .with_partition_leader_epoch(-1)
                     .with_magic(2)
                     .with_attributes(0)
                     .with_last_offset_delta(0)
                     .with_base_timestamp(1707058170165)
                     .with_max_timestamp(1707058170165)
                     .with_producer_id(1)
                     .with_producer_epoch(0)
                     .with_base_sequence(1)
                     .record(record::Builder().value(Bytes(def.data(), def.size())))
                     .build(codec);
    CHECK(built.is_ok() && built->batch_length == 59 && built->crc == 1126819645u);
    // it is the batch of the 71 golden bytes
    auto d = deflated::Batch::try_from(codec, *built);
    CHECK(d.is_ok() && d->into_bytes() == Bytes(kBatch71, sizeof(kBatch71)));
    // into_builder().build() of an inflated batch gives it back
    auto again = built->into_builder().build(codec);
    CHECK(again.is_ok() && *again == *built);
    std::vector<uint8_t> v70(70, 7), v20(20, 9);
    inflated::Builder b = inflated::Batch::builder().with_base_timestamp(5).with_max_timestamp(6);
    for (int i = 0; i < 5; i++)
        b = std::move(b).record(record::Builder().offset_delta(i).value(i % 2 ? Bytes(v20.data(), v20.size())
                                                                            : Bytes(v70.data(), v70.size())));
    auto q8 = b.build(codec);
    CHECK(q8.is_ok() && q8->records.size() == 5);
    auto wire = deflated::Batch::try_from(codec, *q8);
    CHECK(wire.is_ok() && wire->crc == q8->crc && wire->batch_length == q8->batch_length + 3);
}

int main(int argc, char **argv) {
    std::string mode = argc > 1 ? argv[1] : "cpu";
    deflated_batch_from_bytes();
    deflated_batch_errors();
    frame_split();
    builder_sizes();
    batch_predicates();
    if (mode == "gpu") {
        Codec codec(0);
        inflate_batch(codec);
        attributes_by_value_and_by_reference(codec);
        inflate_compressed_batches(codec);
        deflate_with_a_codec(codec);
        records_with_keys_and_headers_round_trip(codec);
        truncated_record_data_is_an_error(codec);
        combine_empty(codec);
        combine_batches(codec);
        produce_rows(codec);
        fetch_rows(codec);
        compact_batches(codec);
        builder_build(codec);
    }
    printf("ok %d checks (%s)\n", checks, mode.c_str());
    return 0;
}
