//! One declaration per symbol of `include/nisshi_hip.h` (the C ABI of
//! libnisshi_hip.so), and the structs it passes, field for field.  The `extern "C"`
//! block is what `tools/gen_ffi.py` prints from the header; `tests/test_rust_ffi.py`
//! keeps the two in step.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

/// Per-batch and per-call status (the anonymous enum of the header).
pub const NISSHI_OK: c_int = 0;
pub const NISSHI_TRY_GET: c_int = 1;
pub const NISSHI_OVERFLOW: c_int = 2;
pub const NISSHI_TRY_FROM_INT: c_int = 3;
pub const NISSHI_UNKNOWN_COMPRESSION: c_int = 4;
pub const NISSHI_MALFORMED: c_int = 5;
pub const NISSHI_COMPRESSED: c_int = 6;
pub const NISSHI_CAPACITY: c_int = 7;
pub const NISSHI_DEVICE_ERROR: c_int = 8;
pub const NISSHI_INVALID_ARGUMENT: c_int = 9;
pub const NISSHI_NONE: c_int = 10;
pub const NISSHI_INFLATE: c_int = 11;

/// Limits of `nisshi_hip_decode_dev_static`.
pub const NISSHI_STATIC_MAX_BATCHES: u32 = 16384;
pub const NISSHI_STATIC_MAX_BYTES: u64 = 512 << 20;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, Eq, PartialEq)]
pub struct RecordKey {
    pub length: i32,
    pub offset_delta: i32,
    pub key_off: u32,
    pub key_len: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, Eq, PartialEq)]
pub struct RecordValue {
    pub val_off: u32,
    pub val_len: i32,
    pub hdr_base: u32,
    pub hdr_count: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, Eq, PartialEq)]
pub struct HeaderSpan {
    pub key_off: u32,
    pub key_len: i32,
    pub val_off: u32,
    pub val_len: i32,
}

/// `nisshi_columns`: one pointer a column; field order is ABI.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct Columns {
    pub base_offset: *mut i64,
    pub batch_length: *mut i32,
    pub partition_leader_epoch: *mut i32,
    pub magic: *mut i8,
    pub crc: *mut u32,
    pub attributes: *mut i16,
    pub last_offset_delta: *mut i32,
    pub base_timestamp: *mut i64,
    pub max_timestamp: *mut i64,
    pub producer_id: *mut i64,
    pub producer_epoch: *mut i16,
    pub base_sequence: *mut i32,
    pub record_count: *mut u32,
    pub crc_computed: *mut u32,
    pub data_off: *mut u64,
    pub data_len: *mut u32,
    pub rec_base: *mut u32,
    pub hdr_base: *mut u32,
    pub hdr_count: *mut u32,
    pub status: *mut i32,
    pub rkey: *mut RecordKey,
    pub rval: *mut RecordValue,
    pub timestamp_delta: *mut i64,
    pub rattributes: *mut u8,
    pub hdr: *mut HeaderSpan,
    /// Optional (may be null): per batch, the bytes its records encode to plus one where
    /// the decode found them written the shortest way, else zero.
    pub enc_len: *mut u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, Eq, PartialEq)]
pub struct Totals {
    pub n_records: u64,
    pub n_headers: u64,
    pub n_failed: u32,
    pub n_exact: u32,
    pub out_bytes: u64,
}

/// `nisshi_rows`: the row columns of the produce seam.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct Rows {
    pub offset: *mut i64,
    pub timestamp: *mut i64,
    pub batch: *mut u32,
    pub hdr_offset: *mut i64,
    pub status: *mut i32,
}

/// `nisshi_fetch_rows`: the result set of the fetch seam as columns.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct FetchRows {
    pub offset: *const i64,
    pub attributes: *const i16,
    pub timestamp: *const i64,
    pub producer_id: *const i64,
    pub producer_epoch: *const i16,
    pub completed: *const u8,
    pub key_off: *const u32,
    pub key_len: *const i32,
    pub val_off: *const u32,
    pub val_len: *const i32,
    pub hdr_base: *const u32,
    pub hdr: *const HeaderSpan,
}

/// `nisshi_hip_ctx`: opaque.
#[repr(C)]
pub struct Ctx {
    _private: [u8; 0],
}

/// `nisshi_hip_pool`: opaque.
#[repr(C)]
pub struct Pool {
    _private: [u8; 0],
}

/// `nisshi_pool_decode_fn`: the decode of one device of a custom pool, the arguments of `nisshi_hip_decode`.
pub type PoolDecodeFn = unsafe extern "C" fn(
    decoder_ctx: *mut c_void,
    blob: *const u8,
    blob_len: u64,
    batch_off: *const u64,
    n_batches: u32,
    by_value: c_int,
    cols: *const Columns,
    rec_cap: u64,
    hdr_cap: u64,
    totals: *mut Totals,
) -> c_int;

/// `NISSHI_HIP_ABI_VERSION` of the header these declarations follow.
pub const NISSHI_HIP_ABI_VERSION: c_int = 2;

extern "C" {
    pub fn nisshi_hip_abi_version() -> c_int;
    pub fn nisshi_hip_ctx_create(device: c_int, ctx: *mut *mut Ctx) -> c_int;
    pub fn nisshi_hip_ctx_destroy(ctx: *mut Ctx);
    pub fn nisshi_hip_status_str(status: c_int) -> *const c_char;
    pub fn nisshi_hip_last_error(ctx: *const Ctx) -> *const c_char;
    pub fn nisshi_hip_last_walked_whole(ctx: *const Ctx) -> u64;
    pub fn nisshi_frame_split(blob: *const u8, len: u64, batch_off: *mut u64, cap: u64, n_batches: *mut u64) -> c_int;
    pub fn nisshi_records_prefix_decode(buf: *const u8, len: u64, flexible: c_int, pos: *mut u64, records_len: *mut u32, is_null: *mut c_int) -> c_int;
    pub fn nisshi_records_prefix_encode(records_len: u32, flexible: c_int, out: *mut u8) -> u64;
    pub fn nisshi_records_fields_split(buf: *const u8, len: u64, field_pos: *const u64, n_fields: u32, flexible: c_int, batch_off: *mut u64, cap: u64, first_batch: *mut u32, n_batches: *mut u64) -> c_int;
    pub fn nisshi_hip_count_dev(ctx: *mut Ctx, d_blob: *const u8, blob_len: u64, d_batch_off: *const u64, n_batches: u32, n_records: *mut u64, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_decode_dev(ctx: *mut Ctx, d_blob: *const u8, blob_len: u64, d_batch_off: *const u64, n_batches: u32, by_value: c_int, d_cols: *const Columns, rec_cap: u64, hdr_cap: u64, totals: *mut Totals, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_decode_dev_static(ctx: *mut Ctx, d_blob: *const u8, blob_len: u64, d_batch_off: *const u64, n_batches: u32, by_value: c_int, d_cols: *const Columns, rec_cap: u64, hdr_cap: u64, totals: *mut Totals, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_encode_dev(ctx: *mut Ctx, d_cols: *const Columns, n_batches: u32, d_src: *const u8, d_src_data_off: *const u64, d_out: *mut u8, out_cap: u64, d_out_off: *mut u64, totals: *mut Totals, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_decode(ctx: *mut Ctx, blob: *const u8, blob_len: u64, batch_off: *const u64, n_batches: u32, by_value: c_int, cols: *const Columns, rec_cap: u64, hdr_cap: u64, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_encode(ctx: *mut Ctx, cols: *const Columns, n_batches: u32, n_records: u64, n_headers: u64, src: *const u8, src_len: u64, src_data_off: *const u64, out: *mut u8, out_cap: u64, out_off: *mut u64, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_produce_rows_dev(ctx: *mut Ctx, d_cols: *const Columns, n_batches: u32, n_slots: u64, hdr_cap: u64, d_high: *const i64, check_time: c_int, d_rows: *const Rows, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_produce_rows(ctx: *mut Ctx, blob: *const u8, blob_len: u64, batch_off: *const u64, n_batches: u32, high: *const i64, check_time: c_int, cols: *const Columns, rows: *const Rows, rec_cap: u64, hdr_cap: u64, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_combine_dev(ctx: *mut Ctx, d_cols: *const Columns, n_batches: u32, n_slots: u64, hdr_cap: u64, d_group_off: *const u32, n_groups: u32, d_out: *const Columns, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_combine(ctx: *mut Ctx, blob: *const u8, blob_len: u64, batch_off: *const u64, n_batches: u32, group_off: *const u32, n_groups: u32, out: *mut u8, out_cap: u64, out_off: *mut u64, group_status: *mut i32, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_fetch_batches_dev(ctx: *mut Ctx, d_rows: *const FetchRows, n_rows: u64, now: i64, d_out: *const Columns, batch_cap: u32, n_batches: *mut u32, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_builder_fields_dev(ctx: *mut Ctx, d_cols: *const Columns, n_batches: u32, d_out: *const u8, d_out_off: *const u64, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_fetch(ctx: *mut Ctx, rows: *const FetchRows, n_rows: u64, payload: *const u8, payload_len: u64, now: i64, out: *mut u8, out_cap: u64, out_off: *mut u64, batch_cap: u32, n_batches: *mut u32, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_compact_dev(ctx: *mut Ctx, d_blob: *const u8, d_cols: *const Columns, n_batches: u32, n_slots: u64, d_group: *const u32, d_out: *const Columns, d_removed: *mut u32, d_action: *mut i32, totals: *mut Totals, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_compact(ctx: *mut Ctx, blob: *const u8, blob_len: u64, batch_off: *const u64, n_batches: u32, group: *const u32, out: *mut u8, out_cap: u64, out_off: *mut u64, removed: *mut u32, action: *mut i32, status: *mut i32, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_pool_create(devices: *const c_int, n_devices: u32, pool: *mut *mut Pool) -> c_int;
    pub fn nisshi_hip_pool_destroy(pool: *mut Pool);
    pub fn nisshi_hip_pool_device_count(pool: *const Pool) -> u32;
    pub fn nisshi_hip_pool_device_of(pool: *const Pool, partition: u32) -> u32;
    pub fn nisshi_hip_pool_last_error(pool: *const Pool) -> *const c_char;
    pub fn nisshi_hip_pool_last_times(pool: *const Pool, ms: *mut f64);
    pub fn nisshi_hip_pool_decode(pool: *mut Pool, blob: *const u8, blob_len: u64, batch_off: *const u64, n_batches: u32, partition: *const u32, by_value: c_int, cols: *const Columns, rec_cap: u64, hdr_cap: u64, totals: *mut Totals, device_totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_pool_create_custom(n_devices: u32, decode: PoolDecodeFn, decoder_ctx: *const *mut c_void, pool: *mut *mut Pool) -> c_int;
    pub fn nisshi_hip_inflate_dev(ctx: *mut Ctx, d_blob: *const u8, blob_len: u64, d_batch_off: *const u64, n_batches: u32, d_out: *mut u8, out_cap: u64, d_out_off: *mut u64, d_status: *mut i32, totals: *mut Totals, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_inflate(ctx: *mut Ctx, blob: *const u8, blob_len: u64, batch_off: *const u64, n_batches: u32, out: *mut u8, out_cap: u64, out_off: *mut u64, status: *mut i32, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_deflate_dev(ctx: *mut Ctx, d_blob: *const u8, blob_len: u64, d_batch_off: *const u64, n_batches: u32, d_codec: *const u8, d_out: *mut u8, out_cap: u64, d_out_off: *mut u64, d_status: *mut i32, totals: *mut Totals, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_deflate(ctx: *mut Ctx, blob: *const u8, blob_len: u64, batch_off: *const u64, n_batches: u32, codec: *const u8, out: *mut u8, out_cap: u64, out_off: *mut u64, status: *mut i32, totals: *mut Totals) -> c_int;
    pub fn nisshi_hip_host_alloc(ctx: *mut Ctx, bytes: u64, ptr: *mut *mut c_void) -> c_int;
    pub fn nisshi_hip_host_free(ctx: *mut Ctx, ptr: *mut c_void);
    pub fn nisshi_hip_crc32c_dev(ctx: *mut Ctx, d_data: *const u8, len: u64, crc: *mut u32, stream: *mut c_void) -> c_int;
    pub fn nisshi_hip_profile_enable(ctx: *mut Ctx, on: c_int) -> c_int;
    pub fn nisshi_hip_profile_get(ctx: *mut Ctx, kernel_id: c_int, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn nisshi_hip_profile_name(kernel_id: c_int) -> *const c_char;
}
