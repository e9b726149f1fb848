//! Safe surface over `libnisshi_hip.so`, the MI355X (gfx950) RecordBatch v2 codec.
//!
//! The crate knows nothing of `nisshi-sans-io`'s types: it hands decoded batches
//! over as columns ([`HostColumns`]) and takes columns back for the encode, so
//! `nisshi-sans-io` can depend on it (feature `hip`) and do the conversion to and
//! from `inflated::Batch` / `Record` itself (`src/record/hip.rs` of the patch in
//! `patches/nisshi-sans-io-hip.diff`).
//!
//! All `unsafe` of the workspace that concerns this codec is in this file and in
//! `ffi.rs`: raw pointers into vectors that outlive the call, nothing else.

pub mod ffi;

use std::{cell::RefCell, ffi::CStr, fmt, os::raw::c_int, ptr};

/// The per-batch and per-call outcomes of the C ABI.
#[derive(Clone, Copy, Debug, Eq, PartialEq)]
pub enum Status {
    Ok,
    /// `Error::TryGet`: the buffer ends inside a field
    TryGet,
    /// `Error::Overflow`
    Overflow,
    /// `Error::TryFromInt`: a negative batch_length, an offset delta that does not fit
    TryFromInt,
    /// `Error::UnknownCompressionType(attributes & 7)`, 5..=7
    UnknownCompression,
    /// an input on which the reference panics (codec.rs:70-74, 291-295; record.rs:260)
    Malformed,
    /// `attributes & 7` in 1..=4: not decoded by the device codec
    Compressed,
    Capacity,
    Device,
    InvalidArgument,
    /// no batch: `Ok(None)` of combine, a batch compaction leaves alone
    None,
    /// compressed record data that does not inflate: the inflator's `Error::Snap` / `Error::Io`
    Inflate,
}

impl From<c_int> for Status {
    fn from(v: c_int) -> Self {
        match v {
            ffi::NISSHI_OK => Self::Ok,
            ffi::NISSHI_TRY_GET => Self::TryGet,
            ffi::NISSHI_OVERFLOW => Self::Overflow,
            ffi::NISSHI_TRY_FROM_INT => Self::TryFromInt,
            ffi::NISSHI_UNKNOWN_COMPRESSION => Self::UnknownCompression,
            ffi::NISSHI_MALFORMED => Self::Malformed,
            ffi::NISSHI_COMPRESSED => Self::Compressed,
            ffi::NISSHI_CAPACITY => Self::Capacity,
            ffi::NISSHI_INVALID_ARGUMENT => Self::InvalidArgument,
            ffi::NISSHI_NONE => Self::None,
            ffi::NISSHI_INFLATE => Self::Inflate,
            _ => Self::Device,
        }
    }
}

/// A call that failed as a whole (a batch that fails leaves its status in the columns).
#[derive(Clone, Debug)]
pub struct Error {
    pub status: Status,
    pub detail: String,
}

impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "nisshi_hip: {:?} {}", self.status, self.detail)
    }
}

impl std::error::Error for Error {}

pub type Result<T> = std::result::Result<T, Error>;

/// Columns in host memory, one vector a column of `nisshi_columns`.
#[derive(Clone, Debug, Default)]
pub struct HostColumns {
    // per batch: the 13 fields of the 61-byte header (deflated.rs:67-82), then what the decode derives
    pub base_offset: Vec<i64>,
    pub batch_length: Vec<i32>,
    pub partition_leader_epoch: Vec<i32>,
    pub magic: Vec<i8>,
    pub crc: Vec<u32>,
    pub attributes: Vec<i16>,
    pub last_offset_delta: Vec<i32>,
    pub base_timestamp: Vec<i64>,
    pub max_timestamp: Vec<i64>,
    pub producer_id: Vec<i64>,
    pub producer_epoch: Vec<i16>,
    pub base_sequence: Vec<i32>,
    pub record_count: Vec<u32>,
    pub crc_computed: Vec<u32>,
    pub data_off: Vec<u64>,
    pub data_len: Vec<u32>,
    pub rec_base: Vec<u32>,
    pub hdr_base: Vec<u32>,
    pub hdr_count: Vec<u32>,
    pub status: Vec<i32>,
    // per record (record.rs:172-198)
    pub rkey: Vec<ffi::RecordKey>,
    pub rval: Vec<ffi::RecordValue>,
    pub timestamp_delta: Vec<i64>,
    pub rattributes: Vec<u8>,
    // per header (header.rs:21-29)
    pub hdr: Vec<ffi::HeaderSpan>,
    /// per batch: encoded bytes of its records plus one where the decode knows them, else zero
    /// (zero it for a batch whose records are edited before it is encoded again)
    pub enc_len: Vec<u32>,
}

impl HostColumns {
    /// Zeroed columns for `n_batches` batches, `rec_cap` records and `hdr_cap` headers
    /// (every array keeps at least one entry, so its pointer is never dangling).
    pub fn with_capacity(n_batches: usize, rec_cap: usize, hdr_cap: usize) -> Self {
        let (b, r, h) = (n_batches.max(1), rec_cap.max(1), hdr_cap.max(1));
        Self {
            base_offset: vec![0; b],
            batch_length: vec![0; b],
            partition_leader_epoch: vec![0; b],
            magic: vec![0; b],
            crc: vec![0; b],
            attributes: vec![0; b],
            last_offset_delta: vec![0; b],
            base_timestamp: vec![0; b],
            max_timestamp: vec![0; b],
            producer_id: vec![0; b],
            producer_epoch: vec![0; b],
            base_sequence: vec![0; b],
            record_count: vec![0; b],
            crc_computed: vec![0; b],
            data_off: vec![0; b],
            data_len: vec![0; b],
            rec_base: vec![0; b],
            hdr_base: vec![0; b],
            hdr_count: vec![0; b],
            status: vec![0; b],
            rkey: vec![ffi::RecordKey::default(); r],
            rval: vec![ffi::RecordValue::default(); r],
            timestamp_delta: vec![0; r],
            rattributes: vec![0; r],
            hdr: vec![ffi::HeaderSpan::default(); h],
            enc_len: vec![0; b],
        }
    }

    /// The pointer view the C ABI takes.  Valid while `self` is neither moved nor resized.
    fn as_ffi(&mut self) -> ffi::Columns {
        ffi::Columns {
            base_offset: self.base_offset.as_mut_ptr(),
            batch_length: self.batch_length.as_mut_ptr(),
            partition_leader_epoch: self.partition_leader_epoch.as_mut_ptr(),
            magic: self.magic.as_mut_ptr(),
            crc: self.crc.as_mut_ptr(),
            attributes: self.attributes.as_mut_ptr(),
            last_offset_delta: self.last_offset_delta.as_mut_ptr(),
            base_timestamp: self.base_timestamp.as_mut_ptr(),
            max_timestamp: self.max_timestamp.as_mut_ptr(),
            producer_id: self.producer_id.as_mut_ptr(),
            producer_epoch: self.producer_epoch.as_mut_ptr(),
            base_sequence: self.base_sequence.as_mut_ptr(),
            record_count: self.record_count.as_mut_ptr(),
            crc_computed: self.crc_computed.as_mut_ptr(),
            data_off: self.data_off.as_mut_ptr(),
            data_len: self.data_len.as_mut_ptr(),
            rec_base: self.rec_base.as_mut_ptr(),
            hdr_base: self.hdr_base.as_mut_ptr(),
            hdr_count: self.hdr_count.as_mut_ptr(),
            status: self.status.as_mut_ptr(),
            rkey: self.rkey.as_mut_ptr(),
            rval: self.rval.as_mut_ptr(),
            timestamp_delta: self.timestamp_delta.as_mut_ptr(),
            rattributes: self.rattributes.as_mut_ptr(),
            hdr: self.hdr.as_mut_ptr(),
            enc_len: self.enc_len.as_mut_ptr(),
        }
    }

    /// The status of batch `b` as an outcome.
    pub fn status_of(&self, b: usize) -> Status {
        Status::from(self.status[b] as c_int)
    }
}

/// What a decode or an encode counted.
pub type Totals = ffi::Totals;

/// One context on one GPU.  A context serves one thread at a time: keep one per
/// thread ([`with_thread_codec`]) or behind a lock.
pub struct Codec {
    ctx: *mut ffi::Ctx,
}

// The context holds device buffers and streams, no thread-local state.
unsafe impl Send for Codec {}

impl Drop for Codec {
    fn drop(&mut self) {
        unsafe { ffi::nisshi_hip_ctx_destroy(self.ctx) }
    }
}

impl Codec {
    /// `nisshi_hip_ctx_create`: fails with [`Status::Device`] without a gfx950 GPU (there is no CPU path).
    pub fn new(device: i32) -> Result<Self> {
        // (a library built from another header would read struct fields that are not there)
        let abi = unsafe { ffi::nisshi_hip_abi_version() };
        if abi != ffi::NISSHI_HIP_ABI_VERSION {
            return Err(Error {
                status: Status::InvalidArgument,
                detail: format!("libnisshi_hip has ABI version {abi}, this crate {}", ffi::NISSHI_HIP_ABI_VERSION),
            });
        }
        let mut ctx = ptr::null_mut();
        let st = unsafe { ffi::nisshi_hip_ctx_create(device as c_int, &mut ctx) };
        if st != ffi::NISSHI_OK {
            return Err(Error {
                status: st.into(),
                detail: format!("nisshi_hip_ctx_create(device {device})"),
            });
        }
        Ok(Self { ctx })
    }

    /// Batches of the last decode that were decoded chunk by chunk on guesses
    /// that did not hold, and so walked once more by one wave.
    pub fn last_walked_whole(&self) -> u64 {
        unsafe { ffi::nisshi_hip_last_walked_whole(self.ctx) }
    }

    fn fail(&self, st: c_int, what: &str) -> Error {
        let detail = unsafe {
            let p = ffi::nisshi_hip_last_error(self.ctx);
            if p.is_null() {
                String::new()
            } else {
                CStr::from_ptr(p).to_string_lossy().into_owned()
            }
        };
        Error {
            status: st.into(),
            detail: format!("{what}: {detail}"),
        }
    }

    /// The batches of a records blob (de.rs:1056-1093): the start of every batch and the end of the last.
    pub fn frame_split(blob: &[u8]) -> Result<Vec<u64>> {
        let mut off = vec![0u64; blob.len() / 61 + 2];
        let mut n = 0u64;
        let st = unsafe {
            ffi::nisshi_frame_split(
                blob.as_ptr(),
                blob.len() as u64,
                off.as_mut_ptr(),
                off.len() as u64,
                &mut n,
            )
        };
        if st != ffi::NISSHI_OK {
            return Err(Error {
                status: st.into(),
                detail: "nisshi_frame_split".into(),
            });
        }
        off.truncate(n as usize + 1);
        Ok(off)
    }

    /// `nisshi_hip_decode`: the batches `batch_off[i] .. batch_off[i + 1]` of `blob`
    /// into columns.  `by_value` clears the compression bits of `attributes` and drops
    /// unknown bits, as `inflated::Batch::try_from(deflated::Batch)` does
    /// (inflated.rs:123-140); `false` keeps them (`try_from(&deflated::Batch)`).
    /// `record_counts`: the sum of the batches' record counts when the caller knows it
    /// (`deflated::Batch::record_count`), else `None`.
    pub fn decode(
        &mut self,
        blob: &[u8],
        batch_off: &[u64],
        by_value: bool,
        record_counts: Option<u64>,
    ) -> Result<(HostColumns, Totals)> {
        let n = Self::checked_batches(blob, batch_off, "nisshi_hip_decode")?;
        // a record is at least seven bytes, a header at least two (header.rs:31-41)
        let most = blob.len() as u64 / 7 + 1;
        let rec_cap = record_counts.map_or(most, |c| c.min(most));
        // blob.len() + 1 header slots cover every call (see nisshi_hip_decode_dev in nisshi_hip.h), but
        // zeroing sixteen bytes a blob byte would cost more than the decode: start at an eighth and grow
        // while a batch reports that its headers had no room
        let hdr_most = blob.len() as u64 + 1;
        let mut hdr_cap = hdr_most.min((blob.len() as u64 / 8).max(1024));
        loop {
            let mut cols = HostColumns::with_capacity(n, rec_cap as usize, hdr_cap as usize);
            let mut totals = Totals::default();
            let c = cols.as_ffi();
            let st = unsafe {
                ffi::nisshi_hip_decode(
                    self.ctx,
                    blob.as_ptr(),
                    blob.len() as u64,
                    batch_off.as_ptr(),
                    n as u32,
                    by_value as c_int,
                    &c,
                    rec_cap,
                    hdr_cap,
                    &mut totals,
                )
            };
            if st != ffi::NISSHI_OK {
                return Err(self.fail(st, "nisshi_hip_decode"));
            }
            let short = hdr_cap < hdr_most
                && totals.n_failed > 0
                && (0..n).any(|b| cols.status_of(b) == Status::Capacity);
            if !short {
                return Ok((cols, totals));
            }
            hdr_cap = hdr_most.min((2 * hdr_cap).max(totals.n_headers));
        }
    }

    /// What the safe entry points check before the `unsafe` call: `batch_off` holds
    /// the start of every batch and the end of the last (so at least one entry), in
    /// ascending order, none past the end of `blob`.  The library checks the same
    /// (`NISSHI_INVALID_ARGUMENT`); the kernels read `blob + batch_off[b]` unchecked.
    fn checked_batches(blob: &[u8], batch_off: &[u64], what: &str) -> Result<usize> {
        let ok = !batch_off.is_empty()
            && batch_off.len() - 1 <= u32::MAX as usize
            && batch_off.windows(2).all(|w| w[0] <= w[1])
            && batch_off[batch_off.len() - 1] <= blob.len() as u64;
        if ok {
            Ok(batch_off.len() - 1)
        } else {
            Err(Error {
                status: Status::InvalidArgument,
                detail: format!("{what}: batch_off must hold n + 1 ascending offsets within the blob"),
            })
        }
    }

        Ok((cols, totals))
    }

    /// `nisshi_hip_encode`: batches from columns whose spans index `payload` at
    /// `src_data_off[b]`.  Returns the deflated bytes of all batches back to back and
    /// the start of each (`n + 1` offsets).  `record_count`, `batch_length` and `crc`
    /// are recomputed; every record's `length` is written as stored (record.rs:215-252).
    pub fn encode(
        &mut self,
        cols: &mut HostColumns,
        n_batches: usize,
        n_records: usize,
        n_headers: usize,
        payload: &[u8],
        src_data_off: &[u64],
    ) -> Result<(Vec<u8>, Vec<u64>, Totals)> {
        let out_cap = 2 * payload.len() + 128 * n_batches + 64 * n_records + 1024;
        let mut out = vec![0u8; out_cap];
        let mut out_off = vec![0u64; n_batches + 1];
        let mut totals = Totals::default();
        let c = cols.as_ffi();
        let st = unsafe {
            ffi::nisshi_hip_encode(
                self.ctx,
                &c,
                n_batches as u32,
                n_records as u64,
                n_headers as u64,
                payload.as_ptr(),
                payload.len() as u64,
                src_data_off.as_ptr(),
                out.as_mut_ptr(),
                out_cap as u64,
                out_off.as_mut_ptr(),
                &mut totals,
            )
        };
        if st != ffi::NISSHI_OK {
            return Err(self.fail(st, "nisshi_hip_encode"));
        }
        out.truncate(totals.out_bytes as usize);
        Ok((out, out_off, totals))
    }

    /// `nisshi_hip_inflate`: the frame with its gzip, Snappy and LZ4 batches rewritten as uncompressed
    /// ones (`Compression::inflator`, lib.rs:1716-1755), every other batch passed through.  Returns the
    /// frame's bytes, the start of each batch (`n + 1` offsets) and a status a batch: `Ok` inflated,
    /// `Inflate` the record data does not inflate (passed through), `None` passed through.
    pub fn inflate(&mut self, blob: &[u8], batch_off: &[u64]) -> Result<(Vec<u8>, Vec<u64>, Vec<Status>, Totals)> {
        let n = Self::checked_batches(blob, batch_off, "nisshi_hip_inflate")?;
        let mut out_off = vec![0u64; n + 1];
        let mut status = vec![0i32; n.max(1)];
        let mut totals = Totals::default();
        // the first call sizes (NISSHI_CAPACITY and the bytes needed), the second has the room
        let mut out: Vec<u8> = Vec::new();
        loop {
            let st = unsafe {
                ffi::nisshi_hip_inflate(
                    self.ctx,
                    blob.as_ptr(),
                    blob.len() as u64,
                    batch_off.as_ptr(),
                    n as u32,
                    if out.is_empty() { std::ptr::null_mut() } else { out.as_mut_ptr() },
                    out.len() as u64,
                    out_off.as_mut_ptr(),
                    status.as_mut_ptr(),
                    &mut totals,
                )
            };
            if st == ffi::NISSHI_CAPACITY && out.is_empty() && totals.out_bytes > 0 {
                out = vec![0u8; totals.out_bytes as usize];
                continue;
            }
            if st != ffi::NISSHI_OK {
                return Err(self.fail(st, "nisshi_hip_inflate"));
            }
            break;
        }
        out.truncate(totals.out_bytes as usize);
        status.truncate(n);
        Ok((out, out_off, status.into_iter().map(Status::from).collect(), totals))
    }

    /// `nisshi_hip_deflate`: a frame of uncompressed batches (what `encode` writes) with every batch whose
    /// `codec` is 3, 1 or 4 compressed as `into_record_data`'s Lz4, Gzip and Zstd arms do (deflated.rs:341-386):
    /// its record data one LZ4 frame, gzip member or zstd frame, `batch_length`, the codec bits and the crc to
    /// match.  `codec` 0: the batch as it came.  2 (Snappy) is not produced: the batch comes back as it went in
    /// with `Status::Compressed`, for the caller's own `into_record_data` (which refuses it too).  Returns the frame's bytes, the start of each batch and a status a batch.
    pub fn deflate(&mut self, blob: &[u8], batch_off: &[u64], codec: &[u8]) -> Result<(Vec<u8>, Vec<u64>, Vec<Status>, Totals)> {
        let n = Self::checked_batches(blob, batch_off, "nisshi_hip_deflate")?;
        if codec.len() != n {
            return Err(Error {
                status: Status::InvalidArgument,
                detail: format!("nisshi_hip_deflate: {} codecs for {n} batches", codec.len()),
            });
        }
        let mut out_off = vec![0u64; n + 1];
        let mut status = vec![0i32; n.max(1)];
        let mut totals = Totals::default();
        // (a batch grows by a few bytes an 8 KiB block at most)
        let mut out = vec![0u8; blob.len() + blob.len() / 512 + 64 * n + 64];
        let st = unsafe {
            ffi::nisshi_hip_deflate(
                self.ctx,
                blob.as_ptr(),
                blob.len() as u64,
                batch_off.as_ptr(),
                n as u32,
                codec.as_ptr(),
                out.as_mut_ptr(),
                out.len() as u64,
                out_off.as_mut_ptr(),
                status.as_mut_ptr(),
                &mut totals,
            )
        };
        if st != ffi::NISSHI_OK {
            return Err(self.fail(st, "nisshi_hip_deflate"));
        }
        out.truncate(totals.out_bytes as usize);
        status.truncate(n);
        Ok((out, out_off, status.into_iter().map(Status::from).collect(), totals))
    }

    /// The raw context, for the `*_dev` entry points and the storage operators of `ffi`.
    pub fn as_ptr(&mut self) -> *mut ffi::Ctx {
        self.ctx
    }
}

thread_local! {
    static CODEC: RefCell<Option<Codec>> = const { RefCell::new(None) };
}

/// Runs `f` with this thread's codec, created on first use on the device named by
/// `NISSHI_HIP_DEVICE` (default 0).  The conversions of `nisshi-sans-io` are called
/// from `spawn_blocking` threads and inline on async workers (frame.rs:318, 394;
/// pg.rs:797): a context a thread needs no lock.
pub fn with_thread_codec<T>(f: impl FnOnce(&mut Codec) -> T) -> Result<T> {
    CODEC.with(|cell| {
        let mut slot = cell.borrow_mut();
        if slot.is_none() {
            let device = std::env::var("NISSHI_HIP_DEVICE")
                .ok()
                .and_then(|v| v.parse().ok())
                .unwrap_or(0);
            *slot = Some(Codec::new(device)?);
        }
        Ok(f(slot.as_mut().expect("set above")))
    })
}
