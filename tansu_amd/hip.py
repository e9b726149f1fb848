"""ctypes binding of the C ABI in include/nisshi_hip.h (tansu_amd/lib/libnisshi_hip.so).

PyTorch is used only for device memory and streams.  There is no CPU fallback:
loading fails loudly when the HIP library has not been built, and creating a
codec fails when no GPU is present.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

from . import columns as C

# (NISSHI_HIP_LIB in the environment: another build of the library, for the tools that compare builds)
LIB_PATH = os.environ.get("NISSHI_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnisshi_hip.so")

ABI_VERSION = 2     # NISSHI_HIP_ABI_VERSION of the header these bindings follow

EXPORTS = [
    "nisshi_hip_abi_version", "nisshi_hip_ctx_create", "nisshi_hip_ctx_destroy", "nisshi_hip_status_str",
    "nisshi_hip_last_error", "nisshi_hip_last_walked_whole", "nisshi_frame_split", "nisshi_records_prefix_decode",
    "nisshi_records_prefix_encode", "nisshi_records_fields_split", "nisshi_hip_count_dev", "nisshi_hip_decode_dev",
    "nisshi_hip_decode_dev_static",
    "nisshi_hip_encode_dev", "nisshi_hip_decode", "nisshi_hip_encode", "nisshi_hip_crc32c_dev",
    "nisshi_hip_profile_enable", "nisshi_hip_profile_get", "nisshi_hip_profile_name",
    "nisshi_hip_host_alloc", "nisshi_hip_host_free",
    "nisshi_hip_produce_rows_dev", "nisshi_hip_produce_rows", "nisshi_hip_combine_dev",
    "nisshi_hip_combine", "nisshi_hip_fetch_batches_dev", "nisshi_hip_fetch", "nisshi_hip_builder_fields_dev",
    "nisshi_hip_compact_dev", "nisshi_hip_compact", "nisshi_hip_inflate_dev", "nisshi_hip_inflate",
    "nisshi_hip_deflate_dev", "nisshi_hip_deflate",
    "nisshi_hip_pool_create", "nisshi_hip_pool_create_custom", "nisshi_hip_pool_destroy",
    "nisshi_hip_pool_device_count", "nisshi_hip_pool_device_of", "nisshi_hip_pool_last_error",
    "nisshi_hip_pool_last_times", "nisshi_hip_pool_decode",
]

POOL_DECODE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                  ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                  ctypes.c_void_p)


class Totals(ctypes.Structure):
    _fields_ = [("n_records", ctypes.c_uint64), ("n_headers", ctypes.c_uint64),
                ("n_failed", ctypes.c_uint32), ("n_exact", ctypes.c_uint32),
                ("out_bytes", ctypes.c_uint64)]


# `nisshi_rows`: (name, dtype, level) with level 0 per batch, 1 per record slot, 2 per header slot
ROW_ARRAYS = [("offset", np.int64, 1), ("timestamp", np.int64, 1), ("batch", np.uint32, 1),
              ("hdr_offset", np.int64, 2), ("status", np.int32, 0)]
NO_BATCH = 0xFFFFFFFF
NONE = 10  # NISSHI_NONE
INFLATE = 11  # NISSHI_INFLATE


class CRows(ctypes.Structure):
    """ctypes view of `nisshi_rows`."""
    _fields_ = [(name, ctypes.c_void_p) for name, _, _ in ROW_ARRAYS]


# `nisshi_fetch_rows`: (name, dtype, level) with level 1 per row, 3 per row plus one, 2 per header
FETCH_ARRAYS = [("offset", np.int64, 1), ("attributes", np.int16, 1), ("timestamp", np.int64, 1),
                ("producer_id", np.int64, 1), ("producer_epoch", np.int16, 1), ("completed", np.uint8, 1),
                ("key_off", np.uint32, 1), ("key_len", np.int32, 1), ("val_off", np.uint32, 1),
                ("val_len", np.int32, 1), ("hdr_base", np.uint32, 3), ("hdr", np.uint32, 2)]


class CFetchRows(ctypes.Structure):
    """ctypes view of `nisshi_fetch_rows`."""
    _fields_ = [(name, ctypes.c_void_p) for name, _, _ in FETCH_ARRAYS]


def fetch_rows_host(rows: dict):
    """`rows`: numpy arrays by the names of FETCH_ARRAYS (`completed` may be None;
    `hdr` is (n_headers, 4) uint32).  Returns (struct, arrays kept alive)."""
    keep, c = {}, CFetchRows()
    for name, dt, _ in FETCH_ARRAYS:
        v = rows.get(name)
        if v is None:
            setattr(c, name, None)
            continue
        a = np.ascontiguousarray(v, dtype=dt)
        keep[name] = a if a.size else np.zeros(4, dtype=dt)
        setattr(c, name, keep[name].ctypes.data)
    return c, keep


class NisshiError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        name = C.STATUS_NAMES[status] if 0 <= status < len(C.STATUS_NAMES) else \
            {8: "DEVICE_ERROR", 9: "INVALID_ARGUMENT", 10: "NONE", 11: "INFLATE"}.get(status, str(status))
        super().__init__(f"nisshi_hip: {name} {detail}".strip())


_lib = None


def load() -> ctypes.CDLL:
    """Load libnisshi_hip.so; raises if it is missing (build it with tansu_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -m tansu_amd.build` "
                          "(there is no CPU fallback for the HIP codec)")
    # PyTorch ships a HIP runtime of its own; loading it first lets this library
    # bind to the same copy, whatever the order the caller imports things in (two
    # runtimes in one process do not share a device context)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    if ABI_VERSION is not None and (not hasattr(lib, "nisshi_hip_abi_version") or lib.nisshi_hip_abi_version() != ABI_VERSION):
        raise RuntimeError(f"{LIB_PATH} was built from another nisshi_hip.h than these bindings (rebuild: "
                           "python -m tansu_amd.build)")
    lib.nisshi_hip_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    lib.nisshi_hip_ctx_destroy.argtypes = [vp]
    lib.nisshi_hip_ctx_destroy.restype = None
    lib.nisshi_hip_host_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
    lib.nisshi_hip_host_free.argtypes = [vp, vp]
    lib.nisshi_hip_host_free.restype = None
    lib.nisshi_hip_status_str.argtypes = [i32]
    lib.nisshi_hip_status_str.restype = ctypes.c_char_p
    lib.nisshi_hip_last_error.argtypes = [vp]
    lib.nisshi_hip_last_error.restype = ctypes.c_char_p
    if ABI_VERSION is not None or hasattr(lib, "nisshi_hip_last_walked_whole"):   # (tools load older builds to compare)
        lib.nisshi_hip_last_walked_whole.argtypes = [vp]
        lib.nisshi_hip_last_walked_whole.restype = ctypes.c_uint64
    lib.nisshi_frame_split.argtypes = [vp, u64, vp, u64, ctypes.POINTER(u64)]
    lib.nisshi_records_prefix_decode.argtypes = [vp, u64, i32, ctypes.POINTER(u64),
                                                 ctypes.POINTER(u32), ctypes.POINTER(i32)]
    lib.nisshi_records_prefix_encode.argtypes = [u32, i32, vp]
    lib.nisshi_records_prefix_encode.restype = u64
    lib.nisshi_records_fields_split.argtypes = [vp, u64, vp, u32, i32, vp, u64, vp, ctypes.POINTER(u64)]
    lib.nisshi_hip_count_dev.argtypes = [vp, vp, u64, vp, u32, ctypes.POINTER(u64), vp]
    lib.nisshi_hip_decode_dev.argtypes = [vp, vp, u64, vp, u32, i32, ctypes.POINTER(C.CColumns),
                                          u64, u64, ctypes.POINTER(Totals), vp]
    lib.nisshi_hip_decode_dev_static.argtypes = lib.nisshi_hip_decode_dev.argtypes
    lib.nisshi_hip_encode_dev.argtypes = [vp, ctypes.POINTER(C.CColumns), u32, vp, vp, vp, u64, vp,
                                          ctypes.POINTER(Totals), vp]
    lib.nisshi_hip_decode.argtypes = [vp, vp, u64, vp, u32, i32, ctypes.POINTER(C.CColumns), u64,
                                      u64, ctypes.POINTER(Totals)]
    lib.nisshi_hip_encode.argtypes = [vp, ctypes.POINTER(C.CColumns), u32, u64, u64, vp, u64, vp,
                                      vp, u64, vp, ctypes.POINTER(Totals)]
    lib.nisshi_hip_crc32c_dev.argtypes = [vp, vp, u64, ctypes.POINTER(u32), vp]
    lib.nisshi_hip_produce_rows_dev.argtypes = [vp, ctypes.POINTER(C.CColumns), u32, u64, u64, vp, i32,
                                                ctypes.POINTER(CRows), vp]
    lib.nisshi_hip_produce_rows.argtypes = [vp, vp, u64, vp, u32, vp, i32, ctypes.POINTER(C.CColumns),
                                            ctypes.POINTER(CRows), u64, u64, ctypes.POINTER(Totals)]
    lib.nisshi_hip_combine_dev.argtypes = [vp, ctypes.POINTER(C.CColumns), u32, u64, u64, vp, u32,
                                           ctypes.POINTER(C.CColumns), vp]
    lib.nisshi_hip_combine.argtypes = [vp, vp, u64, vp, u32, vp, u32, vp, u64, vp, vp,
                                       ctypes.POINTER(Totals)]
    lib.nisshi_hip_fetch_batches_dev.argtypes = [vp, ctypes.POINTER(CFetchRows), u64, ctypes.c_int64,
                                                 ctypes.POINTER(C.CColumns), u32, ctypes.POINTER(u32), vp]
    lib.nisshi_hip_fetch.argtypes = [vp, ctypes.POINTER(CFetchRows), u64, vp, u64, ctypes.c_int64, vp, u64,
                                     vp, u32, ctypes.POINTER(u32), ctypes.POINTER(Totals)]
    lib.nisshi_hip_builder_fields_dev.argtypes = [vp, ctypes.POINTER(C.CColumns), u32, vp, vp, vp]
    lib.nisshi_hip_compact_dev.argtypes = [vp, vp, ctypes.POINTER(C.CColumns), u32, u64, vp,
                                           ctypes.POINTER(C.CColumns), vp, vp, ctypes.POINTER(Totals), vp]
    lib.nisshi_hip_compact.argtypes = [vp, vp, u64, vp, u32, vp, vp, u64, vp, vp, vp, vp, ctypes.POINTER(Totals)]
    lib.nisshi_hip_inflate_dev.argtypes = [vp, vp, u64, vp, u32, vp, u64, vp, vp, ctypes.POINTER(Totals), vp]
    lib.nisshi_hip_inflate.argtypes = [vp, vp, u64, vp, u32, vp, u64, vp, vp, ctypes.POINTER(Totals)]
    lib.nisshi_hip_deflate_dev.argtypes = [vp, vp, u64, vp, u32, vp, vp, u64, vp, vp, ctypes.POINTER(Totals), vp]
    lib.nisshi_hip_deflate.argtypes = [vp, vp, u64, vp, u32, vp, vp, u64, vp, vp, ctypes.POINTER(Totals)]
    lib.nisshi_hip_pool_create.argtypes = [vp, u32, ctypes.POINTER(vp)]
    lib.nisshi_hip_pool_create_custom.argtypes = [u32, POOL_DECODE_FN, vp, ctypes.POINTER(vp)]
    lib.nisshi_hip_pool_destroy.argtypes = [vp]
    lib.nisshi_hip_pool_destroy.restype = None
    lib.nisshi_hip_pool_device_count.argtypes = [vp]
    lib.nisshi_hip_pool_device_count.restype = u32
    lib.nisshi_hip_pool_device_of.argtypes = [vp, u32]
    lib.nisshi_hip_pool_device_of.restype = u32
    lib.nisshi_hip_pool_last_error.argtypes = [vp]
    lib.nisshi_hip_pool_last_error.restype = ctypes.c_char_p
    lib.nisshi_hip_pool_last_times.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    lib.nisshi_hip_pool_last_times.restype = None
    lib.nisshi_hip_pool_decode.argtypes = [vp, vp, u64, vp, u32, vp, i32, ctypes.POINTER(C.CColumns), u64, u64,
                                           ctypes.POINTER(Totals), vp]
    lib.nisshi_hip_profile_enable.argtypes = [vp, i32]
    lib.nisshi_hip_profile_get.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64)]
    lib.nisshi_hip_profile_name.argtypes = [i32]
    lib.nisshi_hip_profile_name.restype = ctypes.c_char_p
    _lib = lib
    return lib


def frame_split(blob: bytes) -> np.ndarray:
    """Split a records blob into batches (host side).  Returns n + 1 offsets."""
    lib = load()
    cap = len(blob) // 12 + 2
    off = np.zeros(cap, dtype=np.uint64)
    n = ctypes.c_uint64(0)
    buf = np.frombuffer(blob, dtype=np.uint8) if len(blob) else np.zeros(1, dtype=np.uint8)
    st = lib.nisshi_frame_split(buf.ctypes.data, len(blob), off.ctypes.data, cap, ctypes.byref(n))
    if st:
        raise NisshiError(st)
    return off[:n.value + 1].copy()


def records_fields_split(buf, field_pos, flexible: bool = True, cap: Optional[int] = None):
    """The batches of all records fields of one receive buffer (host side):
    (batch_off np.uint64[n + 1] into buf, first_batch np.uint32[n_fields + 1])."""
    lib = load()
    arr = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    pos = np.ascontiguousarray(field_pos, dtype=np.uint64)
    cap = len(arr) // 61 + 2 if cap is None else cap
    off = np.zeros(cap, dtype=np.uint64)
    first = np.zeros(len(pos) + 1, dtype=np.uint32)
    n = ctypes.c_uint64(0)
    st = lib.nisshi_records_fields_split(arr.ctypes.data, len(arr), pos.ctypes.data if len(pos) else None, len(pos),
                                         int(flexible), off.ctypes.data, cap, first.ctypes.data, ctypes.byref(n))
    if st:
        raise NisshiError(st)
    return off[:n.value + 1].copy(), first


_TORCH_BY_SIZE = None


def _torch_dtype(np_dtype):
    import torch
    global _TORCH_BY_SIZE
    if _TORCH_BY_SIZE is None:
        _TORCH_BY_SIZE = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
    return _TORCH_BY_SIZE[np.dtype(np_dtype).itemsize]


class DeviceColumns:
    """Columns resident in device memory: one torch tensor per array of the device
    layout (`t`), batch columns by field name, records and headers in their groups."""

    def __init__(self, n_batches: int, rec_cap: int, hdr_cap: int, device):
        import torch
        self.n_batches, self.rec_cap, self.hdr_cap = n_batches, rec_cap, hdr_cap
        counts = (n_batches, rec_cap, hdr_cap)
        self.t = {}
        self.c = C.CColumns()
        for name, dt, words, level in C.DEVICE_ARRAYS:
            n = max(counts[level], 1)
            self.t[name] = torch.empty((n, words) if words > 1 else n, dtype=_torch_dtype(dt),
                                       device=device)
            setattr(self.c, name, self.t[name].data_ptr())
        self.t["enc_len"].zero_()   # (nothing is known of the sizes of batches that no decode has filled in)

    def field(self, name: str):
        """The device tensor (a strided view, for grouped fields) of one field."""
        for gname, fields in C.RECORD_GROUPS + C.HEADER_GROUPS:
            if name in fields:
                return self.t[gname][:, fields.index(name)]
        return self.t[name]

    @classmethod
    def from_host(cls, cols: dict, device):
        """From one numpy array per field."""
        import torch
        self = cls(len(cols["status"]), len(cols["length"]), len(cols["hkey_off"]), device)
        for name, arr in C.pack(cols).items():
            n = len(cols[{"rkey": "length", "rval": "length", "hdr": "hkey_off", "enc_len": "status"}.get(name, name)])
            if n:
                signed = np.ascontiguousarray(arr[:n]).view(np.dtype(f"i{arr.dtype.itemsize}"))
                self.t[name][:n].copy_(torch.from_numpy(signed.copy()))
        return self

    def to_host(self, n_records: Optional[int] = None, n_headers: Optional[int] = None) -> dict:
        """One numpy array per field."""
        n_records = self.rec_cap if n_records is None else n_records
        n_headers = self.hdr_cap if n_headers is None else n_headers
        counts = (self.n_batches, n_records, n_headers)
        arrays = {}
        for name, dt, words, level in C.DEVICE_ARRAYS:
            arrays[name] = self.t[name][:counts[level]].cpu().numpy().view(dt).copy()
        return C.unpack(arrays)


class DeviceRows:
    """`nisshi_rows` in device memory: one torch tensor per array (`t`)."""

    def __init__(self, n_batches: int, rec_cap: int, hdr_cap: int, device):
        import torch
        counts = (n_batches, rec_cap, hdr_cap)
        self.t = {}
        self.c = CRows()
        for name, dt, level in ROW_ARRAYS:
            self.t[name] = torch.empty(max(counts[level], 1), dtype=_torch_dtype(dt), device=device)
            setattr(self.c, name, self.t[name].data_ptr())

    def to_host(self) -> dict:
        return {name: self.t[name].cpu().numpy().view(dt) for name, dt, _ in ROW_ARRAYS}


class Codec:
    """One context on one GPU.  Not thread safe: use one per thread."""

    def __init__(self, device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the HIP codec has no CPU fallback")
        self.lib = load()
        self.device = torch.device("cuda", device)
        ctx = ctypes.c_void_p()
        st = self.lib.nisshi_hip_ctx_create(device, ctypes.byref(ctx))
        if st:
            raise NisshiError(st, "ctx_create")
        self.ctx = ctx
        self._pinned = []

    def close(self):
        if getattr(self, "ctx", None):
            for ptr in getattr(self, "_pinned", []):
                self.lib.nisshi_hip_host_free(self.ctx, ptr)
            self._pinned = []
            self.lib.nisshi_hip_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st:
            raise NisshiError(st, (self.lib.nisshi_hip_last_error(self.ctx) or b"").decode())

    def last_walked_whole(self) -> int:
        """Batches of the last decode whose chunks' guesses did not hold (walked whole by one wave)."""
        return int(self.lib.nisshi_hip_last_walked_whole(self.ctx))

    def _stream(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- per-kernel timing ----------------------------------------------------
    def profile(self, on: bool, only: Optional[str] = None, every: int = 1):
        """Per-kernel event marks on or off; `only`: the name of the one decode kernel to
        time, in one decode call of `every` (the other calls carry no events at all)."""
        mode = int(bool(on))
        if on and only is not None:
            k = 0
            while True:
                name = self.lib.nisshi_hip_profile_name(k)
                if name is None:
                    raise ValueError(f"no kernel named {only}")
                if name.decode() == only:
                    break
                k += 1
            mode = 2 + k + ((max(1, min(every, 256)) - 1) << 8)
        self._check(self.lib.nisshi_hip_profile_enable(self.ctx, mode))

    def kernel_times(self) -> dict:
        """{kernel name: (total ms, launches)} since profile(True); waits for the events."""
        out = {}
        k = 0
        while True:
            name = self.lib.nisshi_hip_profile_name(k)
            if name is None:
                break
            ms, n = ctypes.c_double(0), ctypes.c_uint64(0)
            self._check(self.lib.nisshi_hip_profile_get(self.ctx, k, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[name.decode()] = (ms.value, n.value)
            k += 1
        return out

    # ---- device-resident path -------------------------------------------------
    def to_device(self, blob, batch_off):
        """Copy a host blob and its n + 1 batch offsets to the GPU."""
        import torch
        arr = np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        d_blob = torch.empty(len(arr) + 16, dtype=torch.uint8, device=self.device)
        if len(arr):
            d_blob[:len(arr)].copy_(torch.from_numpy(arr.copy()))
        off = np.ascontiguousarray(batch_off, dtype=np.uint64).view(np.int64)
        d_off = torch.from_numpy(off.copy()).to(self.device)
        return d_blob, d_off

    def count_dev(self, d_blob, blob_len: int, d_off, n_batches: int) -> int:
        n = ctypes.c_uint64(0)
        self._check(self.lib.nisshi_hip_count_dev(self.ctx, d_blob.data_ptr(), blob_len,
                                                  d_off.data_ptr(), n_batches, ctypes.byref(n),
                                                  self._stream()))
        return n.value

    def decode_dev(self, d_blob, blob_len: int, d_off, n_batches: int, cols: DeviceColumns,
                   by_value: bool = True, sync: bool = True, static: bool = False) -> Optional[Totals]:
        """`static`: through nisshi_hip_decode_dev_static, the entry point that allocates nothing."""
        totals = Totals()
        fn = self.lib.nisshi_hip_decode_dev_static if static else self.lib.nisshi_hip_decode_dev
        self._check(fn(
            self.ctx, d_blob.data_ptr(), blob_len, d_off.data_ptr(), n_batches, int(by_value),
            ctypes.byref(cols.c), cols.rec_cap, cols.hdr_cap,
            ctypes.byref(totals) if sync else None, self._stream()))
        return totals if sync else None

    def encode_dev(self, cols: DeviceColumns, n_batches: int, d_src, d_src_off, d_out, out_cap: int,
                   d_out_off, sync: bool = True) -> Optional[Totals]:
        totals = Totals()
        self._check(self.lib.nisshi_hip_encode_dev(
            self.ctx, ctypes.byref(cols.c), n_batches, d_src.data_ptr(), d_src_off.data_ptr(),
            d_out.data_ptr(), out_cap, d_out_off.data_ptr(),
            ctypes.byref(totals) if sync else None, self._stream()))
        return totals if sync else None

    def produce_rows_dev(self, cols: DeviceColumns, n_batches: int, n_slots: int, d_high,
                         check_time: bool = True) -> DeviceRows:
        """Rows of the SQL engines' produce seam from decoded columns (enqueues only)."""
        rows = DeviceRows(n_batches, n_slots, cols.hdr_cap, self.device)
        self._check(self.lib.nisshi_hip_produce_rows_dev(
            self.ctx, ctypes.byref(cols.c), n_batches, n_slots, cols.hdr_cap, d_high.data_ptr(),
            int(check_time), ctypes.byref(rows.c), self._stream()))
        return rows

    def combine_dev(self, cols: DeviceColumns, n_batches: int, n_slots: int, d_group_off,
                    n_groups: int) -> DeviceColumns:
        """Combine groups of decoded batches in place; returns the sinks' columns, which
        share the record and header arrays of `cols` (enqueues only)."""
        sink = DeviceColumns(n_groups, 0, 0, self.device)
        for name in ("rkey", "rval", "timestamp_delta", "rattributes", "hdr"):
            sink.t[name] = cols.t[name]
            setattr(sink.c, name, cols.t[name].data_ptr())
        sink.rec_cap, sink.hdr_cap = cols.rec_cap, cols.hdr_cap
        self._check(self.lib.nisshi_hip_combine_dev(
            self.ctx, ctypes.byref(cols.c), n_batches, n_slots, cols.hdr_cap, d_group_off.data_ptr(),
            n_groups, ctypes.byref(sink.c), self._stream()))
        return sink

    def fetch_rows_to_device(self, rows: dict):
        """Copy fetch rows (see fetch_rows_host) to the GPU; returns (struct, tensors kept alive)."""
        import torch
        keep, c = {}, CFetchRows()
        for name, dt, _ in FETCH_ARRAYS:
            v = rows.get(name)
            if v is None:
                setattr(c, name, None)
                continue
            a = np.ascontiguousarray(v, dtype=dt).reshape(-1)
            if a.size == 0:
                a = np.zeros(4, dtype=dt)
            keep[name] = torch.from_numpy(a.view(np.dtype(f"i{a.dtype.itemsize}")).copy()).to(self.device)
            setattr(c, name, keep[name].data_ptr())
        return c, keep

    def fetch_batches_dev(self, d_rows: "CFetchRows", d_hdr, n_rows: int, now: int, batch_cap: int,
                          sync: bool = True):
        """Batches of the SQL engines' fetch seam from rows on the GPU.  Returns
        (columns ready for encode_dev, number of batches or None)."""
        out = DeviceColumns(batch_cap, n_rows, 0, self.device)
        if d_hdr is not None:
            out.t["hdr"] = d_hdr
            out.c.hdr = d_hdr.data_ptr()
        n = ctypes.c_uint32(0)
        self._check(self.lib.nisshi_hip_fetch_batches_dev(
            self.ctx, ctypes.byref(d_rows), n_rows, now, ctypes.byref(out.c), batch_cap,
            ctypes.byref(n) if sync else None, self._stream()))
        return out, (n.value if sync else None)

    def builder_fields_dev(self, cols: DeviceColumns, n_batches: int, d_out, d_out_off):
        """batch_length and crc as inflated::Builder::build stores them, for batches
        already deflated into d_out (enqueues only)."""
        self._check(self.lib.nisshi_hip_builder_fields_dev(
            self.ctx, ctypes.byref(cols.c), n_batches, d_out.data_ptr(), d_out_off.data_ptr(), self._stream()))

    def fetch(self, rows: dict, payload: bytes, now: int, batch_cap: int = 64, out_cap: Optional[int] = None):
        """The fetch seam over host buffers: returns (list of deflated batches, totals)."""
        c, keep = fetch_rows_host(rows)
        n_rows = len(rows["offset"])
        arr = np.frombuffer(bytes(payload), dtype=np.uint8)
        n_hdr = int(keep["hdr_base"][n_rows]) if n_rows else 0
        if out_cap is None:
            out_cap = len(arr) + 61 * batch_cap + 64 * n_rows + 32 * n_hdr + 1024
        out = np.zeros(out_cap, dtype=np.uint8)
        out_off = np.zeros(batch_cap + 1, dtype=np.uint64)
        n, totals = ctypes.c_uint32(0), Totals()
        self._check(self.lib.nisshi_hip_fetch(
            self.ctx, ctypes.byref(c), n_rows, arr.ctypes.data if len(arr) else None, len(arr), now,
            out.ctypes.data, out_cap, out_off.ctypes.data, batch_cap, ctypes.byref(n), ctypes.byref(totals)))
        return [out[int(out_off[b]):int(out_off[b + 1])].tobytes() for b in range(n.value)], totals

    def compact_dev(self, d_blob, cols: DeviceColumns, n_batches: int, n_slots: int, d_group=None,
                    sync: bool = True):
        """Log compaction of decoded batches (oldest first).  Returns (compacted columns
        sharing the header array of `cols`, removed tensor, action tensor, totals or None)."""
        import torch
        out = DeviceColumns(n_batches, n_slots, 0, self.device)
        out.t["hdr"] = cols.t["hdr"]
        out.c.hdr = cols.t["hdr"].data_ptr()
        out.hdr_cap = cols.hdr_cap
        removed = torch.zeros(max(n_batches, 1), dtype=torch.int32, device=self.device)
        action = torch.zeros(max(n_batches, 1), dtype=torch.int32, device=self.device)
        totals = Totals()
        self._check(self.lib.nisshi_hip_compact_dev(
            self.ctx, d_blob.data_ptr(), ctypes.byref(cols.c), n_batches, n_slots,
            d_group.data_ptr() if d_group is not None else None, ctypes.byref(out.c), removed.data_ptr(),
            action.data_ptr(), ctypes.byref(totals) if sync else None, self._stream()))
        return out, removed, action, (totals if sync else None)

    def inflate_dev(self, d_blob, blob_len: int, d_batch_off, n_batches: int):
        """A frame with its gzip, Snappy and LZ4 batches inflated.  Asks for the room first, then allocates the output:
        returns (d_out tensor, d_out_off tensor, d_status tensor, totals)."""
        import torch
        d_out_off = torch.zeros(n_batches + 1, dtype=torch.int64, device=self.device)
        d_status = torch.zeros(max(n_batches, 1), dtype=torch.int32, device=self.device)
        room = Totals()
        self._check(self.lib.nisshi_hip_inflate_dev(
            self.ctx, d_blob.data_ptr(), blob_len, d_batch_off.data_ptr(), n_batches, None, 0,
            d_out_off.data_ptr(), d_status.data_ptr(), ctypes.byref(room), self._stream()))
        d_out = torch.empty(room.out_bytes + 16, dtype=torch.uint8, device=self.device)
        return self.inflate_into(d_blob, blob_len, d_batch_off, n_batches, d_out, room.out_bytes, d_out_off, d_status)

    def inflate_into(self, d_blob, blob_len, d_batch_off, n_batches, d_out, out_cap, d_out_off, d_status):
        """nisshi_hip_inflate_dev into buffers the caller holds."""
        totals = Totals()
        self._check(self.lib.nisshi_hip_inflate_dev(
            self.ctx, d_blob.data_ptr(), blob_len, d_batch_off.data_ptr(), n_batches, d_out.data_ptr(),
            out_cap, d_out_off.data_ptr(), d_status.data_ptr(), ctypes.byref(totals), self._stream()))
        return d_out, d_out_off, d_status, totals

    def inflate(self, blob: bytes, batch_off, out_cap: Optional[int] = None):
        """Host bytes in, host bytes out: (frame bytes, out_off, status np.int32[n], totals).
        out_cap None: the call is made twice, the first time to learn the size."""
        arr = np.frombuffer(bytes(blob), dtype=np.uint8)
        off = np.ascontiguousarray(batch_off, dtype=np.uint64)
        n_b = len(off) - 1
        out_off = np.zeros(n_b + 1, dtype=np.uint64)
        status = np.zeros(max(n_b, 1), dtype=np.int32)
        totals = Totals()
        cap = 0 if out_cap is None else out_cap
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        st = self.lib.nisshi_hip_inflate(self.ctx, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data,
                                         n_b, out.ctypes.data if cap else None, cap, out_off.ctypes.data,
                                         status.ctypes.data, ctypes.byref(totals))
        if st == C.CAPACITY and out_cap is None:
            cap = int(totals.out_bytes)
            out = np.zeros(max(cap, 1), dtype=np.uint8)
            st = self.lib.nisshi_hip_inflate(self.ctx, arr.ctypes.data if len(arr) else None, len(arr),
                                             off.ctypes.data, n_b, out.ctypes.data, cap, out_off.ctypes.data,
                                             status.ctypes.data, ctypes.byref(totals))
        self._check(st)
        return out[:int(totals.out_bytes)].tobytes(), out_off, status[:n_b], totals

    def deflate_dev(self, d_blob, blob_len: int, d_batch_off, n_batches: int, d_codec):
        """A frame of uncompressed batches with those whose codec is 3 (LZ4), 1 (gzip) or 4 (zstd) compressed.  Asks for the room first,
        then allocates the output: returns (d_out tensor, d_out_off tensor, d_status tensor, totals)."""
        import torch
        d_out_off = torch.zeros(n_batches + 1, dtype=torch.int64, device=self.device)
        d_status = torch.zeros(max(n_batches, 1), dtype=torch.int32, device=self.device)
        room = Totals()
        self._check(self.lib.nisshi_hip_deflate_dev(
            self.ctx, d_blob.data_ptr(), blob_len, d_batch_off.data_ptr(), n_batches, d_codec.data_ptr(), None, 0,
            d_out_off.data_ptr(), d_status.data_ptr(), ctypes.byref(room), self._stream()))
        d_out = torch.empty(room.out_bytes + 16, dtype=torch.uint8, device=self.device)
        totals = Totals()
        self._check(self.lib.nisshi_hip_deflate_dev(
            self.ctx, d_blob.data_ptr(), blob_len, d_batch_off.data_ptr(), n_batches, d_codec.data_ptr(), d_out.data_ptr(),
            room.out_bytes, d_out_off.data_ptr(), d_status.data_ptr(), ctypes.byref(totals), self._stream()))
        return d_out, d_out_off, d_status, totals

    def deflate(self, blob: bytes, batch_off, codec, out_cap: Optional[int] = None):
        """Host bytes in, host bytes out: (frame bytes, out_off, status np.int32[n], totals)."""
        arr = np.frombuffer(bytes(blob), dtype=np.uint8)
        off = np.ascontiguousarray(batch_off, dtype=np.uint64)
        n_b = len(off) - 1
        cod = np.ascontiguousarray(codec, dtype=np.uint8)
        assert len(cod) == n_b
        out_off = np.zeros(n_b + 1, dtype=np.uint64)
        status = np.zeros(max(n_b, 1), dtype=np.int32)
        totals = Totals()
        cap = len(arr) + 64 * n_b + 64 if out_cap is None else out_cap
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        st = self.lib.nisshi_hip_deflate(self.ctx, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data, n_b,
                                         cod.ctypes.data if n_b else None, out.ctypes.data, cap, out_off.ctypes.data,
                                         status.ctypes.data, ctypes.byref(totals))
        self._check(st)
        return out[:int(totals.out_bytes)].tobytes(), out_off, status[:n_b], totals

    def compact(self, blob: bytes, batch_off, group=None, out_cap: Optional[int] = None):
        """Inflate, compact and deflate batches held in host bytes (oldest first).  Returns
        (rewritten bytes or None per batch, removed, action, status, totals)."""
        arr = np.frombuffer(bytes(blob), dtype=np.uint8)
        off = np.ascontiguousarray(batch_off, dtype=np.uint64)
        n_b = len(off) - 1
        grp = None if group is None else np.ascontiguousarray(group, dtype=np.uint32)
        # a rebuilt length may take up to four bytes more than the stored one, on records of seven and more
        out_cap = len(arr) + len(arr) // 7 * 4 + 64 if out_cap is None else out_cap
        out = np.zeros(max(out_cap, 1), dtype=np.uint8)
        out_off = np.zeros(n_b + 1, dtype=np.uint64)
        removed = np.zeros(max(n_b, 1), dtype=np.uint32)
        action = np.zeros(max(n_b, 1), dtype=np.int32)
        status = np.zeros(max(n_b, 1), dtype=np.int32)
        totals = Totals()
        self._check(self.lib.nisshi_hip_compact(
            self.ctx, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data, n_b,
            grp.ctypes.data if grp is not None else None, out.ctypes.data, out_cap, out_off.ctypes.data,
            removed.ctypes.data, action.ctypes.data, status.ctypes.data, ctypes.byref(totals)))
        pieces = [out[int(out_off[b]):int(out_off[b + 1])].tobytes() if action[b] == 1 else None
                  for b in range(n_b)]
        return pieces, removed[:n_b], action[:n_b], status[:n_b], totals

    def crc32c_dev(self, d_data, length: int, offset: int = 0) -> int:
        out = ctypes.c_uint32(0)
        self._check(self.lib.nisshi_hip_crc32c_dev(self.ctx, d_data.data_ptr() + offset, length,
                                                   ctypes.byref(out), self._stream()))
        return out.value

    # ---- host-buffer path -----------------------------------------------------
    def pinned(self, shape, dtype) -> np.ndarray:
        """A numpy array in page-locked memory from nisshi_hip_host_alloc (freed with
        the codec): blobs and columns held in such arrays cross PCIe by DMA."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        ptr = ctypes.c_void_p()
        self._check(self.lib.nisshi_hip_host_alloc(self.ctx, n, ctypes.byref(ptr)))
        self._pinned.append(ptr)
        buf = (ctypes.c_uint8 * max(n, 1)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def pinned_columns(self, n_batches: int, rec_cap: int, hdr_cap: int) -> dict:
        """Host arrays in the device layout, page-locked."""
        counts = (n_batches, rec_cap, hdr_cap)
        out = {}
        for name, dt, words, level in C.DEVICE_ARRAYS:
            n = max(counts[level], 1)
            out[name] = self.pinned((n, words) if words > 1 else (n,), dt)
        return out

    def decode_into(self, blob: np.ndarray, batch_off: np.ndarray, arrays: dict, rec_cap: int,
                    hdr_cap: int, by_value: bool = True) -> "Totals":
        """Decode host bytes into caller-held arrays in the device layout (pinned or not)."""
        cc = C.as_c(arrays)
        totals = Totals()
        self._check(self.lib.nisshi_hip_decode(
            self.ctx, blob.ctypes.data if len(blob) else None, len(blob), batch_off.ctypes.data,
            len(batch_off) - 1, int(by_value), ctypes.byref(cc), rec_cap, hdr_cap, ctypes.byref(totals)))
        return totals

    @staticmethod
    def _record_slots(arr: np.ndarray, off: np.ndarray) -> int:
        """Record slots a decode of these batches can use: the sum of the record counts
        the headers claim, which the bytes bound (a record is at least seven bytes)."""
        starts = off[:-1].astype(np.int64)
        whole = starts[(off[1:].astype(np.int64) - starts) >= 61]
        claimed = 0
        if len(whole):
            b = arr[(whole[:, None] + np.arange(57, 61)).ravel()].reshape(-1, 4).astype(np.int64)
            claimed = int(((b[:, 0] << 24) | (b[:, 1] << 16) | (b[:, 2] << 8) | b[:, 3]).sum())
        return min(claimed, len(arr) // 7) + 1

    def decode(self, blob: bytes, batch_off, by_value: bool = True, rec_cap: Optional[int] = None,
               hdr_cap: Optional[int] = None):
        """Decode host bytes; returns (cols: dict of numpy arrays, totals).  Without
        caps the record columns are sized from the batch headers, and the header
        columns start at an eighth of the blob and grow while a batch reports
        NISSHI_CAPACITY (len(blob) + 1 slots cover every case: see
        nisshi_hip_decode_dev in include/nisshi_hip.h)."""
        arr = np.frombuffer(bytes(blob), dtype=np.uint8)
        off = np.ascontiguousarray(batch_off, dtype=np.uint64)
        n_b = len(off) - 1
        rec_cap = self._record_slots(arr, off) if rec_cap is None else rec_cap
        grow = hdr_cap is None
        most = len(arr) + 1
        hdr_cap = min(most, max(len(arr) // 8, 1024)) if grow else hdr_cap
        while True:
            arrays = C.alloc_device_layout(n_b, rec_cap, hdr_cap)
            cc = C.as_c(arrays)
            totals = Totals()
            self._check(self.lib.nisshi_hip_decode(
                self.ctx, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data, n_b,
                int(by_value), ctypes.byref(cc), rec_cap, hdr_cap, ctypes.byref(totals)))
            cols = C.unpack(arrays)
            short = grow and hdr_cap < most and totals.n_failed and (cols["status"][:n_b] == C.CAPACITY).any()
            if not short:
                break
            hdr_cap = min(most, max(2 * hdr_cap, int(totals.n_headers)))
        return C.trim(cols, n_b, min(totals.n_records, rec_cap), min(totals.n_headers, hdr_cap)), totals

    def encode(self, cols: dict, src: bytes, src_data_off, out_cap: Optional[int] = None):
        """Encode host columns; returns (bytes, out_off np.uint64[n + 1], totals)."""
        n_b = len(cols["status"])
        sarr = np.frombuffer(bytes(src), dtype=np.uint8)
        soff = np.ascontiguousarray(src_data_off, dtype=np.uint64)
        if out_cap is None:
            out_cap = 2 * len(sarr) + 128 * n_b + 64 * len(cols["length"]) + 1024
        out = np.zeros(out_cap, dtype=np.uint8)
        out_off = np.zeros(n_b + 1, dtype=np.uint64)
        full = C.pack(cols)
        full = {k: (v if len(v) else np.zeros((1,) + v.shape[1:], dtype=v.dtype)) for k, v in full.items()}
        cc = C.as_c(full)
        totals = Totals()
        self._check(self.lib.nisshi_hip_encode(
            self.ctx, ctypes.byref(cc), n_b, len(cols["length"]), len(cols["hkey_off"]),
            sarr.ctypes.data if len(sarr) else None, len(sarr),
            soff.ctypes.data if len(soff) else None, out.ctypes.data, out_cap,
            out_off.ctypes.data, ctypes.byref(totals)))
        return out[:totals.out_bytes].tobytes(), out_off, totals

    def produce_rows(self, blob: bytes, batch_off, high, check_time: bool = True,
                     rec_cap: Optional[int] = None, hdr_cap: Optional[int] = None):
        """Decode host bytes and form the produce rows; returns (cols, rows, totals)."""
        arr = np.frombuffer(bytes(blob), dtype=np.uint8)
        off = np.ascontiguousarray(batch_off, dtype=np.uint64)
        hi = np.ascontiguousarray(high, dtype=np.int64)
        n_b = len(off) - 1
        rec_cap = self._record_slots(arr, off) if rec_cap is None else rec_cap
        hdr_cap = len(arr) + 1 if hdr_cap is None else hdr_cap   # covers every case: see nisshi_hip_decode_dev
        arrays = C.alloc_device_layout(n_b, rec_cap, hdr_cap)
        cc = C.as_c(arrays)
        counts = (n_b, rec_cap, hdr_cap)
        rows = {name: np.zeros(max(counts[level], 1), dtype=dt) for name, dt, level in ROW_ARRAYS}
        cr = CRows()
        for name, _, _ in ROW_ARRAYS:
            setattr(cr, name, rows[name].ctypes.data)
        totals = Totals()
        self._check(self.lib.nisshi_hip_produce_rows(
            self.ctx, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data, n_b,
            hi.ctypes.data if len(hi) else None, int(check_time), ctypes.byref(cc), ctypes.byref(cr),
            rec_cap, hdr_cap, ctypes.byref(totals)))
        nr, nh = min(totals.n_records, rec_cap), min(totals.n_headers, hdr_cap)
        cols = C.trim(C.unpack(arrays), n_b, nr, nh)
        used = (n_b, nr, nh)
        return cols, {name: rows[name][:used[level]] for name, _, level in ROW_ARRAYS}, totals

    def combine(self, blob: bytes, batch_off, group_off, out_cap: Optional[int] = None):
        """Inflate, combine and deflate groups of batches held in host bytes.  Returns
        (list of bytes or None per group, group_status np.int32[n_groups], totals)."""
        arr = np.frombuffer(bytes(blob), dtype=np.uint8)
        off = np.ascontiguousarray(batch_off, dtype=np.uint64)
        groups = np.ascontiguousarray(group_off, dtype=np.uint32)
        n_b, n_g = len(off) - 1, len(groups) - 1
        # rebased deltas may need longer varints: at most 13 bytes more on a record of 7
        out_cap = 3 * len(arr) + 64 if out_cap is None else out_cap
        out = np.zeros(max(out_cap, 1), dtype=np.uint8)
        out_off = np.zeros(n_g + 1, dtype=np.uint64)
        status = np.zeros(max(n_g, 1), dtype=np.int32)
        totals = Totals()
        self._check(self.lib.nisshi_hip_combine(
            self.ctx, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data, n_b,
            groups.ctypes.data, n_g, out.ctypes.data, out_cap, out_off.ctypes.data,
            status.ctypes.data, ctypes.byref(totals)))
        pieces = [out[int(out_off[g]):int(out_off[g + 1])].tobytes() if status[g] == C.OK else None
                  for g in range(n_g)]
        return pieces, status[:n_g], totals


class Pool:
    """nisshi_hip_pool: a host frame's batches over several devices by partition id,
    columns back in input order.  `devices`: HIP device ids (an id may repeat), or,
    with `decoder`, the number of devices of a pool whose devices decode by calling
    `decoder(index, blob, batch_off, by_value, cols, rec_cap, hdr_cap)` -> (status, totals):
    the form the CPU tests use."""

    def __init__(self, devices, decoder=None):
        self.lib = load()
        pool = ctypes.c_void_p()
        if decoder is None:
            ids = (ctypes.c_int * len(devices))(*devices)
            st = self.lib.nisshi_hip_pool_create(ids, len(devices), ctypes.byref(pool))
            self.n = len(devices)
        else:
            self.n = int(devices)
            self._cb = POOL_DECODE_FN(decoder)
            ctxs = (ctypes.c_void_p * self.n)(*[i + 1 for i in range(self.n)])   # the device's index + 1 as its context
            self._ctxs = ctxs
            st = self.lib.nisshi_hip_pool_create_custom(self.n, self._cb, ctxs, ctypes.byref(pool))
        if st:
            raise NisshiError(st, "pool_create")
        self.pool = pool

    def close(self):
        if getattr(self, "pool", None):
            self.lib.nisshi_hip_pool_destroy(self.pool)
            self.pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_of(self, partition: int) -> int:
        return self.lib.nisshi_hip_pool_device_of(self.pool, partition)

    def last_times(self) -> dict:
        """Host-clock milliseconds of the last decode: the slowest device's gather,
        decode (copies and kernels) and scatter, and the whole call."""
        ms = (ctypes.c_double * 4)()
        self.lib.nisshi_hip_pool_last_times(self.pool, ms)
        return {"gather_ms": ms[0], "decode_ms": ms[1], "scatter_ms": ms[2], "call_ms": ms[3]}

    def decode_into(self, arr: np.ndarray, off: np.ndarray, part: np.ndarray, arrays: dict, rec_cap: int,
                    hdr_cap: int, by_value: bool = True):
        """Decode into caller-held arrays in the device layout: (totals, per-device totals)."""
        cc = C.as_c(arrays)
        totals = Totals()
        per_device = (Totals * self.n)()
        n_b = len(off) - 1
        st = self.lib.nisshi_hip_pool_decode(self.pool, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data,
                                             n_b, part.ctypes.data if n_b else None, int(by_value), ctypes.byref(cc),
                                             rec_cap, hdr_cap, ctypes.byref(totals), per_device)
        if st:
            raise NisshiError(st, (self.lib.nisshi_hip_pool_last_error(self.pool) or b"").decode())
        return totals, list(per_device)

    def decode(self, blob, batch_off, partition, rec_cap: int, hdr_cap: int, by_value: bool = True):
        """Returns (cols: dict of numpy arrays in the device layout's fields, totals, per-device totals)."""
        arr = np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else np.ascontiguousarray(blob)
        off = np.ascontiguousarray(batch_off, dtype=np.uint64)
        part = np.ascontiguousarray(partition, dtype=np.uint32)
        n_b = len(off) - 1
        arrays = C.alloc_device_layout(n_b, rec_cap, hdr_cap)
        cc = C.as_c(arrays)
        totals = Totals()
        per_device = (Totals * self.n)()
        st = self.lib.nisshi_hip_pool_decode(self.pool, arr.ctypes.data if len(arr) else None, len(arr), off.ctypes.data,
                                             n_b, part.ctypes.data if n_b else None, int(by_value), ctypes.byref(cc),
                                             rec_cap, hdr_cap, ctypes.byref(totals), per_device)
        if st:
            raise NisshiError(st, (self.lib.nisshi_hip_pool_last_error(self.pool) or b"").decode())
        cols = C.trim(C.unpack(arrays), n_b, min(totals.n_records, rec_cap), min(totals.n_headers, hdr_cap))
        return cols, totals, list(per_device)
