"""Minimal ctypes view of the HIP runtime: device buffers, events and streams for hosts that do not
want PyTorch in the process (tests, bench.py at N=1, rocprofv3 runs).

It binds the SAME libamdhip64 the HIP library links against (the loader returns the already
mapped object), so pointers, streams and events are interchangeable with liblantern_gpu.so.
When PyTorch is used in the same process, `import torch` BEFORE importing lantern_amd.capi:
torch bundles its own libamdhip64 with the same SONAME and two HIP runtimes cannot share a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

_hip = None
H2D, D2H, D2D = 1, 2, 3


class HipError(RuntimeError):
    pass


def rt() -> C.CDLL:
    global _hip
    if _hip is None:
        from . import capi

        capi.lib()  # make sure the runtime our library uses is the one that gets bound
        last = None
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError as e:  # pragma: no cover
                last = e
        if _hip is None:
            raise HipError(f"cannot load the HIP runtime: {last}")
        _hip.hipGetErrorString.restype = C.c_char_p
    return _hip


def check(code: int, what: str = "HIP call"):
    if code != 0:
        raise HipError(f"{what} failed: {rt().hipGetErrorString(code).decode()}")


def set_device(i: int):
    check(rt().hipSetDevice(C.c_int(i)), "hipSetDevice")


def device_bus_id(i: int) -> str:
    """PCI bus id of device `i` ("0000:05:00.0"): what tells two ranks apart that were handed the same GPU."""
    buf = C.create_string_buffer(64)
    check(rt().hipDeviceGetPCIBusId(buf, C.c_int(64), C.c_int(i)), "hipDeviceGetPCIBusId")
    return buf.value.decode()


def synchronize():
    check(rt().hipDeviceSynchronize(), "hipDeviceSynchronize")


class Buffer:
    """A device allocation."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(rt().hipMalloc(C.byref(p), C.c_size_t(max(self.nbytes, 16))), "hipMalloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "Buffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.upload(a)
        return b

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(rt().hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.c_int(H2D)), "hipMemcpy H2D")

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(rt().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), C.c_int(D2H)), "hipMemcpy D2H")
        return out

    def zero(self):
        check(rt().hipMemset(C.c_void_p(self.ptr), C.c_int(0), C.c_size_t(self.nbytes)), "hipMemset")

    def free(self):
        if getattr(self, "ptr", None):
            rt().hipFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    def __init__(self):
        s = C.c_void_p()
        check(rt().hipStreamCreate(C.byref(s)), "hipStreamCreate")
        self.handle = s.value

    def synchronize(self):
        check(rt().hipStreamSynchronize(C.c_void_p(self.handle)), "hipStreamSynchronize")


class Event:
    def __init__(self):
        e = C.c_void_p()
        check(rt().hipEventCreate(C.byref(e)), "hipEventCreate")
        self.handle = e.value

    def record(self, stream_handle=None):
        check(rt().hipEventRecord(C.c_void_p(self.handle), C.c_void_p(stream_handle)), "hipEventRecord")

    def synchronize(self):
        check(rt().hipEventSynchronize(C.c_void_p(self.handle)), "hipEventSynchronize")

    def elapsed_ms(self, end: "Event") -> float:
        ms = C.c_float()
        check(rt().hipEventElapsedTime(C.byref(ms), C.c_void_p(self.handle), C.c_void_p(end.handle)), "hipEventElapsedTime")
        return float(ms.value)


def padded_rows(x: np.ndarray, metric_is_hamming: bool, f16: bool = False, i8: bool = False, b1: bool = False, row_bytes: int | None = None) -> np.ndarray:
    out = _padded_rows(x, metric_is_hamming, f16, i8, b1)
    if row_bytes is not None and out.shape[1] * out.itemsize < row_bytes:  # the index stores its rows at a wider stride (GpuIndex.row_bytes())
        wide = np.zeros((out.shape[0], row_bytes // out.itemsize), dtype=out.dtype)
        wide[:, : out.shape[1]] = out
        out = wide
    return out


def _padded_rows(x: np.ndarray, metric_is_hamming: bool, f16: bool = False, i8: bool = False, b1: bool = False) -> np.ndarray:
    """Rows in STORAGE format, zero-padded to whole 16-byte chunks: the layout the device entry points expect
    (u32 words for hamming, f32, halves for an f16 index: the cast is round-to-nearest-even, or bytes for an i8
    index: trunc(clamp(x * 100, -100, 100)), the library's own rule for f32 input)."""
    if b1:  # quant_bits = 1: bit i = (x_i > 0), most significant bit of each byte first, rows padded to 16 bytes
        a = np.ascontiguousarray(x, dtype=np.float32)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        bits = (a > 0).astype(np.uint8)
        pad = (-bits.shape[1]) % 128
        if pad:
            bits = np.concatenate([bits, np.zeros((bits.shape[0], pad), np.uint8)], axis=1)
        return np.packbits(bits, axis=1, bitorder="big")
    if i8:
        a = np.ascontiguousarray(x, dtype=np.float32)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        v = a * np.float32(100.0)
        v = np.where(np.isnan(v), np.float32(0), v)
        q = np.trunc(np.clip(v, np.float32(-100.0), np.float32(100.0))).astype(np.int8)
        out = np.zeros((q.shape[0], (q.shape[1] + 15) // 16 * 16), dtype=np.int8)
        out[:, : q.shape[1]] = q
        return out
    if f16:
        a = np.ascontiguousarray(x, dtype=np.float32).astype(np.float16)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        w = (a.shape[1] + 7) // 8 * 8
        if w == a.shape[1]:
            return a
        out = np.zeros((a.shape[0], w), dtype=np.float16)
        out[:, : a.shape[1]] = a
        return out
    a = np.ascontiguousarray(x, dtype=np.uint32 if metric_is_hamming else np.float32)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    w = (a.shape[1] + 3) // 4 * 4
    if w == a.shape[1]:
        return a
    out = np.zeros((a.shape[0], w), dtype=a.dtype)
    out[:, : a.shape[1]] = a
    return out
