"""bitnet-rs_amd -- MI355X (gfx950) drop-in for the BitNet-rs I2_S / QK256 hot path.

The product is libbitnet_hip.so (hand-written HIP behind the C ABI in
include/bitnet_hip.h).  This module is the thin ctypes binding the tests and
bench.py drive it through; it contains no arithmetic and NO fallback: if the
library is missing or there is no GPU, calls raise.

The directory name carries a hyphen (as the project is named), so import it with
    importlib.import_module("bitnet-rs_amd")
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libbitnet_hip.so")
HEADER_PATH = os.path.join(ROOT, "include", "bitnet_hip.h")

OK = 0
ERR_INVALID_ARGUMENT, ERR_GPU, ERR_UNSUPPORTED, ERR_EXECUTION = -1, -2, -3, -4
KERNEL_AUTO, KERNEL_EXACT, KERNEL_VALU, KERNEL_MFMA = 0, 1, 2, 3
QTYPE_I2S, QTYPE_TL1, QTYPE_TL2 = 0, 1, 2

_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)
_f32p = C.POINTER(C.c_float)
_sz = C.c_size_t
_vp = C.c_void_p


class BitNetHipError(RuntimeError):
    """Mirrors BitNetError::Kernel(KernelError::*) (crates/bitnet-common/src/error.rs:80-97)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code
        self.kind = {
            ERR_INVALID_ARGUMENT: "InvalidArguments",
            ERR_GPU: "GpuError",
            ERR_UNSUPPORTED: "UnsupportedHardware",
            ERR_EXECUTION: "ExecutionFailed",
        }.get(code, "ExecutionFailed")


def _load_build_module():
    spec = importlib.util.spec_from_file_location("_bitnet_hip_build", os.path.join(HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libbitnet_hip.so (in-tree)."""
    return _load_build_module().build(force=force, verbose=verbose)


class DeviceInfo(C.Structure):
    _fields_ = [
        ("device_id", C.c_int32),
        ("name", C.c_char * 128),
        ("gcn_arch", C.c_char * 64),
        ("total_memory", C.c_uint64),
        ("compute_unit_count", C.c_int32),
        ("max_wavefront_size", C.c_int32),
        ("max_shared_memory_per_workgroup", C.c_uint64),
        ("supports_fp16", C.c_int32),
        ("supports_bf16", C.c_int32),
    ]


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(t):
    """Device pointer of a torch tensor (or a raw int)."""
    return _vp(t if isinstance(t, int) else t.data_ptr())


class HipLib:
    """ctypes view of libbitnet_hip.so.  One method per exported symbol."""

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with __graft_entry__.build() -- there is no fallback path"
            )
        self.path = path
        self.c = C.CDLL(path)
        L = self.c
        L.bitnet_hip_init.argtypes = [C.c_int]
        L.bitnet_hip_cleanup.restype = None
        L.bitnet_hip_get_last_error.restype = C.c_char_p
        L.bitnet_hip_get_device_info.argtypes = [C.c_int, C.POINTER(DeviceInfo)]
        L.bitnet_hip_set_kernel.argtypes = [C.c_int]
        L.bitnet_hip_gemv_qk256.argtypes = [_u8p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, _sz]
        L.bitnet_hip_i2s_matmul_f32.argtypes = [_f32p, _sz, _u8p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, _sz, _sz]
        L.bitnet_hip_qk256_gemv.argtypes = [_u8p, _sz, _f32p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, _sz]
        L.bitnet_hip_matmul_i2s.argtypes = [_i8p, _sz, _u8p, _sz, _f32p, _sz, _sz, _sz, _sz]
        L.bitnet_hip_quantize.argtypes = [_f32p, _sz, _u8p, _sz, _f32p, _sz, C.c_int]
        L.bitnet_hip_dequant_i2s.argtypes = [_u8p, _sz, _sz, _sz, C.c_int, C.c_float, C.c_int, _f32p, _sz]
        L.bitnet_hip_weights_upload_qk256.argtypes = [_u8p, _sz, _sz, _sz, _sz, C.POINTER(C.c_uint64)]
        L.bitnet_hip_weights_upload_i2s.argtypes = [_u8p, _sz, _f32p, _sz, _sz, _sz, _sz, C.POINTER(C.c_uint64)]
        L.bitnet_hip_weights_free.argtypes = [C.c_uint64]
        L.bitnet_hip_weights_info.argtypes = [C.c_uint64, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]
        L.bitnet_hip_gemv_dev.argtypes = [C.c_uint64, _vp, _vp, _vp]
        L.bitnet_hip_matmul_dev.argtypes = [C.c_uint64, _vp, _vp, _sz, _vp]

    # -- helpers ---------------------------------------------------------
    def last_error(self) -> str:
        e = self.c.bitnet_hip_get_last_error()
        return e.decode() if e else ""

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise BitNetHipError(rc, self.last_error() or f"bitnet_hip rc={rc}")

    # -- lifecycle -------------------------------------------------------
    def init(self, device: int = -1) -> None:
        self._check(self.c.bitnet_hip_init(device))

    def cleanup(self) -> None:
        self.c.bitnet_hip_cleanup()

    def is_available(self) -> bool:
        return bool(self.c.bitnet_hip_is_available())

    def device_count(self) -> int:
        return int(self.c.bitnet_hip_device_count())

    def device_info(self, device: int = 0) -> DeviceInfo:
        info = DeviceInfo()
        self._check(self.c.bitnet_hip_get_device_info(device, C.byref(info)))
        return info

    def set_kernel(self, kernel: int) -> None:
        self._check(self.c.bitnet_hip_set_kernel(kernel))

    def get_kernel(self) -> int:
        return int(self.c.bitnet_hip_get_kernel())

    # -- host-pointer drop-ins --------------------------------------------
    def gemv_qk256(self, qs, x, rows, cols, row_stride_bytes, y_len=None) -> np.ndarray:
        q, xa = _np(qs, np.uint8), _np(x, np.float32)
        y = np.zeros(rows if y_len is None else y_len, np.float32)
        self._check(
            self.c.bitnet_hip_gemv_qk256(
                q.ctypes.data_as(_u8p), q.size, xa.ctypes.data_as(_f32p), xa.size, y.ctypes.data_as(_f32p), y.size,
                rows, cols, row_stride_bytes,
            )
        )
        return y

    def i2s_matmul_f32(self, act, w, scales, m, n, k, block_size, out_len=None) -> np.ndarray:
        a, wq, s = _np(act, np.float32), _np(w, np.uint8), _np(scales, np.float32)
        out = np.zeros(m * n if out_len is None else out_len, np.float32)
        self._check(
            self.c.bitnet_hip_i2s_matmul_f32(
                a.ctypes.data_as(_f32p), a.size, wq.ctypes.data_as(_u8p), wq.size, s.ctypes.data_as(_f32p), s.size,
                out.ctypes.data_as(_f32p), out.size, m, n, k, block_size,
            )
        )
        return out

    def qk256_gemv(self, weights, scales, inp, m, n, k) -> np.ndarray:
        wq, s, a = _np(weights, np.uint8), _np(scales, np.float32), _np(inp, np.float32)
        out = np.zeros(m * n, np.float32)
        self._check(
            self.c.bitnet_hip_qk256_gemv(
                wq.ctypes.data_as(_u8p), wq.size, s.ctypes.data_as(_f32p), s.size, a.ctypes.data_as(_f32p), a.size,
                out.ctypes.data_as(_f32p), out.size, m, n, k,
            )
        )
        return out

    def matmul_i2s(self, a, b, m, n, k, c_len=None) -> np.ndarray:
        aa, bb = _np(a, np.int8), _np(b, np.uint8)
        c = np.zeros(m * n if c_len is None else c_len, np.float32)
        self._check(
            self.c.bitnet_hip_matmul_i2s(
                aa.ctypes.data_as(_i8p), aa.size, bb.ctypes.data_as(_u8p), bb.size, c.ctypes.data_as(_f32p), c.size, m, n, k
            )
        )
        return c

    def quantize(self, x, qtype=QTYPE_I2S, out_len=None, scales_len=None, out_init=None):
        xa = _np(x, np.float32)
        out = np.zeros(xa.size // 4 if out_len is None else out_len, np.uint8) if out_init is None else _np(out_init, np.uint8).copy()
        scales = np.zeros((xa.size + 31) // 32 if scales_len is None else scales_len, np.float32)
        self._check(
            self.c.bitnet_hip_quantize(
                xa.ctypes.data_as(_f32p), xa.size, out.ctypes.data_as(_u8p), out.size, scales.ctypes.data_as(_f32p), scales.size, qtype
            )
        )
        return out, scales

    def dequant_i2s(self, data, rows, cols, inv=False, k=1.0, transposed=False) -> np.ndarray:
        d = _np(data, np.uint8)
        out = np.zeros(rows * cols, np.float32)
        self._check(
            self.c.bitnet_hip_dequant_i2s(d.ctypes.data_as(_u8p), d.size, rows, cols, int(inv), k, int(transposed), out.ctypes.data_as(_f32p), out.size)
        )
        return out

    # -- device-resident API ------------------------------------------------
    def weights_upload_qk256(self, qs, rows, cols, row_stride_bytes) -> int:
        q = _np(qs, np.uint8)
        h = C.c_uint64(0)
        self._check(self.c.bitnet_hip_weights_upload_qk256(q.ctypes.data_as(_u8p), q.size, rows, cols, row_stride_bytes, C.byref(h)))
        return h.value

    def weights_upload_i2s(self, w, scales, n, k, block_size) -> int:
        wq, s = _np(w, np.uint8), _np(scales, np.float32)
        h = C.c_uint64(0)
        self._check(self.c.bitnet_hip_weights_upload_i2s(wq.ctypes.data_as(_u8p), wq.size, s.ctypes.data_as(_f32p), s.size, n, k, block_size, C.byref(h)))
        return h.value

    def weights_free(self, h: int) -> None:
        self._check(self.c.bitnet_hip_weights_free(h))

    def weights_info(self, h: int):
        r, c, b = _sz(0), _sz(0), _sz(0)
        self._check(self.c.bitnet_hip_weights_info(h, C.byref(r), C.byref(c), C.byref(b)))
        return r.value, c.value, b.value

    def gemv_dev(self, h: int, x, y, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_gemv_dev(h, _ptr(x), _ptr(y), _vp(stream)))

    def matmul_dev(self, h: int, x, y, m: int, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_matmul_dev(h, _ptr(x), _ptr(y), m, _vp(stream)))


_lib = None


def load() -> HipLib:
    """Load libbitnet_hip.so (building nothing: call build() first)."""
    global _lib
    if _lib is None:
        _lib = HipLib()
    return _lib


def declared_symbols() -> list[str]:
    """Every function include/bitnet_hip.h declares (used by the export test)."""
    import re

    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bitnet_hip_[a-z0-9_]+)\s*\(", text)))
