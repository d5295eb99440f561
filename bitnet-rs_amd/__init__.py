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
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libbitnet_hip.so")
HEADER_PATH = os.path.join(ROOT, "include", "bitnet_hip.h")

OK = 0
ERR_INVALID_ARGUMENT, ERR_GPU, ERR_UNSUPPORTED, ERR_EXECUTION = -1, -2, -3, -4
KERNEL_AUTO, KERNEL_EXACT, KERNEL_VALU, KERNEL_MFMA, KERNEL_MFMA_TILED = 0, 1, 2, 3, 4
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
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(a, dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(t):
    """Device pointer of a torch tensor (or a raw int)."""
    return _vp(t if isinstance(t, int) else t.data_ptr())


def _optr(t):
    """Nullable device pointer."""
    return None if t is None else _ptr(t)


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
        L.bitnet_hip_attention.argtypes = [_f32p, _sz, _f32p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, _sz, C.c_int, C.c_float]
        L.bitnet_hip_qk256_gemv_batch.argtypes = [C.c_void_p, _sz]
        L.bitnet_hip_weights_upload_qk256.argtypes = [_u8p, _sz, _sz, _sz, _sz, C.POINTER(C.c_uint64)]
        L.bitnet_hip_weights_upload_i2s.argtypes = [_u8p, _sz, _f32p, _sz, _sz, _sz, _sz, C.POINTER(C.c_uint64)]
        L.bitnet_hip_weights_upload_coded.argtypes = [_u8p, _sz, _f32p, _sz, _sz, _sz, _sz, _i8p, C.POINTER(C.c_uint64)]
        L.bitnet_hip_weights_upload_inline_f16.argtypes = [_u8p, _sz, _sz, _sz, _i8p, C.c_int, C.POINTER(C.c_uint64)]
        L.bitnet_hip_weights_free.argtypes = [C.c_uint64]
        L.bitnet_hip_weights_info.argtypes = [C.c_uint64, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]
        L.bitnet_hip_gemv_dev.argtypes = [C.c_uint64, _vp, _vp, _vp]
        L.bitnet_hip_matmul_dev.argtypes = [C.c_uint64, _vp, _vp, _sz, _vp]
        L.bitnet_hip_matmul_kernel_dev.argtypes = [C.c_uint64, _vp, _vp, _sz, C.c_int, _vp]
        L.bitnet_hip_add_dev.argtypes = [_vp, _vp, _vp, _sz, _vp]
        L.bitnet_hip_silu_mul_dev.argtypes = [_vp, _vp, _vp, _sz, _sz, _vp]
        L.bitnet_hip_weights_device_bytes.argtypes = [C.c_uint64]
        L.bitnet_hip_weights_device_bytes.restype = _sz
        L.bitnet_hip_weights_trim.argtypes = [C.c_uint64]
        L.bitnet_hip_qact_bytes.argtypes = [_sz]
        L.bitnet_hip_qact_bytes.restype = _sz
        L.bitnet_hip_qact_stats_bytes.argtypes = [_sz]
        L.bitnet_hip_qact_stats_bytes.restype = _sz
        L.bitnet_hip_quantize_act_dev.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
        L.bitnet_hip_embed_q_dev.argtypes = [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp]
        L.bitnet_hip_gemv_q_supported.argtypes = [C.c_uint64]
        L.bitnet_hip_gemv_q_dev.argtypes = [C.c_uint64, _vp, _vp, _vp, C.c_float, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
        L.bitnet_hip_attention_decode_q_dev.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, C.c_int, _vp, _vp, _vp]
        L.bitnet_hip_gemv_attn_merge_q_dev.argtypes = [C.c_uint64, _vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.bitnet_hip_matmul_workspace_bytes.argtypes = [_sz, _sz, C.c_int]
        L.bitnet_hip_matmul_workspace_bytes.restype = _sz
        L.bitnet_hip_matmul_fused_dev.argtypes = [C.c_uint64, _vp, _vp, _sz, _vp, C.c_float, _vp, C.c_int, C.c_int, _vp, _sz, _vp]
        L.bitnet_hip_matmul_last_tile.argtypes = [C.POINTER(C.c_int)] * 4
        L.bitnet_hip_matmul_last_wave_rows.argtypes = []
        L.bitnet_hip_weights_bind_ln.argtypes = [C.c_uint64, _vp, _vp]
        L.bitnet_hip_weights_concat.argtypes = [C.POINTER(C.c_uint64), _sz, C.c_int, C.POINTER(C.c_uint64)]
        L.bitnet_hip_gemv_fused_dev.argtypes = [C.c_uint64, _vp, _vp, _sz, _vp, C.c_float, _vp, C.c_int, _vp]
        L.bitnet_hip_rmsnorm.argtypes = [_f32p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, C.c_float]
        L.bitnet_hip_norm_rows_dev.argtypes = [_vp, _vp, _vp, _sz, _sz, C.c_float, C.c_int, _vp]
        L.bitnet_hip_embed_f16_dev.argtypes = [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]
        L.bitnet_hip_advance_pos_dev.argtypes = [_vp, _vp]
        L.bitnet_hip_attention_decode_dev.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp, _vp]
        L.bitnet_hip_attention_decode_wide_dev.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp, _vp]
        L.bitnet_hip_attention_decode_partial_dev.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]
        L.bitnet_hip_gemv_attn_merge_dev.argtypes = [C.c_uint64, _vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp]
        L.bitnet_hip_attention_prefill_workspace_bytes.argtypes = [_sz, _sz, _sz]
        L.bitnet_hip_attention_prefill_workspace_bytes.restype = _sz
        L.bitnet_hip_attention_prefill_dev.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp, _sz, _vp, _vp]
        L.bitnet_hip_attention_prefill_sharded_workspace_bytes.argtypes = [_sz, _sz, _sz, _sz]
        L.bitnet_hip_attention_prefill_sharded_workspace_bytes.restype = _sz
        L.bitnet_hip_attention_prefill_sharded_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _sz, _vp, _vp]
        L.bitnet_hip_attention_scratch_bytes.argtypes = [_sz, _sz]
        L.bitnet_hip_attention_scratch_bytes.restype = _sz
        L.bitnet_hip_logits_f16_dev.argtypes = [_vp, _vp, _vp, C.c_float, _sz, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]
        L.bitnet_hip_argmax_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _vp]
        L.bitnet_hip_hbm_read_ceiling.argtypes = [_sz, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _vp]

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

    def quantized_matmul_i2s(self, x, packed, scales, block_size: int, m: int, n: int, k: int) -> np.ndarray:
        xa, pa, sa = _np(x, np.float32), _np(packed, np.uint8), _np(scales, np.float32)
        out = np.zeros(m * n, np.float32)
        self.c.bitnet_hip_quantized_matmul_i2s.argtypes = [_f32p, _sz, _u8p, _sz, _f32p, _sz, _sz, _f32p, _sz, _sz, _sz, _sz]
        self._check(self.c.bitnet_hip_quantized_matmul_i2s(xa.ctypes.data_as(_f32p), xa.size, pa.ctypes.data_as(_u8p), pa.size, sa.ctypes.data_as(_f32p), sa.size,
                                                           block_size, out.ctypes.data_as(_f32p), out.size, m, n, k))
        return out

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

    def attention(self, q, k, v, seq_len: int, num_heads: int, head_dim: int, causal: bool = True, scale: float | None = None) -> np.ndarray:
        qa, ka, va = _np(q, np.float32).reshape(-1), _np(k, np.float32).reshape(-1), _np(v, np.float32).reshape(-1)
        out = np.zeros(qa.size, np.float32)
        sc = float(scale) if scale is not None else 1.0 / float(np.sqrt(head_dim))
        self._check(self.c.bitnet_hip_attention(qa.ctypes.data_as(_f32p), qa.size, ka.ctypes.data_as(_f32p), ka.size, va.ctypes.data_as(_f32p), va.size,
                                                out.ctypes.data_as(_f32p), out.size, seq_len, num_heads, head_dim, int(causal), sc))
        return out

    def qk256_gemv_batch(self, items) -> list:
        """items: (weights u8, scales f32, input f32, m, n, k) tuples -> list of outputs [m*n]."""
        class Item(C.Structure):
            _fields_ = [("weights", _u8p), ("weights_len", _sz), ("scales", _f32p), ("scales_len", _sz), ("input", _f32p), ("input_len", _sz),
                        ("output", _f32p), ("output_len", _sz), ("m", _sz), ("n", _sz), ("k", _sz)]
        keep, arr = [], (Item * len(items))()
        for i, (w, s, x, m, n, k) in enumerate(items):
            wa, sa, xa, ya = _np(w, np.uint8), _np(s, np.float32), _np(x, np.float32), np.zeros(m * n, np.float32)
            keep.append((wa, sa, xa, ya))
            arr[i] = Item(wa.ctypes.data_as(_u8p), wa.size, sa.ctypes.data_as(_f32p), sa.size, xa.ctypes.data_as(_f32p), xa.size,
                          ya.ctypes.data_as(_f32p), ya.size, m, n, k)
        self._check(self.c.bitnet_hip_qk256_gemv_batch(C.cast(arr, C.c_void_p), len(items)))
        return [kp[3] for kp in keep]

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

    def weights_upload_coded(self, w, scales, n, k, block_size, code_map) -> int:
        wq, s, cm = _np(w, np.uint8), _np(scales, np.float32), _np(code_map, np.int8)
        h = C.c_uint64(0)
        self._check(self.c.bitnet_hip_weights_upload_coded(wq.ctypes.data_as(_u8p), wq.size, s.ctypes.data_as(_f32p), s.size, n, k, block_size, cm.ctypes.data_as(_i8p), C.byref(h)))
        return h.value

    def weights_upload_inline_f16(self, blocks, n, k, code_map, scale_mode: int = 0) -> int:
        b, cm = _np(blocks, np.uint8), _np(code_map, np.int8)
        h = C.c_uint64(0)
        self._check(self.c.bitnet_hip_weights_upload_inline_f16(b.ctypes.data_as(_u8p), b.size, n, k, cm.ctypes.data_as(_i8p), scale_mode, C.byref(h)))
        return h.value

    def matmul_workspace_bytes(self, m: int, k: int, digits: int = 4) -> int:
        return int(self.c.bitnet_hip_matmul_workspace_bytes(m, k, digits))

    def matmul_fused_dev(self, h: int, x, y, m: int, workspace, workspace_bytes: int, ln_gamma=None, ln_eps: float = 0.0, residual=None,
                         flags: int = 0, digits: int = 4, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_matmul_fused_dev(h, _ptr(x), _ptr(y), m, _ptr(ln_gamma) if ln_gamma is not None else None, ln_eps,
                                                       _ptr(residual) if residual is not None else None, flags, digits, _ptr(workspace),
                                                       workspace_bytes, _vp(stream)))

    # -- the prompt forward's f16 activation chain (include/bitnet_hip.h) --------------------------------------------------------
    def matmul_f16_supported(self, h: int) -> bool:
        self.c.bitnet_hip_matmul_f16_supported.argtypes = [C.c_uint64]
        return bool(self.c.bitnet_hip_matmul_f16_supported(h))

    def rows_to_f16_dev(self, x, gamma, m: int, cols: int, xh, stats, stream: int = 0) -> None:
        self.c.bitnet_hip_rows_to_f16_dev.argtypes = [_vp, _vp, _sz, _sz, _vp, _vp, _vp]
        self._check(self.c.bitnet_hip_rows_to_f16_dev(_ptr(x), _optr(gamma), m, cols, _ptr(xh), _optr(stats), _vp(stream)))

    def matmul_f16_dev(self, h: int, xh, m: int, stats_in=None, n_stats: int = 0, ln_gamma=None, ln_eps: float = 0.0, y=None, residual=None,
                       flags: int = 0, yh=None, gamma_out=None, stats_out=None, stream: int = 0) -> None:
        self.c.bitnet_hip_matmul_f16_dev.argtypes = [C.c_uint64, _vp, _sz, _vp, _sz, _vp, C.c_float, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]
        self._check(self.c.bitnet_hip_matmul_f16_dev(h, _ptr(xh), m, _optr(stats_in), n_stats, _optr(ln_gamma), ln_eps, _optr(y), _optr(residual), flags,
                                                     _optr(yh), _optr(gamma_out), _optr(stats_out), _vp(stream)))

    def f16_saturations(self, reset: bool = True) -> int:
        self.c.bitnet_hip_f16_saturations.restype = C.c_uint64
        self.c.bitnet_hip_f16_saturations.argtypes = [C.c_int]
        return int(self.c.bitnet_hip_f16_saturations(1 if reset else 0))

    # ---- QB32: producer-quantised rows for the fp6 x fp4 prompt matmul (include/bitnet_hip.h) ----
    def qb32_bytes(self, m: int, cols: int) -> int:
        self.c.bitnet_hip_qb32_bytes.restype = _sz
        self.c.bitnet_hip_qb32_bytes.argtypes = [_sz, _sz]
        return int(self.c.bitnet_hip_qb32_bytes(m, cols))

    def rows_to_qb32_dev(self, x, gamma, m: int, cols: int, qb, stats, stream: int = 0) -> None:
        self.c.bitnet_hip_rows_to_qb32_dev.argtypes = [_vp, _vp, _sz, _sz, _vp, _vp, _vp]
        self._check(self.c.bitnet_hip_rows_to_qb32_dev(_ptr(x), _optr(gamma), m, cols, _ptr(qb), _optr(stats), _vp(stream)))

    def matmul_qb32_supported(self, h: int) -> bool:
        self.c.bitnet_hip_matmul_qb32_supported.argtypes = [C.c_uint64]
        return bool(self.c.bitnet_hip_matmul_qb32_supported(h))

    def matmul_qb32_dev(self, h: int, qb, m: int, stats_in=None, n_stats: int = 0, ln_gamma=None, ln_eps: float = 0.0, y=None, residual=None,
                        flags: int = 0, yh=None, gamma_out=None, stats_out=None, stream: int = 0) -> None:
        self.c.bitnet_hip_matmul_qb32_dev.argtypes = [C.c_uint64, _vp, _sz, _vp, _sz, _vp, C.c_float, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]
        self._check(self.c.bitnet_hip_matmul_qb32_dev(h, _ptr(qb), m, _optr(stats_in), n_stats, _optr(ln_gamma), ln_eps, _optr(y), _optr(residual), flags,
                                                      _optr(yh), _optr(gamma_out), _optr(stats_out), _vp(stream)))

    def attention_prefill_flags_dev(self, qkv, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv, head_dim, max_pos, seq_len, workspace, workspace_bytes,
                                    out, flags: int, stream: int = 0) -> None:
        self.c.bitnet_hip_attention_prefill_flags_dev.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp, _sz, _vp, C.c_int, _vp]
        self._check(self.c.bitnet_hip_attention_prefill_flags_dev(_ptr(qkv), _ptr(rope_sin), _ptr(rope_cos), _ptr(kcache), _ptr(vcache), n_heads, n_kv,
                                                                  head_dim, max_pos, seq_len, _ptr(workspace), workspace_bytes, _ptr(out), flags, _vp(stream)))

    def matmul_last_tile(self) -> dict:
        """The tile form this thread's last tiled matmul ran: {digits, wave_tokens, waves, scale_mode}."""
        v = [C.c_int() for _ in range(4)]
        self._check(self.c.bitnet_hip_matmul_last_tile(*[C.byref(x) for x in v]))
        return dict(zip(("digits", "wave_tokens", "waves", "scale_mode"), (x.value for x in v)))

    def matmul_last_wave_rows(self) -> int:
        """Output rows per wave of this thread's last tiled matmul: 64, or 80 (320-row workgroups)."""
        return int(self.c.bitnet_hip_matmul_last_wave_rows())

    def matmul_last_resident_fp4(self) -> bool:
        self.c.bitnet_hip_matmul_last_resident_fp4.restype = C.c_int
        return bool(self.c.bitnet_hip_matmul_last_resident_fp4())

    def attention_prefill_workspace_bytes(self, n_heads: int, n_kv: int, seq_len: int) -> int:
        return int(self.c.bitnet_hip_attention_prefill_workspace_bytes(n_heads, n_kv, seq_len))

    def attention_prefill_dev(self, qkv, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv, head_dim, max_pos, seq_len, workspace,
                              workspace_bytes, out, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_attention_prefill_dev(_ptr(qkv), _ptr(rope_sin), _ptr(rope_cos), _ptr(kcache), _ptr(vcache), n_heads, n_kv,
                                                            head_dim, max_pos, seq_len, _ptr(workspace), workspace_bytes, _ptr(out), _vp(stream)))

    def attention_prefill_sharded_workspace_bytes(self, n_heads: int, n_kv: int, n_q: int, n_ctx: int) -> int:
        return int(self.c.bitnet_hip_attention_prefill_sharded_workspace_bytes(n_heads, n_kv, n_q, n_ctx))

    def attention_prefill_sharded_dev(self, q, ld_q, q_block_pos, n_q, kv, ld_kv, n_ctx, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv,
                                      head_dim, max_pos, workspace, workspace_bytes, out, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_attention_prefill_sharded_dev(_ptr(q), ld_q, _ptr(q_block_pos), n_q, _ptr(kv), ld_kv, n_ctx, _ptr(rope_sin),
                                                                    _ptr(rope_cos), _ptr(kcache), _ptr(vcache), n_heads, n_kv, head_dim, max_pos,
                                                                    _ptr(workspace), workspace_bytes, _ptr(out), _vp(stream)))

    def attention_prefill_gathered_dev(self, q, ld_q, q_block_pos, n_q, kv_gathered, n_ctx, world, kv_is_f16, rope_sin, rope_cos, kcache, vcache,
                                       cache_f16, n_heads, n_kv, head_dim, max_pos, workspace, workspace_bytes, out, stream: int = 0) -> None:
        """the k|v rows exactly as an all-gather over `world` ranks left them (rank-major zigzag chunks, f32 or f16)"""
        self.c.bitnet_hip_attention_prefill_gathered_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz, _sz, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _sz, _sz,
                                                                     _sz, _sz, _vp, _sz, _vp, _vp]
        self._check(self.c.bitnet_hip_attention_prefill_gathered_dev(_ptr(q), ld_q, _ptr(q_block_pos), n_q, _ptr(kv_gathered), n_ctx, world,
                                                                     int(kv_is_f16), _ptr(rope_sin), _ptr(rope_cos), _ptr(kcache), _ptr(vcache),
                                                                     int(cache_f16), n_heads, n_kv, head_dim, max_pos, _ptr(workspace), workspace_bytes,
                                                                     _ptr(out), _vp(stream)))

    def weights_bind_ln(self, h: int, ln_gamma, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_weights_bind_ln(h, _ptr(ln_gamma), _vp(stream)))

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

    def matmul_kernel_dev(self, h: int, x, y, m: int, kernel: int, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_matmul_kernel_dev(h, _ptr(x), _ptr(y), m, kernel, _vp(stream)))

    def add_dev(self, a, b, out, n: int, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_add_dev(_ptr(a), _ptr(b), _ptr(out), n, _vp(stream)))

    def silu_mul_dev(self, gate, up, out, n: int, tile: int = 0, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_silu_mul_dev(_ptr(gate), _ptr(up), _ptr(out), n, tile, _vp(stream)))

    def weights_device_bytes(self, h: int) -> int:
        return int(self.c.bitnet_hip_weights_device_bytes(h))

    def weights_fp4_image(self, h: int, enable: bool = True, stream: int = 0) -> None:
        """Build (or free) the resident fp4 image the fp6 x fp4 prompt matmul reads (include/bitnet_hip.h)."""
        self.c.bitnet_hip_weights_fp4_image.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
        self._check(self.c.bitnet_hip_weights_fp4_image(h, 1 if enable else 0, stream))

    def weights_trim(self, h: int) -> None:
        self._check(self.c.bitnet_hip_weights_trim(h))

    # ---- QAct: activations quantised by their producer (csrc/qact.hpp) ----
    def qact_bytes(self, cols: int) -> int:
        return int(self.c.bitnet_hip_qact_bytes(cols))

    def qact_stats_bytes(self, cols: int) -> int:
        return int(self.c.bitnet_hip_qact_stats_bytes(cols))

    def quantize_act_dev(self, x, gamma, cols: int, qact_out, stats_out=None, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_quantize_act_dev(_ptr(x), _optr(gamma), cols, _ptr(qact_out), _optr(stats_out), _vp(stream)))

    def embed_q_dev(self, table, tokens, x_out, hidden: int, vocab: int, gamma, qact_out, stats_out=None, offset=None, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_embed_q_dev(_ptr(table), _ptr(tokens), _optr(offset), hidden, vocab, _ptr(x_out), _optr(gamma), _ptr(qact_out),
                                                  _optr(stats_out), _vp(stream)))

    def gemv_q_supported(self, h: int) -> bool:
        return bool(self.c.bitnet_hip_gemv_q_supported(h))

    def gemv_q_dev(self, h: int, qact_in, y=None, stats_in=None, ln_gamma=None, ln_eps: float = 0.0, residual=None, flags: int = 0, qact_out=None,
                   gamma_out=None, stats_out=None, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_gemv_q_dev(h, _ptr(qact_in), _optr(stats_in), _optr(ln_gamma), ln_eps, _optr(residual), flags, _optr(y),
                                                 _optr(qact_out), _optr(gamma_out), _optr(stats_out), _vp(stream)))

    def attention_decode_q_dev(self, qkv, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv, head_dim, max_pos, pos, scratch, out, qact_out,
                               wide: bool = False, kv_f16: bool = False, stream: int = 0, partial: bool = False) -> None:
        flags = (1 if wide else 0) | (2 if kv_f16 else 0) | (4 if partial else 0)  # BITNET_HIP_ATTN_WIDE / _KV_F16 / _PARTIAL
        self._check(self.c.bitnet_hip_attention_decode_q_dev(_ptr(qkv), _ptr(rope_sin), _ptr(rope_cos), _ptr(kcache), _ptr(vcache), n_heads, n_kv,
                                                             head_dim, max_pos, _ptr(pos), _ptr(scratch), flags, _optr(out), _optr(qact_out), _vp(stream)))

    def gemv_attn_merge_q_dev(self, h: int, scratch, n_heads, n_kv, max_pos, pos, y, qact_out, residual=None, gamma_out=None, stats_out=None,
                              stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_gemv_attn_merge_q_dev(h, _ptr(scratch), n_heads, n_kv, max_pos, _ptr(pos), _ptr(y), _optr(residual),
                                                            _ptr(qact_out), _optr(gamma_out), _optr(stats_out), _vp(stream)))

    def weights_concat(self, parts, interleave16: bool = False) -> int:
        arr = (C.c_uint64 * len(parts))(*parts)
        h = C.c_uint64(0)
        self._check(self.c.bitnet_hip_weights_concat(arr, len(parts), int(interleave16), C.byref(h)))
        return h.value

    def gemv_fused_dev(self, h: int, x, y, m: int = 1, ln_gamma=None, ln_eps: float = 0.0, residual=None, flags: int = 0, stream: int = 0) -> None:
        self._check(
            self.c.bitnet_hip_gemv_fused_dev(h, _ptr(x), _ptr(y), m, _ptr(ln_gamma) if ln_gamma is not None else None, ln_eps,
                                             _ptr(residual) if residual is not None else None, flags, _vp(stream))
        )

    # -- decode-step operators ------------------------------------------------
    def rmsnorm(self, x, gamma, num_rows: int, hidden: int, eps: float = 1e-6) -> np.ndarray:
        xa, ga = _np(x, np.float32), _np(gamma, np.float32)
        out = np.zeros(num_rows * hidden, np.float32)
        self._check(self.c.bitnet_hip_rmsnorm(xa.ctypes.data_as(_f32p), xa.size, ga.ctypes.data_as(_f32p), ga.size, out.ctypes.data_as(_f32p), out.size, num_rows, hidden, eps))
        return out

    def norm_rows_dev(self, x, gamma, out, rows: int, hidden: int, eps: float, rms: bool, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_norm_rows_dev(_ptr(x), _ptr(gamma), _ptr(out), rows, hidden, eps, int(rms), _vp(stream)))

    def embed_f16_dev(self, table, tokens, out, n: int, hidden: int, vocab: int, offset=None, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_embed_f16_dev(_ptr(table), _ptr(tokens), _ptr(offset) if offset is not None else None, n, hidden, vocab, _ptr(out), _vp(stream)))

    def attention_decode_dev(self, qkv, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv, head_dim, max_pos, pos, scratch, out, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_attention_decode_dev(_ptr(qkv), _ptr(rope_sin), _ptr(rope_cos), _ptr(kcache), _ptr(vcache), n_heads, n_kv, head_dim, max_pos, _ptr(pos), _ptr(scratch), _ptr(out), _vp(stream)))

    def attention_decode_wide_dev(self, qkv, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv, head_dim, max_pos, pos, scratch, out, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_attention_decode_wide_dev(_ptr(qkv), _ptr(rope_sin), _ptr(rope_cos), _ptr(kcache), _ptr(vcache), n_heads, n_kv, head_dim, max_pos, _ptr(pos), _ptr(scratch), _ptr(out), _vp(stream)))

    def attention_decode_partial_dev(self, qkv, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv, head_dim, max_pos, pos, scratch, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_attention_decode_partial_dev(_ptr(qkv), _ptr(rope_sin), _ptr(rope_cos), _ptr(kcache), _ptr(vcache), n_heads, n_kv, head_dim, max_pos, _ptr(pos), _ptr(scratch), _vp(stream)))

    def gemv_attn_merge_dev(self, h: int, scratch, n_heads: int, n_kv: int, max_pos: int, pos, y, residual=None, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_gemv_attn_merge_dev(h, _ptr(scratch), n_heads, n_kv, max_pos, _ptr(pos), _ptr(y), _ptr(residual) if residual is not None else None, _vp(stream)))

    def logits_f16_dev(self, table, x, gamma, eps, hidden, vocab, logits, scratch, n_wg, token=None, pos=None, history=None, n_forced=None, stream: int = 0) -> None:
        opt = lambda t: _ptr(t) if t is not None else None
        self._check(self.c.bitnet_hip_logits_f16_dev(_ptr(table), _ptr(x), opt(gamma), eps, hidden, vocab, _ptr(logits), _ptr(scratch), n_wg, opt(token), opt(pos), opt(history), opt(n_forced), _vp(stream)))

    def argmax_dev(self, v, n: int, scratch, n_wg: int, token, stream: int = 0) -> None:
        self._check(self.c.bitnet_hip_argmax_dev(_ptr(v), n, _ptr(scratch), n_wg, _ptr(token), _vp(stream)))

    def hbm_read_ceiling(self, nbytes: int = 2 << 30, iters: int = 10, stream: int = 0):
        """Measured read-only stream ceiling of the device: (best, mean) GB/s."""
        best, mean = C.c_double(0.0), C.c_double(0.0)
        self._check(self.c.bitnet_hip_hbm_read_ceiling(nbytes, iters, C.byref(best), C.byref(mean), _vp(stream)))
        return best.value, mean.value


_lib = None


def load() -> HipLib:
    """Load libbitnet_hip.so (building nothing: call build() first)."""
    global _lib
    if _lib is None:
        # torch carries its own copy of the HIP runtime; when both it and /opt/rocm's (which this library links) end up in one
        # process the one loaded SECOND finds no device ("no ROCm-capable device is detected").  Loading torch's first makes this
        # library bind to the runtime that is already there -- these bindings use torch for device memory anyway.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = HipLib()
    return _lib


def declared_symbols() -> list[str]:
    """Every function include/bitnet_hip.h declares (used by the export test)."""
    import re

    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bitnet_hip_[a-z0-9_]+)\s*\(", text)))


# ---------------------------------------------------------------------------
# Host decode loop (C++ above the C ABI: bitnet-rs_amd/host/decoder.{hpp,cpp})
# ---------------------------------------------------------------------------

HOST_LIB_PATH = os.path.join(HERE, "libbitnet_host.so")


def blake3_hex(data: bytes) -> str:
    """BLAKE3 as the trace records carry it (host/blake3.hpp)."""
    load()
    L = C.CDLL(HOST_LIB_PATH)
    L.bitnet_host_blake3_hex.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    out = C.create_string_buffer(65)
    if L.bitnet_host_blake3_hex(data, len(data), out) != 0:
        raise BitNetHipError(ERR_INVALID_ARGUMENT, "blake3_hex failed")
    return out.value.decode()


class HostConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden", "n_layers", "n_heads", "n_kv_heads", "head_dim", "ffn", "vocab", "max_pos")] + [
        ("eps", C.c_float),
        ("rope_theta", C.c_float),
    ]


FLAVORS = ("BitNet32F16", "Split32WithSibling", "GgmlQk256NoScale")


class GgufFile:
    """ctypes view of the C++ GGUF reader (bitnet-rs_amd/host/gguf.cpp): header / KV / tensor
    records, sizes from offsets, I2_S flavour detection.  Pure host code: usable without a GPU."""

    def __init__(self, path: str | None = None, data: bytes | None = None, lib_path: str = HOST_LIB_PATH):
        if not os.path.exists(lib_path):
            raise FileNotFoundError(f"{lib_path} not found: build it with __graft_entry__.build()")
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not found: build it with __graft_entry__.build()")
        C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L = self.c = C.CDLL(lib_path)
        L.bitnet_host_gguf_open.restype = C.c_void_p
        L.bitnet_host_gguf_open.argtypes = [C.c_char_p]
        L.bitnet_host_gguf_from_memory.restype = C.c_void_p
        L.bitnet_host_gguf_from_memory.argtypes = [_u8p, _sz]
        L.bitnet_host_gguf_close.argtypes = [C.c_void_p]
        L.bitnet_host_gguf_error.restype = C.c_char_p
        L.bitnet_host_gguf_tensor_count.restype = C.c_int64
        L.bitnet_host_gguf_tensor_count.argtypes = [C.c_void_p]
        L.bitnet_host_gguf_tensor_info.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, _sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.bitnet_host_gguf_data_start.restype = C.c_uint64
        L.bitnet_host_gguf_data_start.argtypes = [C.c_void_p]
        L.bitnet_host_gguf_config.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
        L.bitnet_host_gguf_detect_i2s_flavor.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        L.bitnet_host_gguf_loader_is_qk256.argtypes = [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64]
        L.bitnet_host_gguf_check_projection.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64]
        self.h = None
        if path is not None:
            self.h = L.bitnet_host_gguf_open(path.encode())
        elif data is not None:
            self._buf = np.frombuffer(data, np.uint8).copy()  # must outlive the reader
            self.h = L.bitnet_host_gguf_from_memory(self._buf.ctypes.data_as(_u8p), self._buf.size)
        if (path is not None or data is not None) and not self.h:
            raise BitNetHipError(ERR_INVALID_ARGUMENT, self.error() or "GGUF parse failed")

    def error(self) -> str:
        e = self.c.bitnet_host_gguf_error()
        return e.decode(errors="replace") if e else ""

    def close(self) -> None:
        if self.h:
            self.c.bitnet_host_gguf_close(self.h)
            self.h = None

    @property
    def data_start(self) -> int:
        return int(self.c.bitnet_host_gguf_data_start(self.h))

    def tensors(self) -> list:
        out = []
        for i in range(self.c.bitnet_host_gguf_tensor_count(self.h)):
            name = C.create_string_buffer(512)
            shape = (C.c_uint64 * 8)()
            nd, ty, off, size = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
            rc = self.c.bitnet_host_gguf_tensor_info(self.h, i, name, 512, shape, C.byref(nd), C.byref(ty), C.byref(off), C.byref(size))
            if rc != 0:
                raise BitNetHipError(ERR_INVALID_ARGUMENT, self.error())
            out.append({"name": name.value.decode(), "shape": tuple(int(shape[j]) for j in range(nd.value)), "type": ty.value,
                        "offset": off.value, "size": size.value})
        return out

    def config(self) -> dict:
        cfg, f = (C.c_uint64 * 6)(), (C.c_float * 2)()
        if self.c.bitnet_host_gguf_config(self.h, cfg, f) != 0:
            raise BitNetHipError(ERR_INVALID_ARGUMENT, self.error())
        keys = ("vocab", "hidden", "n_layers", "n_heads", "n_kv_heads", "ffn")
        d = {k: int(cfg[i]) for i, k in enumerate(keys)}
        d["rope_theta"] = None if f[0] != f[0] else float(f[0])
        d["eps"] = None if f[1] != f[1] else float(f[1])
        return d

    def detect_i2s_flavor(self, available: int, nelems: int, has_scale_sibling: bool, strict: bool = False) -> str:
        rc = self.c.bitnet_host_gguf_detect_i2s_flavor(available, nelems, int(has_scale_sibling), int(strict))
        if rc < 0:
            raise BitNetHipError(ERR_INVALID_ARGUMENT, self.error())
        return FLAVORS[rc]

    def check_projection(self, idx: int, rows: int, cols: int) -> None:
        """The loader's per-projection step without a device (flavour decision, size / bounds checks, codes + scales
        split); raises with the loader's message when the tensor would be refused."""
        if self.c.bitnet_host_gguf_check_projection(self.h, idx, rows, cols) != 0:
            raise BitNetHipError(ERR_INVALID_ARGUMENT, self.error())

    def loader_is_qk256(self, shape, available: int) -> bool:
        sh = (C.c_uint64 * len(shape))(*shape)
        return bool(self.c.bitnet_host_gguf_loader_is_qk256(sh, len(shape), available))


class HostDecoder:
    """ctypes view of the C++ Decoder (mirror of the reference's Rust-side
    TransformerModel / KVCache / greedy loop).  No arithmetic here."""

    def __init__(self, cfg, path: str = HOST_LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: build it with __graft_entry__.build() -- there is no fallback path")
        load()  # the kernel library must be loadable first
        self.c = C.CDLL(path)
        L = self.c
        L.bitnet_host_create.restype = C.c_void_p
        L.bitnet_host_create.argtypes = [C.POINTER(HostConfig)]
        L.bitnet_host_destroy.argtypes = [C.c_void_p]
        L.bitnet_host_error.restype = C.c_char_p
        L.bitnet_host_error.argtypes = [C.c_void_p]
        L.bitnet_host_set_layer_qk256.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p] + [_u8p] * 7
        L.bitnet_host_set_layer_i2s.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.POINTER(_u8p), C.POINTER(_f32p), _sz]
        L.bitnet_host_set_globals.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), _f32p]
        L.bitnet_host_reset.argtypes = [C.c_void_p]
        L.bitnet_host_feed.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        L.bitnet_host_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.bitnet_host_run_reference.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.bitnet_host_prepare_graphs.argtypes = [C.c_void_p, C.c_int]
        L.bitnet_host_set_act_mode.argtypes = [C.c_void_p, C.c_int]
        L.bitnet_host_act_mode.argtypes = [C.c_void_p]
        L.bitnet_host_position.argtypes = [C.c_void_p]
        L.bitnet_host_last_prefill_path.argtypes = [C.c_void_p]
        L.bitnet_host_saturation_fallbacks.argtypes = [C.c_void_p]
        L.bitnet_host_history.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        L.bitnet_host_last_logits.argtypes = [C.c_void_p, _f32p]
        L.bitnet_host_last_hidden.argtypes = [C.c_void_p, _f32p]
        L.bitnet_host_probe_gateup.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.bitnet_host_probe_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.bitnet_host_weight_bytes.argtypes = [C.c_void_p]
        L.bitnet_host_weight_bytes.restype = C.c_uint64
        self.cfg = cfg
        hc = HostConfig(**{k: (float(v) if k in ("eps", "rope_theta") else int(v)) for k, v in cfg.asdict().items()})
        self.h = L.bitnet_host_create(C.byref(hc))
        if not self.h:
            raise BitNetHipError(ERR_GPU, "bitnet_host_create failed (allocation)")
        err = self.error()
        if err:
            self.close()  # a dead decoder (rejected configuration, failed allocation) is destroyed before the error is raised
            raise BitNetHipError(ERR_GPU, err)

    def error(self) -> str:
        e = self.c.bitnet_host_error(self.h)
        return e.decode() if e else ""

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise BitNetHipError(rc, self.error() or f"bitnet_host rc={rc}")

    def close(self) -> None:
        if self.h:
            self.c.bitnet_host_destroy(self.h)
            self.h = None

    def load_gguf(self, gguf: "GgufFile") -> None:
        self.c.bitnet_host_load_gguf.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.c.bitnet_host_load_gguf(self.h, gguf.h))

    def set_layer_qk256(self, layer: int, w: dict) -> None:
        a = [_np(w["attn_norm"], np.float32), _np(w["ffn_norm"], np.float32)] + [_np(w[k], np.uint8) for k in ("q", "k", "v", "o", "gate", "up", "down")]
        self._check(self.c.bitnet_host_set_layer_qk256(self.h, layer, a[0].ctypes.data_as(_f32p), a[1].ctypes.data_as(_f32p), *[x.ctypes.data_as(_u8p) for x in a[2:]]))

    def set_layer_i2s(self, layer: int, w: dict, block: int) -> None:
        names = ("q", "k", "v", "o", "gate", "up", "down")
        ws = [_np(w[k], np.uint8) for k in names]
        ss = [_np(w[k + "_scales"], np.float32) for k in names]
        wp = (_u8p * 7)(*[x.ctypes.data_as(_u8p) for x in ws])
        sp = (_f32p * 7)(*[x.ctypes.data_as(_f32p) for x in ss])
        an, fn = _np(w["attn_norm"], np.float32), _np(w["ffn_norm"], np.float32)
        self._check(self.c.bitnet_host_set_layer_i2s(self.h, layer, an.ctypes.data_as(_f32p), fn.ctypes.data_as(_f32p), wp, sp, block))

    def set_globals(self, g: dict) -> None:
        e, f = _np(g["embed_f16"], np.uint16), _np(g["final_norm"], np.float32)
        self._check(self.c.bitnet_host_set_globals(self.h, e.ctypes.data_as(C.POINTER(C.c_uint16)), f.ctypes.data_as(_f32p)))

    def reset(self) -> None:
        self._check(self.c.bitnet_host_reset(self.h))

    def feed(self, tokens) -> None:
        t = _np(tokens, np.int32)
        self._check(self.c.bitnet_host_feed(self.h, t.ctypes.data_as(C.POINTER(C.c_int32)), t.size))

    def run(self, n: int, with_logits: bool = True, use_graph: bool = True) -> float:
        ms = C.c_float(0)
        self._check(self.c.bitnet_host_run(self.h, n, int(with_logits), int(use_graph), C.byref(ms)))
        return ms.value

    def set_kv_f16(self, on: bool) -> None:
        """Opt-in f16 KV cache (fresh sequence only)."""
        self.c.bitnet_host_set_kv_f16.argtypes = [C.c_void_p, C.c_int]
        self._check(self.c.bitnet_host_set_kv_f16(self.h, int(on)))

    def set_act_mode(self, mode: int) -> None:
        """1 (default): activations quantised once by their producer (QAct); 0: exact f32 activations."""
        self._check(self.c.bitnet_host_set_act_mode(self.h, mode))

    def act_mode(self) -> int:
        return int(self.c.bitnet_host_act_mode(self.h))

    def trace_step(self, directory: str, with_logits: bool = True) -> None:
        """One eager step leaving the reference's per-tensor trace records (crates/bitnet-trace) in `directory`."""
        self.c.bitnet_host_trace_step.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        os.makedirs(directory, exist_ok=True)
        self._check(self.c.bitnet_host_trace_step(self.h, directory.encode(), int(with_logits)))

    def prepare_graphs(self, with_logits: bool = True) -> None:
        """capture + instantiate the step graph of every attention form a sequence can reach, ahead of time (nothing executes)"""
        self._check(self.c.bitnet_host_prepare_graphs(self.h, int(with_logits)))

    def run_reference(self, n: int, with_logits: bool = True) -> None:
        """n UNFUSED steps on the bit-exact reference-order kernels (the checker of the fast step; slow)."""
        self._check(self.c.bitnet_host_run_reference(self.h, n, int(with_logits)))

    GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

    def prefill_sharded(self, n: int, rank: int, world: int, gather=None, with_logits: bool = True, digits: int = 3, wire_f16: bool = True,
                        rccl_comm: int | None = None) -> float:
        """Token-parallel prefill of the first n fed tokens: this process is `rank` of `world` (one process per GPU).
        gather(send_ptr, recv_ptr, bytes_per_rank, stream) -> 0 is the all-gather the host supplies (torch.distributed in the
        tests, see prefill_parallel.torch_gather); rccl_comm (an ncclComm_t as int) takes the C entry
        bitnet_host_rccl_allgather instead: the path a Rust host would use, no Python per layer."""
        L = self.c
        L.bitnet_host_prefill_sharded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        ms = C.c_float(0)
        if rccl_comm is not None:
            fn = C.cast(L.bitnet_host_rccl_allgather, C.c_void_p)
            self._check(L.bitnet_host_prefill_sharded(self.h, n, rank, world, fn, C.c_void_p(rccl_comm), int(with_logits), digits, int(wire_f16), C.byref(ms)))
            return ms.value
        cb = None
        if gather is not None:
            def _cb(ctx, send, recv, nbytes, stream):
                try:
                    return int(gather(send, recv, nbytes, stream) or 0)
                except Exception as e:  # noqa: BLE001 -- nothing may propagate through the C frames
                    sys.stderr.write(f"gather callback failed: {e!r}\n")
                    return -1
            cb = self.GATHER_FN(_cb)
        self._check(L.bitnet_host_prefill_sharded(self.h, n, rank, world, C.cast(cb, C.c_void_p) if cb else None, None, int(with_logits), digits,
                                                  int(wire_f16), C.byref(ms)))
        return ms.value

    def set_phase_timing(self, on: bool) -> None:
        """per-phase event timing of the next prefill_sharded calls (8 event records per layer)"""
        self.c.bitnet_host_set_phase_timing.argtypes = [C.c_void_p, C.c_int]
        self._check(self.c.bitnet_host_set_phase_timing(self.h, int(on)))

    def phase_times(self) -> dict:
        """medians over the layers of the last timed prefill_sharded call, microseconds per layer"""
        out = (C.c_float * 4)()
        self.c.bitnet_host_phase_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self._check(self.c.bitnet_host_phase_times(self.h, out))
        return {"matmul_us": round(out[0], 1), "attention_us": round(out[1], 1), "gather_wait_us": round(out[2], 1), "gather_us": round(out[3], 1)}

    def prefill(self, n: int, with_logits: bool = True, digits: int = 4) -> float:
        self.c.bitnet_host_prefill.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        ms = C.c_float(0)
        self._check(self.c.bitnet_host_prefill(self.h, n, int(with_logits), digits, C.byref(ms)))
        return ms.value

    def last_prefill_path(self) -> int:
        """What the last prefill() ran: 0 digit planes (+ f16 hand-over / hybrid o / down), 1 the f16 chain, 2 the QB32 chain."""
        return int(self.c.bitnet_host_last_prefill_path(self.h))

    def saturation_fallbacks(self) -> int:
        """Prompts this decoder repeated on the row-scaled forms because an f16 hand-over value was clamped (bitnet_hip_f16_saturations)."""
        return int(self.c.bitnet_host_saturation_fallbacks(self.h))

    def position(self) -> int:
        return int(self.c.bitnet_host_position(self.h))

    def history(self, n: int) -> np.ndarray:
        out = np.zeros(n, np.int32)
        self._check(self.c.bitnet_host_history(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return out

    def last_logits(self) -> np.ndarray:
        out = np.zeros(self.cfg.vocab, np.float32)
        self._check(self.c.bitnet_host_last_logits(self.h, out.ctypes.data_as(_f32p)))
        return out

    def last_hidden(self) -> np.ndarray:
        out = np.zeros(self.cfg.hidden, np.float32)
        self._check(self.c.bitnet_host_last_hidden(self.h, out.ctypes.data_as(_f32p)))
        return out

    def probe_gateup(self, reps: int):
        us, b = C.c_float(0), C.c_double(0)
        self._check(self.c.bitnet_host_probe_gateup(self.h, reps, C.byref(us), C.byref(b)))
        return us.value, b.value

    def probe_kernel(self, kind: int, reps: int):
        us, b = C.c_float(0), C.c_double(0)
        self._check(self.c.bitnet_host_probe_kernel(self.h, kind, reps, C.byref(us), C.byref(b)))
        return us.value, b.value

    def weight_bytes(self) -> int:
        return int(self.c.bitnet_host_weight_bytes(self.h))
