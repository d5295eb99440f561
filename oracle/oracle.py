"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under bitnet-rs_amd/ does.  See bitnet_oracle.h for what
each function restates (reference file:line).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbitnet_oracle.so")
REF_LIB_PATH = os.path.join(HERE, "_ref", "libggml_iq2s_ref.so")
ERRLEN = 256

_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_sz = C.c_size_t


class OracleError(RuntimeError):
    pass


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "-B", os.path.join(HERE, "libbitnet_oracle.so")])
    if not os.path.exists(REF_LIB_PATH) or force:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L: C.CDLL) -> None:
    gemv_sig = [_u8p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, _sz, C.c_char_p]
    for name in ("bo_gemv_qk256_scalar", "bo_gemv_qk256_avx2", "bo_gemv_qk256"):
        getattr(L, name).argtypes = gemv_sig
        getattr(L, name).restype = C.c_int
    L.bo_gemv_qk256_avx2_mt.argtypes = [_u8p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, _sz, C.c_int, C.c_char_p]
    L.bo_gemv_qk256_avx2_mt.restype = C.c_int
    L.bo_gemv_qk256_f64.argtypes = [_u8p, _f32p, _f64p, _sz, _sz, _sz]
    L.bo_gemv_qk256_f64.restype = None
    L.bo_have_avx2.restype = C.c_int
    L.bo_unpack_qk256_block.argtypes = [_u8p, _u8p]
    L.bo_code_to_f32.argtypes = [C.c_uint8]
    L.bo_code_to_f32.restype = C.c_float
    L.bo_gemv_qk256_row.argtypes = [_u8p, _f32p, _sz]
    L.bo_gemv_qk256_row.restype = C.c_float
    L.bo_i2s_qk256_new.argtypes = [_sz, _sz, _sz, C.POINTER(_sz), C.c_char_p]
    L.bo_i2s_qk256_new.restype = C.c_int
    L.bo_qk256_dispatch_gemv_scalar.argtypes = [_f32p, _sz, _sz, _u8p, _sz, _f32p, _sz, _f32p, C.c_char_p]
    L.bo_qk256_dispatch_gemv_scalar.restype = C.c_int
    L.bo_decode_i2s.argtypes = [C.c_uint8]
    L.bo_decode_i2s.restype = C.c_int8
    L.bo_pack_i2s.argtypes = [_i8p]
    L.bo_pack_i2s.restype = C.c_uint8
    mm_sig = [_f32p, _sz, _u8p, _sz, _f32p, _sz, _f32p, _sz, _sz, _sz, _sz, _sz, C.c_char_p]
    for name in ("bo_i2s_matmul_f32", "bo_i2s_matmul_blocked", "bo_dequantize_and_matmul"):
        getattr(L, name).argtypes = mm_sig
        getattr(L, name).restype = C.c_int
    L.bo_matmul_i2s.argtypes = [_i8p, _sz, _u8p, _sz, _f32p, _sz, _sz, _sz, _sz, C.c_char_p]
    L.bo_matmul_i2s.restype = C.c_int
    L.bo_quantize_i2s.argtypes = [_f32p, _sz, _u8p, _sz, _f32p, _sz, C.c_char_p]
    L.bo_quantize_i2s.restype = C.c_int
    L.bo_f16_to_f32.argtypes = [C.c_uint16]
    L.bo_f16_to_f32.restype = C.c_float
    L.bo_i2s_dequant_block.argtypes = [_f32p, _u8p, _sz, C.c_uint16, C.c_int, C.c_float]
    L.bo_i2s_expected_bytes.argtypes = [_sz, _sz, _sz]
    L.bo_i2s_expected_bytes.restype = _sz
    L.bo_i2s_infer_block_size.argtypes = [_sz, _sz, _sz]
    L.bo_i2s_infer_block_size.restype = _sz
    L.bo_i2s_dequantize_to_f32.argtypes = [_u8p, _sz, _sz, _sz, C.c_int, C.c_float, C.c_int, _f32p, C.c_char_p]
    L.bo_i2s_dequantize_to_f32.restype = C.c_int
    L.bo_pack_2bit_values.argtypes = [_i8p, _sz, _u8p]
    L.bo_unpack_2bit_values.argtypes = [_u8p, _sz, _sz, _i8p]
    L.bo_dequantize_blocks.argtypes = [_i8p, _sz, _f32p, _sz, _f32p]


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


def _i8(a):
    a = np.ascontiguousarray(a, dtype=np.int8)
    return a, a.ctypes.data_as(_i8p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _check(rc: int, err) -> None:
    if rc != 0:
        raise OracleError(err.value.decode() or f"oracle rc={rc}")


# ---- QK256 -------------------------------------------------------------


def unpack_qk256_block(qs64) -> np.ndarray:
    q, qp = _u8(qs64)
    assert q.size == 64
    out = np.zeros(256, np.uint8)
    lib().bo_unpack_qk256_block(qp, out.ctypes.data_as(_u8p))
    return out


def code_to_f32(code: int) -> float:
    return float(lib().bo_code_to_f32(code))


def gemv_qk256_row(qs_row, x, cols: int) -> float:
    q, qp = _u8(qs_row)
    xa, xp = _f32(x)
    return float(lib().bo_gemv_qk256_row(qp, xp, cols))


def gemv_qk256(qs, x, rows: int, cols: int, row_stride_bytes: int, y_len: int | None = None,
               impl: str = "dispatch", threads: int = 1) -> np.ndarray:
    """impl: 'dispatch' (reference behaviour), 'scalar', 'avx2', 'avx2_mt'."""
    q, qp = _u8(qs)
    xa, xp = _f32(x)
    y = np.zeros(rows if y_len is None else y_len, np.float32)
    err = C.create_string_buffer(ERRLEN)
    L = lib()
    args = (qp, q.size, xp, xa.size, y.ctypes.data_as(_f32p), y.size, rows, cols, row_stride_bytes)
    if impl == "avx2_mt":
        rc = L.bo_gemv_qk256_avx2_mt(*args, threads, err)
    else:
        fn = {"dispatch": L.bo_gemv_qk256, "scalar": L.bo_gemv_qk256_scalar, "avx2": L.bo_gemv_qk256_avx2}[impl]
        rc = fn(*args, err)
    _check(rc, err)
    return y


def gemv_qk256_f64(qs, x, rows: int, cols: int, row_stride_bytes: int) -> np.ndarray:
    q, qp = _u8(qs)
    xa, xp = _f32(x)
    y = np.zeros(rows, np.float64)
    lib().bo_gemv_qk256_f64(qp, xp, y.ctypes.data_as(_f64p), rows, cols, row_stride_bytes)
    return y


def have_avx2() -> bool:
    return bool(lib().bo_have_avx2())


def i2s_qk256_new(rows: int, cols: int, qs_len: int) -> int:
    stride = _sz(0)
    err = C.create_string_buffer(ERRLEN)
    _check(lib().bo_i2s_qk256_new(rows, cols, qs_len, C.byref(stride), err), err)
    return stride.value


def qk256_dispatch_gemv_scalar(rows, cols, packed, scales, activations) -> np.ndarray:
    p, pp = _u8(packed)
    s, sp = _f32(scales)
    a, ap = _f32(activations)
    out = np.zeros(rows, np.float32)
    err = C.create_string_buffer(ERRLEN)
    _check(lib().bo_qk256_dispatch_gemv_scalar(out.ctypes.data_as(_f32p), rows, cols, pp, p.size, sp, s.size, ap, err), err)
    return out


# ---- ternary -----------------------------------------------------------


def decode_i2s(bits: int) -> int:
    return int(lib().bo_decode_i2s(bits))


def pack_i2s(vals) -> int:
    v, vp = _i8(vals)
    assert v.size == 4
    return int(lib().bo_pack_i2s(vp))


def i2s_matmul(act, w, scales, m, n, k, block_size, impl="f32", out_len=None) -> np.ndarray:
    a, ap = _f32(act)
    wq, wp = _u8(w)
    s, sp = _f32(scales)
    out = np.zeros(m * n if out_len is None else out_len, np.float32)
    err = C.create_string_buffer(ERRLEN)
    L = lib()
    fn = {"f32": L.bo_i2s_matmul_f32, "blocked": L.bo_i2s_matmul_blocked, "dequant": L.bo_dequantize_and_matmul}[impl]
    _check(fn(ap, a.size, wp, wq.size, sp, s.size, out.ctypes.data_as(_f32p), out.size, m, n, k, block_size, err), err)
    return out


# ---- provider ----------------------------------------------------------


def matmul_i2s(a, b, m, n, k, c_len=None) -> np.ndarray:
    aa, ap = _i8(a)
    bb, bp = _u8(b)
    c = np.zeros(m * n if c_len is None else c_len, np.float32)
    err = C.create_string_buffer(ERRLEN)
    _check(lib().bo_matmul_i2s(ap, aa.size, bp, bb.size, c.ctypes.data_as(_f32p), c.size, m, n, k, err), err)
    return c


def quantized_matmul_i2s(x, packed, scales, block_size, m, n, k) -> np.ndarray:
    """QuantizedLinear::quantized_matmul_i2s (quantized_linear.rs:704-802)."""
    xa, xp = _f32(x)
    pa, pp = _u8(packed)
    sa, sp = _f32(scales)
    out = np.zeros(m * n, np.float32)
    err = C.create_string_buffer(ERRLEN)
    L = lib()
    L.bo_quantized_matmul_i2s.argtypes = [_f32p, _sz, _u8p, _sz, _f32p, _sz, _sz, _f32p, _sz, _sz, _sz, _sz, C.c_char_p]
    L.bo_quantized_matmul_i2s.restype = C.c_int
    _check(L.bo_quantized_matmul_i2s(xp, xa.size, pp, pa.size, sp, sa.size, block_size, out.ctypes.data_as(_f32p), out.size, m, n, k, err), err)
    return out


def quantize_i2s(x, out_len=None, scales_len=None):
    xa, xp = _f32(x)
    out = np.zeros(xa.size // 4 if out_len is None else out_len, np.uint8)
    scales = np.zeros((xa.size + 31) // 32 if scales_len is None else scales_len, np.float32)
    err = C.create_string_buffer(ERRLEN)
    _check(lib().bo_quantize_i2s(xp, xa.size, out.ctypes.data_as(_u8p), out.size, scales.ctypes.data_as(_f32p), scales.size, err), err)
    return out, scales


# ---- block dequant -----------------------------------------------------


def f16_to_f32(bits: int) -> float:
    return float(lib().bo_f16_to_f32(bits))


def i2s_dequant_block(qbits, n, scale_bits, inv=False, k=1.0) -> np.ndarray:
    q, qp = _u8(qbits)
    dst = np.zeros(n, np.float32)
    lib().bo_i2s_dequant_block(dst.ctypes.data_as(_f32p), qp, n, scale_bits, int(inv), k)
    return dst


def i2s_expected_bytes(rows, cols, block) -> int:
    return int(lib().bo_i2s_expected_bytes(rows, cols, block))


def i2s_infer_block_size(nbytes, rows, cols):
    b = int(lib().bo_i2s_infer_block_size(nbytes, rows, cols))
    return b or None


def i2s_dequantize_to_f32(data, rows, cols, inv=False, k=1.0, transposed=False) -> np.ndarray:
    d, dp = _u8(data)
    out = np.zeros(rows * cols, np.float32)
    err = C.create_string_buffer(ERRLEN)
    _check(lib().bo_i2s_dequantize_to_f32(dp, d.size, rows, cols, int(inv), k, int(transposed), out.ctypes.data_as(_f32p), err), err)
    return out


# ---- Q/utils.rs --------------------------------------------------------


def pack_2bit_values(values) -> np.ndarray:
    v, vp = _i8(values)
    out = np.zeros((v.size + 3) // 4, np.uint8)
    lib().bo_pack_2bit_values(vp, v.size, out.ctypes.data_as(_u8p))
    return out


def unpack_2bit_values(packed, output_len) -> np.ndarray:
    p, pp = _u8(packed)
    out = np.zeros(output_len, np.int8)
    lib().bo_unpack_2bit_values(pp, p.size, output_len, out.ctypes.data_as(_i8p))
    return out


def dequantize_blocks(q, scales, block_size) -> np.ndarray:
    qa, qp = _i8(q)
    s, sp = _f32(scales)
    out = np.zeros(qa.size, np.float32)
    lib().bo_dequantize_blocks(qp, qa.size, sp, block_size, out.ctypes.data_as(_f32p))
    return out


# ---- oracle/_ref: the reference's own vendored C, compiled where it lies ----


class _BlockIq2s(C.Structure):
    _pack_ = 1
    _fields_ = [("d", C.c_uint16), ("qs", C.c_uint8 * 64), ("qh", C.c_uint8 * 8), ("scales", C.c_uint8 * 8)]


def ref_available() -> bool:
    return os.path.exists(REF_LIB_PATH)


def ref_dequantize_row_iq2_s(d_bits: int, qs64) -> np.ndarray:
    """Run the reference's dequantize_row_iq2_s
    (crates/bitnet-ggml-ffi/csrc/ggml/src/ggml-quants.c:59-72) on one block."""
    R = C.CDLL(REF_LIB_PATH)
    R.dequantize_row_iq2_s.argtypes = [C.c_void_p, _f32p, C.c_int64]
    blk = _BlockIq2s()
    assert C.sizeof(blk) == 82
    blk.d = d_bits
    q = np.ascontiguousarray(qs64, dtype=np.uint8)
    for i in range(64):
        blk.qs[i] = int(q[i])
    y = np.zeros(256, np.float32)
    R.dequantize_row_iq2_s(C.byref(blk), y.ctypes.data_as(_f32p), 256)
    return y


# ---- decode step (transformer_oracle.c) ---------------------------------------


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("hidden", "n_layers", "n_heads", "n_kv_heads", "head_dim", "ffn", "vocab", "max_pos")] + [
        ("eps", C.c_float),
        ("rope_theta", C.c_float),
    ]


class OracleModel:
    """CPU restatement of the reference's per-token decode step (T:1482-1504)."""

    def __init__(self, cfg, layers: list, glob: dict, n_threads: int = 1):
        L = lib()
        L.bo_model_create.restype = C.c_void_p
        L.bo_model_create.argtypes = [C.POINTER(ModelCfg), C.c_int]
        L.bo_model_destroy.argtypes = [C.c_void_p]
        L.bo_model_set_layer.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p] + [_u8p] * 7
        L.bo_model_set_globals.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), _f32p]
        L.bo_kv_create.restype = C.c_void_p
        L.bo_kv_create.argtypes = [C.c_void_p]
        L.bo_kv_destroy.argtypes = [C.c_void_p]
        L.bo_kv_reset.argtypes = [C.c_void_p]
        L.bo_model_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _f32p, _f32p, _f32p]
        L.bo_argmax.argtypes = [_f32p, _sz]
        self.cfg = cfg
        mc = ModelCfg(**{k: (float(v) if k in ("eps", "rope_theta") else int(v)) for k, v in cfg.asdict().items()})
        self.m = L.bo_model_create(C.byref(mc), n_threads)
        self._keep = []
        L.bo_model_set_layer_dense.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.POINTER(_f32p)]
        for i, w in enumerate(layers):
            a = [np.ascontiguousarray(w["attn_norm"], np.float32), np.ascontiguousarray(w["ffn_norm"], np.float32)]
            if w.get("ternary"):  # BitNet32 storage as synth.make_layer(fmt="i2s") returns it: packed codes + "<name>_scales", W = t(code) * scale
                names = ("q", "k", "v", "o", "gate", "up", "down")
                a += [np.ascontiguousarray(w[k], np.uint8) for k in names] + [np.ascontiguousarray(w[k + "_scales"], np.float32) for k in names]
                self._keep.append(a)
                cp = (_u8p * 7)(*[x.ctypes.data_as(_u8p) for x in a[2:9]])
                sp = (_f32p * 7)(*[x.ctypes.data_as(_f32p) for x in a[9:16]])
                L.bo_model_set_layer_ternary.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.POINTER(_u8p), C.POINTER(_f32p), C.c_int]
                rc = L.bo_model_set_layer_ternary(self.m, i, a[0].ctypes.data_as(_f32p), a[1].ctypes.data_as(_f32p), cp, sp, int(w["ternary"]))
            elif w.get("dense"):  # projections as dense f32 [out, in] (loader-dequantised flavours)
                a += [np.ascontiguousarray(w[k], np.float32) for k in ("q", "k", "v", "o", "gate", "up", "down")]
                self._keep.append(a)
                wp = (_f32p * 7)(*[x.ctypes.data_as(_f32p) for x in a[2:]])
                rc = L.bo_model_set_layer_dense(self.m, i, a[0].ctypes.data_as(_f32p), a[1].ctypes.data_as(_f32p), wp)
            else:
                a += [np.ascontiguousarray(w[k], np.uint8) for k in ("q", "k", "v", "o", "gate", "up", "down")]
                self._keep.append(a)
                rc = L.bo_model_set_layer(self.m, i, a[0].ctypes.data_as(_f32p), a[1].ctypes.data_as(_f32p), *[x.ctypes.data_as(_u8p) for x in a[2:]])
            assert rc == 0
        e, f = np.ascontiguousarray(glob["embed_f16"], np.uint16), np.ascontiguousarray(glob["final_norm"], np.float32)
        self._keep.append((e, f))
        L.bo_model_set_globals(self.m, e.ctypes.data_as(C.POINTER(C.c_uint16)), f.ctypes.data_as(_f32p))
        self.kv = L.bo_kv_create(self.m)

    def reset(self):
        lib().bo_kv_reset(self.kv)

    def step(self, token: int, want_logits: bool = True, want_trace: bool = False):
        """Returns (final-normed hidden, logits or None, per-layer residual trace or None)."""
        hid = np.zeros(self.cfg.hidden, np.float32)
        logits = np.zeros(self.cfg.vocab, np.float32) if want_logits else None
        trace = np.zeros((self.cfg.n_layers, self.cfg.hidden), np.float32) if want_trace else None
        rc = lib().bo_model_step(self.m, self.kv, int(token), hid.ctypes.data_as(_f32p),
                                 logits.ctypes.data_as(_f32p) if want_logits else None,
                                 trace.ctypes.data_as(_f32p) if want_trace else None)
        if rc != 0:
            raise OracleError(f"bo_model_step rc={rc}")
        return hid, logits, trace

    def logits(self, hidden) -> np.ndarray:
        """bo_logits (T:1599-1630): hidden . E^T over the f16-sourced table, rows dealt to this model's host threads."""
        h, hp = _f32(hidden)
        out = np.zeros(self.cfg.vocab, np.float32)
        L = lib()
        L.bo_logits.argtypes = [C.c_void_p, _f32p, _f32p]
        L.bo_logits.restype = None
        L.bo_logits(self.m, hp, out.ctypes.data_as(_f32p))
        return out

    def close(self):
        if self.m:
            lib().bo_kv_destroy(self.kv)
            lib().bo_model_destroy(self.m)
            self.m = None


def argmax(logits) -> int:
    a, ap = _f32(logits)
    lib().bo_argmax.argtypes = [_f32p, _sz]
    return int(lib().bo_argmax(ap, a.size))


def layernorm(x, w, eps) -> np.ndarray:
    xa, xp = _f32(x)
    wa, wp = _f32(w)
    out = np.zeros(xa.size, np.float32)
    L = lib()
    L.bo_layernorm.argtypes = [_f32p, _f32p, C.c_float, C.c_int, _f32p]
    L.bo_layernorm(xp, wp, eps, xa.size, out.ctypes.data_as(_f32p))
    return out


def rmsnorm(x, w, eps) -> np.ndarray:
    xa, xp = _f32(x)
    wa, wp = _f32(w)
    out = np.zeros(xa.size, np.float32)
    L = lib()
    L.bo_rmsnorm.argtypes = [_f32p, _f32p, C.c_float, C.c_int, _f32p]
    L.bo_rmsnorm(xp, wp, eps, xa.size, out.ctypes.data_as(_f32p))
    return out


def rope_tables(dim: int, max_seq_len: int, base: float):
    sin = np.zeros((max_seq_len, dim // 2), np.float32)
    cos = np.zeros((max_seq_len, dim // 2), np.float32)
    L = lib()
    L.bo_rope_build_tables.argtypes = [C.c_int, C.c_int, C.c_float, _f32p, _f32p]
    L.bo_rope_build_tables(dim, max_seq_len, base, sin.ctypes.data_as(_f32p), cos.ctypes.data_as(_f32p))
    return sin, cos


def rope_apply(x, pos: int, base: float = 10000.0) -> np.ndarray:
    """RotaryEmbedding::apply on head vectors [..., dim] at one position (T:134-163): split-half pairing (i, i + dim/2),
    tables from bo_rope_build_tables (crates/bitnet-rope/src/lib.rs:59-93)."""
    xa = np.ascontiguousarray(x, np.float32).copy()
    dim = xa.shape[-1]
    sin, cos = rope_tables(dim, pos + 1, base)
    L = lib()
    L.bo_rope_apply.argtypes = [_f32p, C.c_int, _f32p, _f32p]
    flat = xa.reshape(-1, dim)
    for r in range(flat.shape[0]):
        row = np.ascontiguousarray(flat[r])
        L.bo_rope_apply(row.ctypes.data_as(_f32p), dim, sin[pos].ctypes.data_as(_f32p), cos[pos].ctypes.data_as(_f32p))
        flat[r] = row
    return flat.reshape(xa.shape)
