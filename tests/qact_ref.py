"""numpy restatement of the QAct activation format (bitnet-rs_amd/csrc/qact.hpp) -- test infrastructure.

Per 16 consecutive elements one power-of-two scale as = 2^(E-13) (E = exponent of the group's absolute maximum, biased
exponent clamped to [32, 254]) and q = floor(u * 2^(13-E) + 1/2) per element, stored as two balanced base-256 digits
(q = 256 d1 + d0, d0 in [-128, 127]).  Per 256 elements one 576-byte record: d0 plane, d1 plane, 16 f32 scales in the
slot order of the consuming MFMA (group t -> slot 4 (t // 4) + 2 (t % 2) + (t % 4) // 2)."""
import numpy as np

QREC = 576


def _slot(t: int) -> int:
    g, m = t // 4, t % 4
    return 4 * g + 2 * (m & 1) + (m >> 1)


def quantize_qact(x: np.ndarray, gamma: np.ndarray | None = None) -> np.ndarray:
    x = np.asarray(x, np.float32)
    n = x.size
    assert n % 16 == 0
    u = x if gamma is None else (x * np.asarray(gamma, np.float32)).astype(np.float32)  # one f32 rounding, as on the device
    nrec = (n + 255) // 256
    out = np.zeros(nrec * QREC, np.uint8)
    ug = u.reshape(-1, 16)
    am = np.abs(ug).max(axis=1)
    be = (am.view(np.uint32) >> 23).astype(np.int64)
    be = np.clip(be, 32, 254)
    sc = ((267 - be).astype(np.uint32) << 23).view(np.float32)
    as_ = ((be - 13).astype(np.uint32) << 23).view(np.float32)
    # u * sc is exact (power of two, result normal or zero... or below the f32 range: flush like the hardware's f32 multiply)
    prod = (ug * sc[:, None]).astype(np.float32)
    q = np.floor(prod.astype(np.float64) + 0.5).astype(np.int64)  # v_cvt_rpi_i32_f32
    t = ((q + 0x80) & 0xFFFFFFFF) ^ 0x80
    d0 = (t & 0xFF).astype(np.uint8)
    d1 = ((t >> 8) & 0xFF).astype(np.uint8)
    for grp in range(n // 16):
        rec, tp = grp // 16, grp % 16
        base = rec * QREC
        out[base + 16 * tp: base + 16 * tp + 16] = d0[grp]
        out[base + 256 + 16 * tp: base + 256 + 16 * tp + 16] = d1[grp]
        out[base + 512 + 4 * _slot(tp): base + 512 + 4 * _slot(tp) + 4] = np.frombuffer(as_[grp].tobytes(), np.uint8)
    return out


def dequantize_qact(rec: np.ndarray, n: int) -> np.ndarray:
    rec = np.asarray(rec, np.uint8)
    out = np.zeros(n, np.float64)
    for grp in range(n // 16):
        r, tp = grp // 16, grp % 16
        base = r * QREC
        d0 = rec[base + 16 * tp: base + 16 * tp + 16].view(np.int8).astype(np.float64)
        d1 = rec[base + 256 + 16 * tp: base + 256 + 16 * tp + 16].view(np.int8).astype(np.float64)
        as_ = float(rec[base + 512 + 4 * _slot(tp): base + 512 + 4 * _slot(tp) + 4].view(np.float32)[0])
        out[16 * grp: 16 * grp + 16] = (256.0 * d1 + d0) * as_
    return out
