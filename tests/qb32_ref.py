"""Host-side reading of a QB32 buffer (include/bitnet_hip.h, kernels_gemm.hip qb32_pack_unit) -- test infrastructure.

Buffer for m rows of `cols` columns (m_pad = m rounded up to 64, nblk = cols / 256):
    records [m_pad][nblk] of 592 bytes: digits [lane group g 4][digit d 3][MFMA m 2][24 bytes] (576 bytes: the 256 columns of the block),
                                        then the block's eight exponent bytes (unit U = 8 blk + 2 g + m: 32 columns) and 8 bytes of padding
A 24-byte piece holds 32 fp6 (e2m3) codes, k-slot j at bits [6 j, 6 j + 6), sign | magnitude of an integer digit in -16 .. 16 (the fp6 value is
digit / 8); k-slot 8 q + n of unit (g, m) is column 64 g + 32 m + 8 q + 4 (n & 1) + (n >> 1) of the 256-block (the order expand16_fp4 leaves the
weights' nibbles in).  value = (d0 + 32 d1 + 1024 d2) * 2^(s - 130), s = the unit's exponent byte (= E + 117, E the exponent of the unit's maximum)."""
import numpy as np


def decode(buf: np.ndarray, m: int, cols: int) -> np.ndarray:
    buf = np.asarray(buf, np.uint8)
    m_pad, nblk = -(-m // 64) * 64, cols // 256
    rec = buf[: m_pad * nblk * 592].reshape(m_pad, nblk, 592)[:m]
    dig = np.ascontiguousarray(rec[:, :, :576]).reshape(m, nblk, 4, 3, 2, 24)
    exps = np.ascontiguousarray(rec[:, :, 576:584]).reshape(m, nblk, 4, 2)
    bits = np.unpackbits(dig, axis=-1, bitorder="little").reshape(m, nblk, 4, 3, 2, 32, 6)
    code = (bits * (1 << np.arange(6))).sum(axis=-1)                 # [m, nblk, g, d, mm, k-slot]
    val = (code & 31).astype(np.int64) * np.where(code & 32, -1, 1)
    q = val[:, :, :, 0] + 32 * val[:, :, :, 1] + 1024 * val[:, :, :, 2]  # [m, nblk, g, mm, k-slot]
    j = np.arange(32)
    col_in_unit = 8 * (j // 8) + 4 * (j & 1) + ((j % 8) >> 1)         # k-slot -> column offset inside the unit
    out = np.zeros((m, nblk, 4, 2, 32), np.float64)
    out[..., col_in_unit] = q * np.exp2(exps.astype(np.float64) - 130.0)[..., None]
    return out.reshape(m, cols), exps.reshape(m, nblk * 8)


def unit_lsb(exps: np.ndarray) -> np.ndarray:
    """Quantisation step of every column: 2^(s - 130) of its unit."""
    return np.repeat(np.exp2(exps.astype(np.float64) - 130.0), 32, axis=1)
