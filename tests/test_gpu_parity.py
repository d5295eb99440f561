"""GPU parity: every call goes through the C ABI (libbitnet_hip.so) and is checked
against the CPU oracle on the same seeded inputs, plus the reference's own
known-answer vectors replayed on the device.

Bars (BASELINE.json north_star / SURVEY.md 8d):
  * integer / byte work (unpack, codes, quantize codes, i8 x u8 matmul): bit-exact;
  * reference-order kernel (KERNEL_EXACT): bit-exact f32 vs the scalar oracle;
  * streaming kernels: the reference's own `approx_eq_with_len`
    (crates/bitnet-models/tests/helpers/qk256_tolerance.rs:104-130):
    |a-b| < min(2e-4*sqrt(cols/256), 1e-3)  OR  |a-b|/max(|a|,|b|) < 2e-2,
    and, stricter, no further from the f64 truth than 4x the reference's own
    AVX2 path is.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODEL_SHAPES = [(2560, 2560), (640, 2560), (6912, 2560), (2560, 6912)]  # q/o, k/v, gate/up, down


def approx_eq_with_len(a, b, length):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    abs_tol = min(2e-4 * np.sqrt(length / 256.0), 1e-3)
    diff = np.abs(a - b)
    mag = np.maximum(np.abs(a), np.abs(b))
    rel_ok = (mag > 1e-6) & (diff / np.maximum(mag, 1e-30) < 2e-2)
    return (diff < abs_tol) | rel_ok


def qk256_inputs(rows, cols, seed):
    """Uniform 2-bit codes (seed) and x in U[-10,10) (seed+1): the generators of
    crates/bitnet-models/tests/qk256_avx2_correctness.rs:72-104, numpy PRNG."""
    stride = -(-cols // 256) * 64
    qs = np.random.default_rng(seed).integers(0, 256, rows * stride, dtype=np.uint8)
    x = np.random.default_rng(seed + 1).uniform(-10, 10, cols).astype(np.float32)
    return qs, x, stride


KERNELS = ["exact", "valu", "mfma", "mfma_tiled", "auto"]
STREAMING = ["valu", "mfma", "mfma_tiled", "auto"]


def _set(hip, pkg, name):
    hip.set_kernel({"auto": pkg.KERNEL_AUTO, "exact": pkg.KERNEL_EXACT, "valu": pkg.KERNEL_VALU, "mfma": pkg.KERNEL_MFMA, "mfma_tiled": pkg.KERNEL_MFMA_TILED}[name])


@pytest.fixture(autouse=True)
def _reset_kernel(hip, pkg):
    yield
    hip.set_kernel(pkg.KERNEL_AUTO)


# ------------------------------------------------------------------ lifecycle


def test_device_is_mi355x(hip):
    info = hip.device_info(0)
    assert info.gcn_arch.decode().startswith("gfx950")
    assert info.max_wavefront_size == 64
    assert hip.is_available() and hip.device_count() >= 1


# --------------------------------------- reference KATs replayed on the device


@pytest.mark.parametrize("kernel", KERNELS)
def test_kat_qk256_on_device(hip, pkg, kernel):
    _set(hip, pkg, kernel)
    # Q/i2s_qk256.rs:630-677: 1x256, code 2, ones -> 256.0
    assert hip.gemv_qk256(np.full(64, 0xAA, np.uint8), np.ones(256, np.float32), 1, 256, 64)[0] == 256.0
    # :440-465: 0x55, 3x256 -> -sum(x)
    x = np.arange(256, dtype=np.float32)
    y = hip.gemv_qk256(np.full(3 * 64, 0x55, np.uint8), x, 3, 256, 64)
    assert np.all(np.abs(y + x.sum()) < 1e-3)
    # :399-416: cols 512, x = i*0.01
    x = (np.arange(512, dtype=np.float32) * np.float32(0.01)).astype(np.float32)
    y = hip.gemv_qk256(np.full(128, 0xAA, np.uint8), x, 1, 512, 128)
    assert abs(y[0] - float(x.astype(np.float64).sum())) < 1e-3
    # :419-437: cols 300 (tail of 44 in the second block)
    x = (np.arange(300) % 7).astype(np.float32)
    y = hip.gemv_qk256(np.full(128, 0xAA, np.uint8), x, 1, 300, 128)
    assert abs(y[0] - float(x.sum())) < 1e-3
    # Q/qk256_dispatch.rs:106-119: 256x256, 0x55, act 0.5 -> all -128.0
    y = hip.gemv_qk256(np.full(256 * 64, 0x55, np.uint8), np.full(256, 0.5, np.float32), 256, 256, 64)
    assert np.all(y == -128.0)
    # LUT (:468-474): one-hot activations read the decoded weights back, bit-exact
    qs = np.array([(0b11100100 + (i & 3)) & 0xFF for i in range(64)], np.uint8)
    for j, want in zip(range(4), [-2.0, -1.0, 1.0, 2.0]):
        e = np.zeros(256, np.float32)
        e[j] = 1.0
        assert hip.gemv_qk256(qs, e, 1, 256, 64)[0] == want


def test_unpack_bit_exact_via_one_hot(hip, pkg, oracle):
    """Integer unpack must be bit-exact (north_star).  One-hot activations turn
    the GEMV into a read-back of every decoded weight of a random 16x512 matrix."""
    rows, cols = 16, 512
    qs, _, stride = qk256_inputs(rows, cols, 7)
    codes = np.stack([np.concatenate([oracle.unpack_qk256_block(qs[r * stride + b * 64 : r * stride + (b + 1) * 64]) for b in range(2)]) for r in range(rows)])
    want = np.array([-2, -1, 1, 2], np.float32)[codes]
    h = hip.weights_upload_qk256(qs, rows, cols, stride)
    import torch

    eye = torch.eye(cols, dtype=torch.float32, device="cuda")
    for kernel in KERNELS:
        _set(hip, pkg, kernel)
        y = torch.empty(cols, rows, dtype=torch.float32, device="cuda")
        hip.matmul_dev(h, eye, y, cols)
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy().T, want), kernel
    hip.weights_free(h)


@pytest.mark.parametrize("kernel", KERNELS)
def test_kat_ternary_on_device(hip, pkg, oracle, kernel):
    _set(hip, pkg, kernel)
    fmt = lambda out: "[" + ", ".join(f"{v:.6f}" for v in out) + "]"
    # snapshot_kernel_outputs.rs:151-177 / *.snap
    wp = np.array([oracle.pack_i2s([1, 1, 1, 1])] * 2, np.uint8)
    assert fmt(hip.i2s_matmul_f32([1, 2, 3, 4], wp, [1.0, 1.0], 1, 2, 4, 4)) == "[10.000000, 10.000000]"
    wp = np.array([oracle.pack_i2s([1, -1, 0, 1])], np.uint8)
    assert fmt(hip.i2s_matmul_f32([1, 2, 3, 4], wp, [2.0], 1, 1, 4, 4)) == "[6.000000]"
    # K/cpu/quantized_matmul.rs:555-613
    out = hip.i2s_matmul_f32(np.ones(8, np.float32), np.full(2, 0x55, np.uint8), [2.0, 0.5], 2, 2, 4, 32)
    assert np.allclose(out, [8, 2, 8, 2], atol=1e-5)
    out = hip.i2s_matmul_f32(np.ones(64, np.float32), np.full(16, 0x55, np.uint8), [1.0, 3.0], 1, 1, 64, 32)
    assert np.allclose(out, [128.0], atol=1e-4)


def test_ternary_shape_sweep_matches_oracle_bit_exact(hip, pkg, oracle):
    """K/cpu/quantized_matmul.rs:343-550 shapes (odd k, k%4!=0, block 32/256):
    these take the reference-order kernel, so the f32 results are bit-identical
    to the oracle's i2s_matmul_f32."""
    from test_oracle_kat import _SHAPES, _pack_weight_matrix

    for m, n, k, bs, wf, af, _tol in _SHAPES:
        w = [wf(i) for i in range(k * n)]
        act = np.array([af(i) for i in range(m * k)], np.float32)
        packed, scales = _pack_weight_matrix(w, k, n, bs)
        scales = (scales * np.linspace(0.5, 1.5, scales.size)).astype(np.float32)
        want = oracle.i2s_matmul(act, packed, scales, m, n, k, bs)
        hip.set_kernel(pkg.KERNEL_EXACT)
        got = hip.i2s_matmul_f32(act, packed, scales, m, n, k, bs)
        assert np.array_equal(got, want), (m, n, k, bs)
        hip.set_kernel(pkg.KERNEL_AUTO)
        got = hip.i2s_matmul_f32(act, packed, scales, m, n, k, bs)
        assert np.all(approx_eq_with_len(got, want, k)), (m, n, k, bs)


# -------------------------------------------------- seeded parity vs oracle


@pytest.mark.parametrize("rows,cols", [(4, 256), (3, 300), (7, 263), (16, 512), (8, 1024), (33, 2048), (5, 4096)] + MODEL_SHAPES)
def test_gemv_qk256_exact_kernel_bit_identical(hip, pkg, oracle, rows, cols):
    """Reference-order kernel == scalar reference (Q/i2s_qk256.rs:196-274), every bit."""
    qs, x, stride = qk256_inputs(rows, cols, 42)
    hip.set_kernel(pkg.KERNEL_EXACT)
    got = hip.gemv_qk256(qs, x, rows, cols, stride)
    want = oracle.gemv_qk256(qs, x, rows, cols, stride, impl="scalar")
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kernel", STREAMING)
@pytest.mark.parametrize("rows,cols", [(4, 256), (3, 300), (7, 263), (16, 512), (8, 1024), (33, 2048), (5, 4096)] + MODEL_SHAPES)
def test_gemv_qk256_streaming_parity(hip, pkg, oracle, kernel, rows, cols):
    qs, x, stride = qk256_inputs(rows, cols, 42)
    _set(hip, pkg, kernel)
    got = hip.gemv_qk256(qs, x, rows, cols, stride)
    ref_scalar = oracle.gemv_qk256(qs, x, rows, cols, stride, impl="scalar")
    ref_avx2 = oracle.gemv_qk256(qs, x, rows, cols, stride, impl="dispatch")
    truth = oracle.gemv_qk256_f64(qs, x, rows, cols, stride)
    assert np.all(approx_eq_with_len(got, ref_scalar, cols))
    assert np.all(approx_eq_with_len(got, ref_avx2, cols))
    # no worse than 4x the reference AVX2 path's own distance from the truth (+1 ulp slack)
    err_gpu = np.max(np.abs(got - truth))
    err_ref = max(np.max(np.abs(ref_avx2 - truth)), np.max(np.abs(ref_scalar - truth)))
    assert err_gpu <= 4 * err_ref + np.max(np.abs(truth)) * 2.0**-23
    cos = float(np.dot(got.astype(np.float64), truth) / (np.linalg.norm(got) * np.linalg.norm(truth)))
    assert cos >= 0.99999


def ternary_inputs(n, k, bs, seed, m=1):
    """codes from {0,1,3} with P(0)=.5, P(+-1)=.25; scales
    1/((i%100)+1) (crates/bitnet-quantization/benches/qk256_gemv.rs:41-43)."""
    rng = np.random.default_rng(seed)
    codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(n, k), p=[0.5, 0.25, 0.25])
    kp = -(-k // 4)
    pad = np.zeros((n, kp * 4), np.uint8)
    pad[:, :k] = codes
    packed = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8).reshape(-1)
    nblk = -(-k // bs)
    scales = (1.0 / ((np.arange(n * nblk) % 100) + 1)).astype(np.float32)
    act = np.random.default_rng(seed + 1).uniform(-10, 10, m * k).astype(np.float32)
    return packed, scales, act


@pytest.mark.parametrize("n,k,bs,m", [(64, 256, 32, 1), (64, 256, 256, 1), (48, 1024, 32, 3), (31, 2560, 32, 1), (640, 2560, 32, 1), (2560, 2560, 256, 1), (129, 6912, 32, 2)])
def test_i2s_ternary_matmul_parity(hip, pkg, oracle, n, k, bs, m):
    packed, scales, act = ternary_inputs(n, k, bs, 42, m)
    want = oracle.i2s_matmul(act, packed, scales, m, n, k, bs)
    hip.set_kernel(pkg.KERNEL_EXACT)
    assert np.array_equal(hip.i2s_matmul_f32(act, packed, scales, m, n, k, bs), want)
    for kernel in STREAMING:
        _set(hip, pkg, kernel)
        got = hip.i2s_matmul_f32(act, packed, scales, m, n, k, bs)
        assert np.all(approx_eq_with_len(got, want, k)), kernel
        # ternary + small scales: also tight in absolute terms relative to the output scale
        assert np.max(np.abs(got - want)) <= 1e-4 * max(1.0, np.max(np.abs(want))), kernel
    if bs == 256:  # the K/rocm/qk256_gemv.rs stub signature
        got = hip.qk256_gemv(packed, scales, act, m, n, k)
        assert np.all(approx_eq_with_len(got, want, k))


def test_device_handle_api_and_batched_rows(hip, pkg, oracle):
    """forward_qk256 (T:683-691) = one GEMV per activation row."""
    import torch

    rows, cols, m = 640, 2560, 5
    qs, _, stride = qk256_inputs(rows, cols, 3)
    X = np.random.default_rng(9).normal(0, 1, (m, cols)).astype(np.float32)
    h = hip.weights_upload_qk256(qs, rows, cols, stride)
    assert hip.weights_info(h) == (rows, cols, rows * stride)
    xd = torch.from_numpy(X).cuda()
    for kernel in KERNELS:
        _set(hip, pkg, kernel)
        yd = torch.full((m, rows), float("nan"), device="cuda")
        hip.matmul_dev(h, xd, yd, m)
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        for i in range(m):
            want = oracle.gemv_qk256(qs, X[i], rows, cols, stride, impl="scalar")
            if kernel == "exact":
                assert np.array_equal(got[i], want)
            assert np.all(approx_eq_with_len(got[i], want, cols))
    # runs on a caller stream too
    s = torch.cuda.Stream()
    yd = torch.zeros(rows, device="cuda")
    with torch.cuda.stream(s):
        hip.gemv_dev(h, xd[0], yd, stream=s.cuda_stream)
    s.synchronize()
    assert np.all(approx_eq_with_len(yd.cpu().numpy(), oracle.gemv_qk256(qs, X[0], rows, cols, stride), cols))
    hip.weights_free(h)
    with pytest.raises(pkg.BitNetHipError, match="unknown weights handle"):
        hip.gemv_dev(h, xd[0], yd)


# ----------------------------------------------------------- provider trait


def test_quantized_matmul_i2s_composite(hip, pkg, oracle):
    """The a8 composite (quantize_input_i2s -> matmul_i2s on raw codes -> block scale) against its oracle restatement, bit
    for bit (every intermediate is a small integer), on both device kernels; shapes from the model's projections."""
    rng = np.random.default_rng(12)
    for m, n, k, bs, per_feature in [(1, 8, 16, 32, True), (3, 640, 2560, 32, False), (2, 2560, 6912, 32, True), (5, 12, 20, 4, False), (1, 6, 10, 32, False)]:
        x = rng.normal(0, 1.2, m * k).astype(np.float32)
        x[::17] = np.round(x[::17]) + 0.5  # ties: f32::round goes away from zero
        packed = rng.integers(0, 256, -(-k * n // 4), dtype=np.uint8)
        scales = rng.uniform(0.1, 2.0, n if per_feature else max(1, k * n // bs // 3)).astype(np.float32)
        want = oracle.quantized_matmul_i2s(x, packed, scales, bs, m, n, k)
        for kern in (pkg.KERNEL_AUTO, pkg.KERNEL_EXACT):
            hip.set_kernel(kern)
            assert np.array_equal(hip.quantized_matmul_i2s(x, packed, scales, bs, m, n, k), want), (m, n, k, kern)
    hip.set_kernel(pkg.KERNEL_AUTO)
    with pytest.raises(pkg.BitNetHipError, match="Matrix B dimension mismatch"):
        hip.quantized_matmul_i2s(np.zeros(4, np.float32), np.zeros(1, np.uint8), np.ones(2, np.float32), 32, 1, 2, 4)


def test_matmul_i2s_tiled_full_byte_ranges(hip, pkg, oracle):
    """The integer-tile kernel on the trait's FULL operand ranges (a in [-128, 127], b in [0, 255]) at k where every f32
    partial sum of the reference is still exact (128 * 255 * k < 2^24): identical to the scalar oracle."""
    rng = np.random.default_rng(13)
    for m, n, k in [(1, 4, 4), (9, 64, 512), (2, 4096, 256), (17, 12, 64)]:
        a = rng.integers(-128, 128, m * k).astype(np.int8)
        b = rng.integers(0, 256, k * n).astype(np.uint8)
        assert np.array_equal(hip.matmul_i2s(a, b, m, n, k), oracle.matmul_i2s(a, b, m, n, k)), (m, n, k)
    # I2_S value ranges at the model's shapes (the call the composite makes)
    a = rng.integers(-2, 2, 4 * 6912).astype(np.int8)
    b = rng.integers(0, 4, 6912 * 2560).astype(np.uint8)
    assert np.array_equal(hip.matmul_i2s(a, b, 4, 2560, 6912), oracle.matmul_i2s(a, b, 4, 2560, 6912))


def test_matmul_i2s_provider_exact(hip, oracle):
    """K/cpu/fallback.rs:306-318 + random: integer-valued, must be bit-exact."""
    assert hip.matmul_i2s([1, 2, 3, 4], [1, 0, 0, 1], 2, 2, 2).tolist() == [1.0, 2.0, 3.0, 4.0]
    rng = np.random.default_rng(5)
    for m, n, k in [(1, 1, 1), (3, 5, 7), (8, 64, 256), (4, 640, 2560)]:
        a = rng.integers(-2, 2, m * k).astype(np.int8)  # clamp(x,-2,1).round() range (quantized_linear.rs:1762-1773)
        b = rng.integers(0, 4, k * n).astype(np.uint8)
        assert np.array_equal(hip.matmul_i2s(a, b, m, n, k), oracle.matmul_i2s(a, b, m, n, k))


def test_quantize_i2s_bit_exact(hip, oracle):
    """K/cpu/fallback.rs:102-159: codes and scales bit-exact, ragged tail included."""
    out, scales = hip.quantize([1.5, -1.0, 0.5, -0.5, 0.0, 2.0, -2.0, 0.1])
    assert out.tolist() == [1 | (3 << 2), (1 << 2) | (3 << 4)] and np.isclose(scales[0], 2.0 / 1.5)
    rng = np.random.default_rng(11)
    for n in (32, 64, 100, 4096, 65536, 37):
        x = rng.normal(0, 1, n).astype(np.float32)
        x[rng.integers(0, n, max(1, n // 10))] = 0.0
        if n >= 64:
            x[32:64] = 0.0  # all-zero block -> scale 1.0
        go, gs = hip.quantize(x, out_len=-(-n // 4))
        wo, ws = oracle.quantize_i2s(x, out_len=-(-n // 4))
        assert np.array_equal(go, wo) and np.array_equal(gs, ws), n
    # OR-pack semantics (:153): existing bits survive
    init = np.full(8, 0x80, np.uint8)
    go, _ = hip.quantize(np.zeros(32, np.float32), out_init=init)
    assert np.array_equal(go, init)


def test_dequant_i2s_block_bit_exact(hip, oracle):
    """M/quant/i2s.rs dequant (Sym LUT x clamped |f16|): bit-exact, all block sizes,
    transposed and cfg forms, ragged tail."""
    rng = np.random.default_rng(13)
    for rows, cols, block in [(1, 256, 256), (4, 2560, 256), (3, 128, 128), (2, 64, 64), (2, 64, 32), (5, 2560, 32), (1, 40, 32)]:
        bpr = -(-cols // block)
        per = block // 4 + 2
        data = rng.integers(0, 256, rows * bpr * per, dtype=np.uint8)
        # plausible f16 scales incl. negative, tiny, huge, zero
        sc = rng.choice(np.array([0x3C00, 0x2E66, 0xB800, 0x0001, 0x7BFF, 0x0000, 0x1400, 0x5640], np.uint16), rows * bpr)
        blocks = data.reshape(rows * bpr, per)
        blocks[:, -2] = sc & 0xFF
        blocks[:, -1] = sc >> 8
        data = blocks.reshape(-1)
        if oracle.i2s_infer_block_size(data.size, rows, cols) != block:
            continue
        for inv, k, tr in [(False, 1.0, False), (False, 1.0, True), (True, 1.0, False), (False, 0.5, True)]:
            got = hip.dequant_i2s(data, rows, cols, inv=inv, k=k, transposed=tr)
            want = oracle.i2s_dequantize_to_f32(data, rows, cols, inv=inv, k=k, transposed=tr)
            assert np.array_equal(got, want), (rows, cols, block, inv, k, tr)
    # M/quant/i2s.rs:922-945 on the device
    assert hip.dequant_i2s([0xE4] * 8 + [0x00, 0x3C], 1, 32).tolist()[:4] == [-2, -1, 1, 2]
    with pytest.raises(Exception, match="byte length mismatch"):
        hip.dequant_i2s([0xE4] * 7, 1, 32)


# ------------------------------------- full-size, size-independent properties


def test_full_size_linearity_and_row_independence(hip, pkg):
    """At bitnet-b1.58-2B-4T sizes the oracle is too slow to run per case, so use
    properties: W(a*x + b*z) == a*Wx + b*Wz on exactly-representable data (exact
    in f32: small integers), row slicing commutes with the GEMV, and the all-ones
    vector returns the row sums of the decoded weights (integer-exact)."""
    import torch

    for kernel in STREAMING:
        _set(hip, pkg, kernel)
        for rows, cols in MODEL_SHAPES:
            qs, _, stride = qk256_inputs(rows, cols, 100 + rows % 7)
            h = hip.weights_upload_qk256(qs, rows, cols, stride)
            rng = np.random.default_rng(1)
            x = rng.integers(-8, 9, cols).astype(np.float32)
            z = rng.integers(-8, 9, cols).astype(np.float32)
            xs = torch.from_numpy(np.stack([x, z, 3 * x - 2 * z, np.ones(cols, np.float32)])).cuda()
            ys = torch.empty(4, rows, device="cuda")
            hip.matmul_dev(h, xs, ys, 4)
            torch.cuda.synchronize()
            y = ys.cpu().numpy()
            assert np.array_equal(y[2], 3 * y[0] - 2 * y[1])  # integers < 2^24: exact
            codes = ((qs.reshape(rows, stride, 1) >> np.array([0, 2, 4, 6])) & 3).reshape(rows, -1)[:, :cols]
            rowsum = np.array([-2, -1, 1, 2], np.int64)[codes].sum(1)
            assert np.array_equal(y[3].astype(np.int64), rowsum)
            # a row slice of W gives the same rows of y
            r0, r1 = rows // 3, rows // 3 + 37
            h2 = hip.weights_upload_qk256(qs[r0 * stride : r1 * stride], r1 - r0, cols, stride)
            y2 = torch.empty(r1 - r0, device="cuda")
            hip.gemv_dev(h2, xs[0], y2)
            torch.cuda.synchronize()
            assert np.array_equal(y2.cpu().numpy(), y[0, r0:r1])
            hip.weights_free(h)
            hip.weights_free(h2)


@pytest.mark.gpu
def test_measured_read_ceiling_is_plausible(hip):
    """bitnet_hip_hbm_read_ceiling: a read-only stream over 1 GiB lands between 2 and 8.2 TB/s on an MI355X."""
    best, mean = hip.hbm_read_ceiling(1 << 30, 5)
    assert 2000.0 < mean <= best < 8200.0, (best, mean)


def test_gemv_qk256_property(hip, pkg, oracle):
    """Property test in the spirit of crates/bitnet-models/tests/qk256_property_tests.rs: any rows / cols (ragged
    tails included), uniform codes, activations in [-10, 10] (qk256_avx2_correctness.rs:72-104) -- the device GEMV stays
    within the reference's own scalar-vs-AVX2 tolerance of the scalar oracle, and is linear in the activations."""
    from hypothesis import given, settings, strategies as st

    _set(hip, pkg, "auto")

    @settings(max_examples=40, deadline=None)
    @given(rows=st.integers(1, 80), cols=st.integers(1, 2200), seed=st.integers(0, 2**31 - 1), scale=st.sampled_from([1e-3, 1.0, 250.0]))
    def run(rows, cols, seed, scale):
        stride = -(-cols // 256) * 64
        rng = np.random.default_rng(seed)
        qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
        x = (rng.uniform(-10, 10, cols) * scale).astype(np.float32)
        got = hip.gemv_qk256(qs, x, rows, cols, stride)
        want = oracle.gemv_qk256(qs, x, rows, cols, stride, impl="scalar")
        assert np.all(approx_eq_with_len(got / scale, want / scale, cols)), (rows, cols, seed, scale)
        # homogeneity: an exact power-of-two factor on x scales y exactly (fixed point with a power-of-two scale)
        got2 = hip.gemv_qk256(qs, x * np.float32(4.0), rows, cols, stride)
        assert np.array_equal(got2, got * np.float32(4.0)), (rows, cols, seed, scale)

    run()


def test_fused_entries_on_shapes_outside_the_mfma_gemv(hip, pkg, oracle):
    """gemv_fused_dev / matmul_fused_dev on matrices the fused MFMA GEMV does not take (32-element scales with cols % 256 != 0):
    the library composes LayerNorm rows -> product -> silu * mul / residual add from separate device launches -- the reference's
    own op order -- instead of refusing.  Found by tools/random_sweep.py."""
    import torch

    rng = np.random.default_rng(77)

    def ternary(rows, cols):
        codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(rows, cols), p=[0.5, 0.25, 0.25])
        packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8).reshape(-1)
        scales = rng.uniform(0.05, 1.5, rows * (cols // 32)).astype(np.float16).astype(np.float32)
        return packed, scales

    for rows, cols, m, ln, res in [(100, 64, 3, True, True), (16, 288, 1, False, True), (333, 32, 2, True, False), (1, 288, 17, True, True)]:
        packed, scales = ternary(rows, cols)
        h = hip.weights_upload_i2s(packed, scales, rows, cols, 32)
        x = rng.uniform(-4, 4, (m, cols)).astype(np.float32)
        gam = rng.uniform(0.5, 1.5, cols).astype(np.float32)
        resid = rng.normal(0, 1, (m, rows)).astype(np.float32)
        xin = np.stack([oracle.layernorm(x[i], gam, 1e-5) for i in range(m)]) if ln else x
        want = oracle.i2s_matmul(xin.reshape(-1), packed, scales, m, rows, cols, 32).reshape(m, rows) + (resid if res else 0.0)
        xd, gd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(gam).cuda(), torch.from_numpy(resid).cuda()
        yd = torch.full((m, rows), float("nan"), device="cuda")
        if m == 1:
            hip.gemv_fused_dev(h, xd, yd, 1, gd if ln else None, 1e-5, rd if res else None, 0)
        else:
            wsb = hip.matmul_workspace_bytes(m, cols, 3)
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            hip.matmul_fused_dev(h, xd, yd, m, ws, wsb, ln_gamma=gd if ln else None, ln_eps=1e-5, residual=rd if res else None, digits=3)
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        assert np.max(np.abs(got - want)) <= 3e-5 * max(1.0, np.max(np.abs(want))) + 2e-4, (rows, cols, m)
        hip.weights_free(h)
    # silu(gate) * up on an interleaved pair of such matrices
    rows, cols = 32, 288
    (pg, sg), (pu, su) = ternary(rows, cols), ternary(rows, cols)
    hg, hu = hip.weights_upload_i2s(pg, sg, rows, cols, 32), hip.weights_upload_i2s(pu, su, rows, cols, 32)
    pair = hip.weights_concat([hg, hu], interleave16=True)
    x = rng.uniform(-2, 2, cols).astype(np.float32)
    g = oracle.i2s_matmul(x, pg, sg, 1, rows, cols, 32).astype(np.float64)
    u = oracle.i2s_matmul(x, pu, su, 1, rows, cols, 32).astype(np.float64)
    want = g / (1.0 + np.exp(-g)) * u
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((rows,), float("nan"), device="cuda")
    hip.gemv_fused_dev(pair, xd, yd, 1, None, 0.0, None, 1)  # BITNET_HIP_FUSE_SILU_MUL
    torch.cuda.synchronize()
    assert np.max(np.abs(yd.cpu().numpy() - want)) <= 3e-5 * max(1.0, np.max(np.abs(want)))
    for hh in (pair, hg, hu):
        hip.weights_free(hh)
