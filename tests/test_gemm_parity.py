"""GPU parity of the many-row (prefill) matmul, kernels_gemm.hip, through the C ABI
(bitnet_hip_matmul_fused_dev / bitnet_hip_matmul_dev) against the CPU oracle's row-by-row
restatement of gemv_qk256 (Q/i2s_qk256.rs:196-321) and i2s_matmul_f32
(K/cpu/quantized_matmul.rs:57-96).  Tolerance: the reference's own approx_eq_with_len
(crates/bitnet-models/tests/helpers/qk256_tolerance.rs) for 3 and 4 digits; 2 digits
(14-bit activations) is gated on cosine >= 0.99999 (benches/qk256_gemv.rs:234)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def approx_tol(cols):
    return min(2e-4 * np.sqrt(cols / 256.0), 1e-3)


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.fixture(scope="module")
def torch_():
    import torch

    return torch


def run_gemm(hip, torch_, h, x, out_cols, digits, **kw):
    m, k = x.shape
    wsb = hip.matmul_workspace_bytes(m, k, digits)
    ws = torch_.empty(wsb, dtype=torch_.uint8, device="cuda")
    xd = torch_.from_numpy(x).cuda()
    yd = torch_.full((m, out_cols), float("nan"), device="cuda")
    hip.matmul_fused_dev(h, xd, yd, m, ws, wsb, digits=digits, **kw)
    torch_.cuda.synchronize()
    return yd.cpu().numpy()


@pytest.mark.parametrize("rows,cols,m", [(256, 256, 16), (640, 2560, 64), (2560, 2560, 100), (1000, 300, 37), (320, 6912, 130), (48, 512, 1)])
def test_qk256_gemm_matches_oracle_rows(hip, oracle, torch_, rows, cols, m):
    rng = np.random.default_rng(rows + cols + m)
    stride = -(-cols // 256) * 64
    qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
    x = rng.uniform(-10, 10, (m, cols)).astype(np.float32)  # crates/bitnet-models/tests/qk256_avx2_correctness.rs:101-104
    x[m // 2] *= 1e-3  # rows of very different magnitude: the scale is per row
    want = np.stack([oracle.gemv_qk256(qs, x[i], rows, cols, stride) for i in range(m)])
    h = hip.weights_upload_qk256(qs, rows, cols, stride)
    for digits in (4, 3, 2):
        got = run_gemm(hip, torch_, h, x, rows, digits)
        assert not np.isnan(got).any()
        if digits >= 3:
            # approx_eq_with_len: abs tol 2e-4*sqrt(cols/256) (<= 1e-3) or 2 % relative; the scalar
            # reference itself rounds ~1e-7*sum|x| per element, so the test scales the abs tol by the row's magnitude
            scale = np.maximum(1.0, np.abs(x).sum(axis=1, keepdims=True) * 2e-7 / approx_tol(cols))
            assert np.all((np.abs(got - want) <= approx_tol(cols) * scale) | (np.abs(got - want) <= 2e-2 * np.abs(want))), (digits, np.abs(got - want).max())
        for i in range(m):
            assert cosine(got[i], want[i]) >= 0.99999, (digits, i)
    # 4 digits agrees with the GEMV path (same arithmetic, per-wave instead of per-row scale) to f32 rounding
    xd = torch_.from_numpy(x).cuda()
    yv = torch_.empty(m, rows, device="cuda")
    for i in range(m):
        hip.gemv_dev(h, xd[i], yv[i])
    torch_.cuda.synchronize()
    got4 = run_gemm(hip, torch_, h, x, rows, 4)
    denom = np.abs(x).sum(axis=1, keepdims=True) * 2.0
    assert np.max(np.abs(got4 - yv.cpu().numpy()) / denom) <= 3e-7
    # bitnet_hip_matmul_dev takes the same path for m >= 16
    yd = torch_.empty(m, rows, device="cuda")
    hip.matmul_dev(h, xd, yd, m)
    torch_.cuda.synchronize()
    if m >= 16:
        assert np.array_equal(yd.cpu().numpy(), got4)
    hip.weights_free(h)


def test_gemm_fusions_ln_residual_silu(hip, oracle, torch_):
    rng = np.random.default_rng(77)
    K, N, m = 2560, 512, 48
    stride = K // 256 * 64
    qa = rng.integers(0, 256, N * stride, dtype=np.uint8)
    qb = rng.integers(0, 256, N * stride, dtype=np.uint8)
    x = rng.normal(0.1, 1.0, (m, K)).astype(np.float32)
    g = (rng.uniform(0.5, 1.5, K) / 80).astype(np.float32)
    res = rng.normal(0, 1, (m, N)).astype(np.float32)
    ha, hb = hip.weights_upload_qk256(qa, N, K, stride), hip.weights_upload_qk256(qb, N, K, stride)
    gd, rd = torch_.from_numpy(g).cuda(), torch_.from_numpy(res).cuda()
    xn = np.stack([oracle.layernorm(x[i], g, 1e-5) for i in range(m)])
    ya = np.stack([oracle.gemv_qk256(qa, xn[i], N, K, stride) for i in range(m)])
    yb = np.stack([oracle.gemv_qk256(qb, xn[i], N, K, stride) for i in range(m)])
    tol = lambda want: 3e-5 * np.max(np.abs(want)) + 1e-6
    for digits in (4, 3):
        got = run_gemm(hip, torch_, ha, x, N, digits, ln_gamma=gd, ln_eps=1e-5, residual=rd)
        assert np.max(np.abs(got - (ya + res))) <= tol(ya), digits
    hc = hip.weights_concat([ha, hb])
    got = run_gemm(hip, torch_, hc, x, 2 * N, 4, ln_gamma=gd, ln_eps=1e-5)
    assert np.max(np.abs(got - np.concatenate([ya, yb], axis=1))) <= tol(ya)
    hg = hip.weights_concat([ha, hb], interleave16=True)
    got = run_gemm(hip, torch_, hg, x, N, 4, ln_gamma=gd, ln_eps=1e-5, flags=1)
    want = (ya / (1 + np.exp(-ya.astype(np.float64)))).astype(np.float32) * yb
    assert np.max(np.abs(got - want)) <= 3e-5 * np.max(np.abs(want)) + 1e-6
    # in place: y aliases the residual (x = x + W h)
    yd = rd.clone()
    wsb = hip.matmul_workspace_bytes(m, K, 4)
    ws = torch_.empty(wsb, dtype=torch_.uint8, device="cuda")
    hip.matmul_fused_dev(ha, torch_.from_numpy(x).cuda(), yd, m, ws, wsb, ln_gamma=gd, ln_eps=1e-5, residual=yd)
    torch_.cuda.synchronize()
    assert np.max(np.abs(yd.cpu().numpy() - (ya + res))) <= tol(ya)
    with pytest.raises(Exception, match="workspace too small"):
        hip.matmul_fused_dev(ha, rd, yd, m, ws, 16)
    with pytest.raises(Exception, match="digits must be"):
        hip.matmul_fused_dev(ha, rd, yd, m, ws, wsb, digits=5)
    for h in (ha, hb, hc, hg):
        hip.weights_free(h)


@pytest.mark.parametrize("block,f16_scales,n,k,m", [(256, False, 384, 1024, 40), (32, False, 384, 1024, 40), (32, True, 384, 1024, 40),
                                                     (32, True, 1000, 2560, 70), (32, True, 48, 256, 1)])
def test_ternary_scaled_gemm(hip, oracle, torch_, block, f16_scales, n, k, m):
    """i2s_matmul_f32 semantics with m rows.  256-element scales fold once per 256-block; 32-element scales that are f16
    values (what a GGUF I2_S file holds) run on the K = 32 MFMA with f16 scale tiles, other 32-element scales on the masked
    K = 64 form with row-major f32 scales."""
    rng = np.random.default_rng(block + n)
    codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(n, k), p=[0.5, 0.25, 0.25])
    packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8)
    scales = (1.0 / ((np.arange(n * (k // block)) % 100) + 1)).astype(np.float32)
    if f16_scales:
        scales = scales.astype(np.float16).astype(np.float32)
    x = rng.uniform(-4, 4, (m, k)).astype(np.float32)
    want = oracle.i2s_matmul(x.reshape(-1), packed.reshape(-1), scales, m, n, k, block).reshape(m, n)
    h = hip.weights_upload_i2s(packed.reshape(-1), scales, n, k, block)
    for digits in (4, 3):
        got = run_gemm(hip, torch_, h, x, n, digits)
        assert np.max(np.abs(got - want)) <= 2e-5 * np.max(np.abs(want)) + 1e-6, (block, digits)
    got = run_gemm(hip, torch_, h, x, n, 2)
    if f16_scales and block == 32 and k % 256 == 0:
        # BitNet32-F16 at digits = 2 runs on the f16 matrix cores with f16 ACTIVATIONS (north_star's "f16 activation dot product"):
        # every element m 2^e is rounded to +-2^(e-11) (uniform; relative to the element, averaged over m: rms 2^-11.5 / sqrt(3)), so an
        # output errs by sigma = 2^-11.5 / sqrt(3) * sqrt(sum_k (s_k x_k)^2); the gate is 7 sigma per output
        assert hip.matmul_last_tile()["scale_mode"] == 4
        s_full = np.repeat(scales.reshape(n, k // block), block, axis=1) * (codes != 0)
        sigma = 2.0 ** -11.5 / np.sqrt(3.0) * np.sqrt((x.astype(np.float64) ** 2) @ (s_full.astype(np.float64) ** 2).T)
        assert np.all(np.abs(got - want) <= 7.0 * sigma + 1e-6), float(np.max(np.abs(got - want) / (sigma + 1e-30)))
    else:  # 16-bit fixed-point activations: 2^-14 of each row's maximum per element
        assert np.max(np.abs(got - want)) <= 2e-4 * np.max(np.abs(want)) + 1e-6, (block, 2)
    for i in range(m):
        assert cosine(got[i], want[i]) >= 0.99999, (block, 2, i)
    if block == 32 and f16_scales and k % 256 == 0:
        # the K = 32 path reads the GEMV tiles and the f16 scale tiles: no second copy of the codes, no row-major scales
        code_bytes = -(-n // 16) * (k // 256) * 1024
        assert hip.weights_device_bytes(h) < 2 * code_bytes
    hip.weights_free(h)


def test_full_size_gemm_properties(hip, torch_):
    """BASELINE shapes (gate|up 13824 x 2560, 1024 rows): properties that need no oracle.
    * row independence: a row's result does not depend on the other rows of the batch;
    * exact power-of-two scaling: scaling a row by 2^k scales its fixed-point image exactly;
    * zero rows give exact zeros; the 3-digit result stays within 2^-20 of the 4-digit one (relative to the row's scale)."""
    rng = np.random.default_rng(123)
    n, k, m = 13824, 2560, 1024
    stride = k // 256 * 64
    h = hip.weights_upload_qk256(rng.integers(0, 256, n * stride, dtype=np.uint8), n, k, stride)
    x = rng.normal(0, 1, (m, k)).astype(np.float32)
    x[7] = 0.0
    y4 = run_gemm(hip, torch_, h, x, n, 4)
    assert not np.isnan(y4).any() and not y4[7].any()
    perm = rng.permutation(m)
    yp = run_gemm(hip, torch_, h, x[perm], n, 4)
    assert np.array_equal(yp, y4[perm])                                   # rows are independent, bit for bit
    xs = x.copy()
    xs[::2] *= 8.0
    xs[1::2] *= 0.25
    ys = run_gemm(hip, torch_, h, xs, n, 4)
    assert np.array_equal(ys[::2], y4[::2] * 8.0) and np.array_equal(ys[1::2], y4[1::2] * 0.25)
    y3 = run_gemm(hip, torch_, h, x, n, 3)
    bound = np.abs(x).max(axis=1, keepdims=True) * 2.0 ** -21 * 2 * k     # <= 2^-22 of the row maximum per element, |w| <= 2
    assert np.all(np.abs(y3 - y4) <= bound + 1e-6)
    hip.weights_free(h)


def test_digits2_with_a_code_map_outside_the_f16_weight_form(hip, torch_):
    """bitnet_hip_weights_upload_coded takes ANY int8 code map; k_gemm_f16w (digits = 2 on f16-exact 32-block scales)
    can only build code values -2 .. 2 (and needs 2 |s| finite in f16).  Any other matrix must keep the exact int8 digit form
    instead of producing NaN / inf: {-3, -1, 1, 3}, and a map the f16 form does take but with scales beyond its range."""
    rng = np.random.default_rng(5)
    n, k, m, block = 256, 512, 40, 32
    codes = rng.integers(0, 4, (n, k), dtype=np.uint8)
    packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8)
    x = rng.uniform(-4, 4, (m, k)).astype(np.float32)
    for cmap, smax, f16w in (((-3, -1, 1, 3), 1.0, False), ((-2, -1, 1, 2), 40000.0, False), ((-2, -1, 1, 2), 1.0, True)):
        scales = (smax / ((np.arange(n * (k // block)) % 7) + 1)).astype(np.float16).astype(np.float32)
        wfull = np.asarray(cmap, np.float64)[codes] * np.repeat(scales.reshape(n, k // block), block, axis=1)
        want = x.astype(np.float64) @ wfull.T
        h = hip.weights_upload_coded(packed.reshape(-1), scales, n, k, block, cmap)
        got = run_gemm(hip, torch_, h, x, n, 2)
        assert np.isfinite(got).all(), cmap
        assert (hip.matmul_last_tile()["scale_mode"] == 4) == f16w, (cmap, smax, hip.matmul_last_tile())
        assert np.max(np.abs(got - want)) <= 1e-3 * np.max(np.abs(want)), (cmap, smax)
        for i in range(m):
            assert cosine(got[i], want[i]) >= 0.99999
        hip.weights_free(h)


def test_matmul_on_block256_scales_keeps_one_copy_of_the_codes(hip, torch_):
    """a matrix with 256-element block scales reads its row-major scales in the prefill matmul; taking the pin must not
    re-materialise the trimmed row-major CODES (a second copy of the weights, a hipMalloc and a host synchronisation in the call)."""
    rng = np.random.default_rng(6)
    n, k, m, block = 512, 1024, 64, 256
    codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(n, k))
    packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8)
    scales = rng.uniform(0.5, 1.5, n * (k // block)).astype(np.float32)
    h = hip.weights_upload_i2s(packed.reshape(-1), scales, n, k, block)
    x = rng.normal(0, 1, (m, k)).astype(np.float32)
    run_gemm(hip, torch_, h, x, n, 4)  # (first use may build the streaming layout)
    hip.weights_trim(h)
    before = hip.weights_device_bytes(h)
    for digits in (2, 3, 4):
        run_gemm(hip, torch_, h, x, n, digits)
        assert hip.weights_device_bytes(h) == before, digits
    code_bytes = -(-n // 16) * (k // 256) * 1024
    assert before < 2 * code_bytes
    hip.weights_free(h)


def test_f16_handover_flags_of_the_digit_form(hip, torch_):
    """BITNET_HIP_FUSE_X_F16 / _Y_F16 (the prompt forward's hand-over on the int8 digit form): f16 input rows give, bit for bit, what the
    same values as f32 rows give; the f16 silu * up output is the f32 output rounded once.  Refused where they have no meaning."""
    rng = np.random.default_rng(8)
    K, N, m = 1024, 512, 70
    stride = K // 256 * 64
    qa = rng.integers(0, 256, N * stride, dtype=np.uint8)
    qb = rng.integers(0, 256, N * stride, dtype=np.uint8)
    ha, hb = hip.weights_upload_qk256(qa, N, K, stride), hip.weights_upload_qk256(qb, N, K, stride)
    hg = hip.weights_concat([ha, hb], interleave16=True)
    x16 = rng.normal(0, 1, (m, K)).astype(np.float16)
    x32 = x16.astype(np.float32)
    g = (rng.uniform(0.5, 1.5, K) / 80).astype(np.float32)
    gd = torch_.from_numpy(g).cuda()
    wsb = hip.matmul_workspace_bytes(m, K, 2)
    ws = torch_.empty(wsb, dtype=torch_.uint8, device="cuda")
    ya, yb = torch_.empty(m, N, device="cuda"), torch_.empty(m, N, device="cuda")
    hip.matmul_fused_dev(ha, torch_.from_numpy(x32).cuda(), ya, m, ws, wsb, digits=2)
    hip.matmul_fused_dev(ha, torch_.from_numpy(x16).cuda(), yb, m, ws, wsb, digits=2, flags=2)
    torch_.cuda.synchronize()
    assert np.array_equal(ya.cpu().numpy(), yb.cpu().numpy())
    yf = torch_.empty(m, N, device="cuda")
    yh = torch_.full((m, N), float("nan"), dtype=torch_.float16, device="cuda")
    hip.matmul_fused_dev(hg, torch_.from_numpy(x32).cuda(), yf, m, ws, wsb, ln_gamma=gd, ln_eps=1e-5, flags=1, digits=2)
    hip.matmul_fused_dev(hg, torch_.from_numpy(x32).cuda(), yh, m, ws, wsb, ln_gamma=gd, ln_eps=1e-5, flags=1 | 4, digits=2)
    torch_.cuda.synchronize()
    assert np.array_equal(yh.cpu().numpy(), yf.cpu().numpy().astype(np.float16))
    with pytest.raises(Exception, match="FUSE_X_F16"):
        hip.matmul_fused_dev(ha, ya, yb, m, ws, wsb, digits=2, flags=4)  # Y_F16 without SILU_MUL
    with pytest.raises(Exception, match="FUSE_X_F16"):
        hip.matmul_fused_dev(ha, ya, yb, m, ws, wsb, ln_gamma=gd, ln_eps=1e-5, digits=2, flags=2)  # X_F16 with a LayerNorm
    for h in (ha, hb, hg):
        hip.weights_free(h)


_F16A_SCRIPT = r"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
pkg = importlib.import_module("bitnet-rs_amd")
hip = pkg.load(); hip.init(0)
rng = np.random.default_rng(8)
K, N, m = 1024, 512, 70
stride = K // 256 * 64
qa = rng.integers(0, 256, N * stride, dtype=np.uint8)
ha = hip.weights_upload_qk256(qa, N, K, stride)
x16 = rng.normal(0, 1, (m, K)).astype(np.float16)
wsb = hip.matmul_workspace_bytes(m, K, 2)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
ya, yb = torch.empty(m, N, device="cuda"), torch.full((m, N), float("nan"), device="cuda")
hip.matmul_fused_dev(ha, torch.from_numpy(x16.astype(np.float32)).cuda(), ya, m, ws, wsb, digits=2)
assert hip.matmul_last_tile()["scale_mode"] == 5, hip.matmul_last_tile()   # k_gemm_f16a, unscaled (the env route was taken)
hip.matmul_fused_dev(ha, torch.from_numpy(x16).cuda(), yb, m, ws, wsb, digits=2, flags=2)
assert hip.matmul_last_tile()["scale_mode"] == 5
torch.cuda.synchronize()
a, b = ya.cpu().numpy(), yb.cpu().numpy()
assert np.array_equal(a, b), float(np.max(np.abs(a - b)))
# and against a float64 product of the same f16 values: one rounding to f16 per element (row maximum in [1, 2)), f32 accumulation
codes = np.stack([(qa.reshape(N, K // 4) >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(N, K)
W = np.array([-2.0, -1.0, 1.0, 2.0])[codes]
want = x16.astype(np.float64) @ W.T
assert np.max(np.abs(a - want)) <= 2e-3 * np.max(np.abs(want)), float(np.max(np.abs(a - want)))
print("F16A_X_F16_OK")
"""


def test_x_f16_handover_under_the_f16a_env_route_reads_f16_rows():
    """with BITNET_HIP_GEMM_F16A=1 an unscaled matrix at 2 digits goes to launch_gemm_f16, whose quantiser
    k_quant_rows_f16 ignored QuantArgs::x_f16 and read the f16 hand-over rows as f32 (garbage + reads past the allocation).  The
    switch is read once per process, hence the child process: f16 rows must give, bit for bit, what the same values as f32 rows
    give, on the f16 kernel (scale_mode 5 asserted)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _F16A_SCRIPT.format(root=root)], env=dict(os.environ, BITNET_HIP_GEMM_F16A="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "F16A_X_F16_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


INT8_DIGITS, FP6_DIGITS = 8, 16  # BITNET_HIP_FUSE_INT8_DIGITS, BITNET_HIP_FUSE_FP6_DIGITS


@pytest.mark.parametrize("rows,cols,m", [(256, 256, 16), (3840, 2560, 200), (2560, 2560, 4096), (1000, 1100, 37), (2560, 6912, 512), (48, 512, 1)])
def test_fp6_form_is_the_int8_two_digit_product_bit_for_bit(hip, oracle, torch_, rows, cols, m):
    """k_gemm_fp6 (BITNET_HIP_FUSE_FP6_DIGITS): the 2-digit form's 15-bit integer per activation as three balanced base-32 digits in
    fp6 (e2m3) x fp4 (e2m1) weights on v_mfma_scale_f32_16x16x128_f8f6f4, digit weights 1 / 32 / 1024 in the instruction's E8M0 block
    scale, one f32 accumulator.  Every product and partial sum is an integer below 2^24 on these inputs, so the result must equal the
    int8 digit planes' (BITNET_HIP_FUSE_INT8_DIGITS) bit for bit -- plain, behind the fused LayerNorm, with the residual, on odd shapes,
    on the 64-token / narrow / 320-row tiles -- and with it inherits that form's oracle gates (checked once more on sampled rows)."""
    rng = np.random.default_rng(rows * 7 + cols + m)
    stride = -(-cols // 256) * 64
    qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
    x = (rng.normal(0.1, 1.0, (m, cols)) * np.exp(rng.uniform(-3, 3, (m, 1)))).astype(np.float32)
    g = (rng.uniform(0.5, 1.5, cols) / 80).astype(np.float32)
    res = rng.normal(0, 1, (m, rows)).astype(np.float32)
    gd, rd = torch_.from_numpy(g).cuda(), torch_.from_numpy(res).cuda()
    h = hip.weights_upload_qk256(qs, rows, cols, stride)
    for kw in ({}, dict(ln_gamma=gd, ln_eps=1e-5), dict(residual=rd)):
        y8 = run_gemm(hip, torch_, h, x, rows, 2, flags=INT8_DIGITS, **kw)
        assert hip.matmul_last_tile()["scale_mode"] == 0
        y6 = run_gemm(hip, torch_, h, x, rows, 2, flags=FP6_DIGITS, **kw)
        t = hip.matmul_last_tile()
        assert t["scale_mode"] == 6 and t["digits"] == 2, t
        if (rows, m) == (2560, 4096):
            assert t["wave_tokens"] == 64 and hip.matmul_last_wave_rows() == 80  # 512 workgroups of 320 rows: one round of the chip
        assert not np.isnan(y6).any()
        assert np.array_equal(y8, y6), (sorted(kw), float(np.abs(y8 - y6).max()))
    pick = np.unique(np.r_[0, m - 1, rng.integers(0, m, 6)])
    want = np.stack([oracle.gemv_qk256(qs, x[i], rows, cols, stride) for i in pick])
    got = run_gemm(hip, torch_, h, x, rows, 2, flags=FP6_DIGITS)[pick]
    for i in range(len(pick)):
        assert cosine(got[i], want[i]) >= 0.99999, int(pick[i])
    hip.weights_free(h)


FP6_EXPAND = 32  # BITNET_HIP_FUSE_FP6_EXPAND


@pytest.mark.parametrize("rows,cols,m", [(2560, 2560, 4096), (512, 1024, 70), (13824, 2560, 300), (336, 512, 33), (3840, 2560, 4096), (2048, 512, 8130), (13824, 2560, 4096)])
def test_fp6_form_on_the_resident_fp4_image_is_the_expanding_form_bit_for_bit(hip, torch_, rows, cols, m):
    """Round 5: k_gemm_fp6<.., RES = 1> loads its A operands from the resident fp4 image (k_retile_fp4: exactly the nibbles expand16_fp4
    produces, stored once) -- the SAME operands, so the same bits as the in-loop expansion (BITNET_HIP_FUSE_FP6_EXPAND) and as the int8
    planes; the image is 4 bits per weight beside the 2-bit tiles (device bytes asserted), built once, freed on request."""
    rng = np.random.default_rng(rows + 3 * cols + m)
    stride = -(-cols // 256) * 64
    qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
    x = (rng.normal(0.1, 1.0, (m, cols)) * np.exp(rng.uniform(-3, 3, (m, 1)))).astype(np.float32)
    g = (rng.uniform(0.5, 1.5, cols) / 80).astype(np.float32)
    gd = torch_.from_numpy(g).cuda()
    h = hip.weights_upload_qk256(qs, rows, cols, stride)
    b0 = hip.weights_device_bytes(h)
    ye = run_gemm(hip, torch_, h, x, rows, 2, flags=FP6_DIGITS | FP6_EXPAND, ln_gamma=gd, ln_eps=1e-5)
    assert hip.matmul_last_tile()["scale_mode"] == 6 and not hip.matmul_last_resident_fp4()
    assert hip.weights_device_bytes(h) == b0  # the expanding form builds nothing
    yr = run_gemm(hip, torch_, h, x, rows, 2, flags=FP6_DIGITS, ln_gamma=gd, ln_eps=1e-5)  # first use builds the image
    assert hip.matmul_last_tile()["scale_mode"] == 6 and hip.matmul_last_resident_fp4()
    if (rows, m) in ((3840, 4096), (2048, 8130), (13824, 4096)):  # (the last: the benchmarked gate|up instance)  # full 64-token grids of 256-row workgroups: the 2 x 2 wave arrangement (k_gemm_fp6w) -- same bits
        import os
        assert hip.matmul_last_tile()["wave_tokens"] == 64 and hip.matmul_last_wave_rows() == (128 if os.environ.get("BITNET_HIP_GEMM_FP6W", "1") != "0" else 64)
    tiles = -(-rows // 16) * (stride // 64)
    assert hip.weights_device_bytes(h) == b0 + tiles * 2048
    y8 = run_gemm(hip, torch_, h, x, rows, 2, flags=INT8_DIGITS, ln_gamma=gd, ln_eps=1e-5)
    assert not np.isnan(yr).any()
    assert np.array_equal(ye, yr) and np.array_equal(y8, yr)
    hip.weights_fp4_image(h, True)   # idempotent
    assert hip.weights_device_bytes(h) == b0 + tiles * 2048
    hip.weights_fp4_image(h, False)
    assert hip.weights_device_bytes(h) == b0
    hip.weights_free(h)


def test_fp6_form_silu_handover_and_refusals(hip, torch_):
    """The fp6 form with the silu * mul epilogue and the f16 hand-over flags, against the int8 planes bit for bit; refused (never
    silently replaced) where it does not apply: other digit counts, together with FUSE_INT8_DIGITS, matrices with block scales (the
    instruction's block scale is a power of two; an unscaled matrix only comes from bitnet_hip_weights_upload_qk256)."""
    rng = np.random.default_rng(21)
    n, k, m = 512, 1024, 90
    x = rng.normal(0, 2, (m, k)).astype(np.float32)
    qa, qb = rng.integers(0, 256, n * k // 4, dtype=np.uint8), rng.integers(0, 256, n * k // 4, dtype=np.uint8)
    ha, hb = hip.weights_upload_qk256(qa, n, k, k // 4), hip.weights_upload_qk256(qb, n, k, k // 4)
    hg = hip.weights_concat([ha, hb], interleave16=True)
    g = (rng.uniform(0.5, 1.5, k) / 80).astype(np.float32)
    gd = torch_.from_numpy(g).cuda()
    wsb = hip.matmul_workspace_bytes(m, k, 2)
    ws = torch_.empty(wsb, dtype=torch_.uint8, device="cuda")
    xd = torch_.from_numpy(x).cuda()
    outs = {}
    for name, fl in (("int8", INT8_DIGITS), ("fp6", FP6_DIGITS)):
        yf = torch_.empty(m, n, device="cuda")
        yh = torch_.full((m, n), float("nan"), dtype=torch_.float16, device="cuda")
        yx = torch_.empty(m, n, device="cuda")
        hip.matmul_fused_dev(hg, xd, yf, m, ws, wsb, ln_gamma=gd, ln_eps=1e-5, flags=1 | fl, digits=2)
        hip.matmul_fused_dev(hg, xd, yh, m, ws, wsb, ln_gamma=gd, ln_eps=1e-5, flags=1 | 4 | fl, digits=2)
        hip.matmul_fused_dev(ha, xd.half(), yx, m, ws, wsb, flags=2 | fl, digits=2)
        torch_.cuda.synchronize()
        assert hip.matmul_last_tile()["scale_mode"] == (6 if name == "fp6" else 0)
        outs[name] = (yf.cpu().numpy(), yh.cpu().numpy(), yx.cpu().numpy())
    for a, b in zip(outs["int8"], outs["fp6"]):
        assert np.isfinite(b.astype(np.float32)).all()
        assert np.array_equal(a, b)
    ya = torch_.empty(m, n, device="cuda")
    with pytest.raises(Exception, match="FUSE_FP6_DIGITS"):
        hip.matmul_fused_dev(ha, xd, ya, m, ws, wsb, digits=3, flags=FP6_DIGITS)
    with pytest.raises(Exception, match="FUSE_FP6_DIGITS"):
        hip.matmul_fused_dev(ha, xd, ya, m, ws, wsb, digits=2, flags=FP6_DIGITS | INT8_DIGITS)
    codes = rng.integers(0, 4, (n, k), dtype=np.uint8)
    packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8)
    hs = hip.weights_upload_i2s(packed.reshape(-1), np.ones(n * (k // 32), np.float32), n, k, 32)
    with pytest.raises(Exception, match="FUSE_FP6_DIGITS"):
        hip.matmul_fused_dev(hs, xd, ya, m, ws, wsb, digits=2, flags=FP6_DIGITS)
    for h in (ha, hb, hg, hs):
        hip.weights_free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [2560, 1024, 1100])
def test_wave_per_row_quantiser_carries_the_documented_integer(hip, torch_, K):
    """k_quant_rows_w (round 5: 2 digits, rows of up to 2560 columns, both digit forms): q = rint(((x - mean) * gamma) * k) with
    k = 2^(13 - E) / denom and E the exponent of max |(x - mean) * gamma| / denom -- restated here in numpy f32 / f64, operation for
    operation.  Every product and partial sum behind the quantiser is an exact integer, so the launch must give f32(W q) * 2^(E - 13)
    EXACTLY wherever the model's q is the kernel's; the f64 statistics may be added up in another order (a mean one f32 ulp off moves a
    handful of q by one unit), hence: at least 99 % of the outputs bit-equal, every output within one unit of q per column."""
    import os
    if os.environ.get("BITNET_HIP_QUANT_W", "1") == "0":
        pytest.skip("the workgroup-per-row quantiser divides per element: another (equally valid) integer in rare ties")
    rng = np.random.default_rng(21 + K)
    N, m = 256, 2085  # (the launcher takes this quantiser from 2048 padded rows on; ragged: 27 padding rows)
    stride = (K + 255) // 256 * 64
    qs = rng.integers(0, 256, N * stride, dtype=np.uint8)
    h = hip.weights_upload_qk256(qs, N, K, stride)
    x = (rng.normal(0.3, 1.0, (m, K)) * np.exp(rng.normal(0, 1, (m, 1)))).astype(np.float32)
    g = rng.uniform(0.5, 1.5, K).astype(np.float32)
    eps = np.float32(1e-5)
    x64 = x.astype(np.float64)
    mean_d = x64.sum(1) / K
    var_d = (x64 * x64).sum(1) / K - mean_d * mean_d
    mean = mean_d.astype(np.float32)
    denom = np.sqrt(np.maximum(var_d, 0).astype(np.float32) + eps).astype(np.float32)
    t = (x - mean[:, None]) * g[None, :]
    assert t.dtype == np.float32
    am = (np.abs(t).max(1) / denom).astype(np.float32)
    be = np.maximum((am.view(np.uint32) >> 23) & 0xFF, 32).astype(np.int64)
    sc = (np.exp2((13 - (be - 127)).astype(np.float64)).astype(np.float32) / denom).astype(np.float32)
    inv_s = np.exp2(((be - 127) - 13).astype(np.float64)).astype(np.float32)
    q = np.rint(t * sc[:, None]).astype(np.int64)
    assert np.abs(q).max() <= 1 << 14
    codes = np.stack([(qs.reshape(N, stride) >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(N, stride * 4)[:, :K]
    W = np.array([-2, -1, 1, 2], dtype=np.int64)[codes]
    want = (q @ W.T).astype(np.float32) * inv_s[:, None]
    slack = np.abs(W).sum(1)[None, :].astype(np.float64) * inv_s[:, None]  # every q one unit off
    wsb = hip.matmul_workspace_bytes(m, K, 2)
    ws = torch_.empty(wsb, dtype=torch_.uint8, device="cuda")
    gd, xd = torch_.from_numpy(g).cuda(), torch_.from_numpy(x).cuda()
    outs = []
    for fl in (8, 16):  # BITNET_HIP_FUSE_INT8_DIGITS, BITNET_HIP_FUSE_FP6_DIGITS
        y = torch_.full((m, N), float("nan"), device="cuda")
        hip.matmul_fused_dev(h, xd, y, m, ws, wsb, ln_gamma=gd, ln_eps=float(eps), digits=2, flags=fl)
        torch_.cuda.synchronize()
        got = y.cpu().numpy()
        assert np.all(np.abs(got.astype(np.float64) - want) <= slack + 1e-30), float(np.max(np.abs(got - want) / (slack + 1e-30)))
        assert np.mean(got == want) >= 0.99, float(np.mean(got == want))
        outs.append(got)
    assert np.array_equal(outs[0], outs[1])  # the two digit forms carry the same integer
    hip.weights_free(h)
