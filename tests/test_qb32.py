"""QB32 activations (round 5): producer-quantised rows for the fp6 x fp4 prompt matmul -- block-scaled 15-bit fixed point, one exponent per 32
columns of a token (include/bitnet_hip.h).  The format is read back on the host (tests/qb32_ref.py), so every stage is held to exact statements:
  * bitnet_hip_rows_to_qb32_dev: each element within half a step of its unit (step = 2^(E - 13), E the exponent of the unit's maximum), the
    statistics partial = the row's (sum, sum of squares);
  * bitnet_hip_matmul_qb32_dev: the product of the DECODED rows with the matrix, to f32 accumulation accuracy -- plain, LayerNorm after the
    product, residual, silu * up as f16 rows -- and against the oracle's per-row loop (gemv_qk256, Q/i2s_qk256.rs:346) on the exact rows within
    the format's error model;
  * bitnet_hip_matmul_f16_dev(... BITNET_HIP_FUSE_YH_QB32): the o- / down-projection's epilogue hand-over decodes to gamma_out * y."""
import numpy as np
import pytest

from tests import qb32_ref

pytestmark = pytest.mark.gpu
LUT = np.array([-2.0, -1.0, 1.0, 2.0])
YH_QB32, SILU = 64, 1


@pytest.fixture(scope="module")
def torch_():
    import torch

    return torch


def dense_qk256(qs, rows, cols):
    p = qs.reshape(rows, cols // 4)
    return LUT[np.stack([(p >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)]


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.mark.parametrize("m,cols", [(70, 2560), (1, 256), (300, 6912)])
def test_rows_to_qb32_decodes_to_the_rows_within_half_a_step(hip, torch_, m, cols):
    rng = np.random.default_rng(m + cols)
    x = (rng.normal(0.2, 1.0, (m, cols)) * np.exp(rng.uniform(-6, 6, (m, 1)))).astype(np.float32)
    x[m // 2, 64:96] = 0.0  # an all-zero unit
    x[0, 5] = 3e4           # an outlier: only ITS unit loses resolution
    g = rng.uniform(0.5, 1.5, cols).astype(np.float32)
    qb = torch_.zeros(hip.qb32_bytes(m, cols), dtype=torch_.uint8, device="cuda")
    st = torch_.zeros(2 * (-(-m // 64) * 64), device="cuda")
    hip.rows_to_qb32_dev(torch_.from_numpy(x).cuda(), torch_.from_numpy(g).cuda(), m, cols, qb, st)
    torch_.cuda.synchronize()
    got, exps = qb32_ref.decode(qb.cpu().numpy(), m, cols)
    want = (x * g).astype(np.float32).astype(np.float64)
    step = qb32_ref.unit_lsb(exps)
    assert np.all(np.abs(got - want) <= 0.5 * step * (1 + 1e-6))
    # the step is 2^-13 of the power of two at or below the unit's maximum
    umax = np.abs(want).reshape(m, cols // 32, 32).max(axis=-1)
    nz = umax > 0
    assert np.all(np.exp2(exps[nz].astype(np.float64) - 117.0) <= umax[nz]) and np.all(umax[nz] < np.exp2(exps[nz].astype(np.float64) - 116.0))
    s = st.cpu().numpy().reshape(-1, 2)[:m]
    assert np.allclose(s[:, 0], x.astype(np.float64).sum(axis=1), rtol=1e-6, atol=1e-3) and np.allclose(s[:, 1], (x.astype(np.float64) ** 2).sum(axis=1), rtol=1e-6)


@pytest.mark.parametrize("rows,cols,m", [(2560, 2560, 4096), (512, 1024, 70), (3840, 2560, 300), (256, 6912, 33)])
def test_matmul_qb32_is_the_product_of_the_decoded_rows(hip, oracle, torch_, rows, cols, m):
    rng = np.random.default_rng(rows + cols + m)
    stride = cols // 256 * 64
    qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
    W = dense_qk256(qs, rows, cols)
    x = (rng.normal(0.1, 1.0, (m, cols)) * np.exp(rng.uniform(-3, 3, (m, 1)))).astype(np.float32)
    g = (rng.uniform(0.5, 1.5, cols) / 80).astype(np.float32)
    h = hip.weights_upload_qk256(qs, rows, cols, stride)
    assert hip.matmul_qb32_supported(h)
    gd = torch_.from_numpy(g).cuda()
    hip.weights_bind_ln(h, gd)
    xd = torch_.from_numpy(x).cuda()
    m_pad = -(-m // 64) * 64
    pick = np.unique(np.r_[0, m - 1, rng.integers(0, m, 10)])
    # ---- plain: y = W . decode(QB32(x)), + residual
    qb = torch_.zeros(hip.qb32_bytes(m, cols), dtype=torch_.uint8, device="cuda")
    hip.rows_to_qb32_dev(xd, None, m, cols, qb, None)
    res = rng.normal(0, 1, (m, rows)).astype(np.float32)
    y = torch_.full((m, rows), float("nan"), device="cuda")
    hip.matmul_qb32_dev(h, qb, m, y=y, residual=torch_.from_numpy(res).cuda())
    torch_.cuda.synchronize()
    t = hip.matmul_last_tile()
    assert t["scale_mode"] == 8 and t["digits"] == 2 and hip.matmul_last_resident_fp4(), t
    dec, exps = qb32_ref.decode(qb.cpu().numpy(), m, cols)
    got = y.cpu().numpy().astype(np.float64) - res
    want = dec[pick] @ W.T
    bound = 4e-6 * (np.abs(dec[pick]) @ np.abs(W).T) + 1e-6 * np.abs(res[pick])  # f32 accumulation of exact products (+ the residual's own rounding)
    assert not np.isnan(got).any()
    assert np.all(np.abs(got[pick] - want) <= bound + 1e-30), float(np.max(np.abs(got[pick] - want) / (bound + 1e-30)))
    # ... and the oracle's per-row loop on the EXACT rows, within the format's error model: each element errs by a uniform half step
    sig = np.sqrt(((qb32_ref.unit_lsb(exps)[pick] ** 2 / 12.0) @ (W ** 2).T))
    for k, i in enumerate(pick):
        ref = oracle.gemv_qk256(qs, x[i], rows, cols, stride).astype(np.float64)
        assert np.all(np.abs(got[i] - ref) <= 7 * sig[k] + 2e-5 * np.abs(ref).max()), int(i)
        assert cosine(got[i], ref) >= 0.999999, int(i)
    # ---- LayerNorm after the product: QB32(gamma * x) + the statistics partial
    st = torch_.zeros(2 * m_pad, device="cuda")
    hip.rows_to_qb32_dev(xd, gd, m, cols, qb, st)
    yl = torch_.full((m, rows), float("nan"), device="cuda")
    hip.matmul_qb32_dev(h, qb, m, stats_in=st, n_stats=1, ln_gamma=gd, ln_eps=1e-5, y=yl)
    torch_.cuda.synchronize()
    gotl = yl.cpu().numpy()
    for i in pick:
        ref = oracle.gemv_qk256(qs, oracle.layernorm(x[i], g, 1e-5), rows, cols, stride).astype(np.float64)
        assert cosine(gotl[i], ref) >= 0.99999, int(i)
        assert np.max(np.abs(gotl[i] - ref)) <= 2e-3 * np.abs(ref).max(), int(i)
    hip.weights_free(h)


def test_matmul_qb32_silu_pair_writes_f16_rows(hip, oracle, torch_):
    rng = np.random.default_rng(12)
    K, F, m = 1024, 768, 130
    stride = K // 256 * 64
    qg, qu = (rng.integers(0, 256, F * stride, dtype=np.uint8) for _ in range(2))
    hg, hu = hip.weights_upload_qk256(qg, F, K, stride), hip.weights_upload_qk256(qu, F, K, stride)
    h = hip.weights_concat([hg, hu], interleave16=True)
    g = (rng.uniform(0.5, 1.5, K) / 40).astype(np.float32)
    gd = torch_.from_numpy(g).cuda()
    hip.weights_bind_ln(h, gd)
    x = rng.normal(0.0, 1.5, (m, K)).astype(np.float32)
    m_pad = -(-m // 64) * 64
    qb = torch_.zeros(hip.qb32_bytes(m, K), dtype=torch_.uint8, device="cuda")
    st = torch_.zeros(2 * m_pad, device="cuda")
    hip.rows_to_qb32_dev(torch_.from_numpy(x).cuda(), gd, m, K, qb, st)
    yh = torch_.full((m_pad, F), float("nan"), dtype=torch_.float16, device="cuda")
    hip.matmul_qb32_dev(h, qb, m, stats_in=st, n_stats=1, ln_gamma=gd, ln_eps=1e-5, flags=SILU, yh=yh)
    torch_.cuda.synchronize()
    got = yh.cpu().numpy()[:m].astype(np.float64)
    for i in (0, 57, m - 1):
        xn = oracle.layernorm(x[i], g, 1e-5)
        a, b = oracle.gemv_qk256(qg, xn, F, K, stride).astype(np.float64), oracle.gemv_qk256(qu, xn, F, K, stride).astype(np.float64)
        want = a / (1.0 + np.exp(-a)) * b
        assert cosine(got[i], want) >= 0.99999
        assert np.max(np.abs(got[i] - want)) <= 2e-3 * np.abs(want).max()
    with pytest.raises(Exception, match="FUSE_SILU_MUL"):
        hip.matmul_qb32_dev(h, qb, m, flags=SILU, yh=yh, residual=yh)
    for hh in (hg, hu, h):
        hip.weights_free(hh)


@pytest.mark.parametrize("fmt,rows,cols", [("qk256", 2560, 2560), ("qk256", 2560, 6912), ("i2s", 2560, 2560), ("qk256", 1024, 512)])
def test_f16_chain_epilogue_hands_over_qb32_rows(hip, torch_, fmt, rows, cols):
    """The producer: bitnet_hip_matmul_f16_dev with BITNET_HIP_FUSE_YH_QB32 (the o- / down-projection of the hybrid prompt forward, 320- and
    256-row workgroups): the QB32 buffer decodes to gamma_out * y of the same launch's f32 rows, to half a step; y, the statistics partials and
    the residual are untouched by the hand-over (compared with the launch without it)."""
    synth = __import__("importlib").import_module("bitnet-rs_amd.synth")
    rng = np.random.default_rng(rows + cols)
    m = 4096 if rows == 2560 else 16384  # 64-token tiles: rows / 256 x m / 64 workgroups must cover the chip
    if fmt == "qk256":
        stride = cols // 256 * 64
        h = hip.weights_upload_qk256(rng.integers(0, 256, rows * stride, dtype=np.uint8), rows, cols, stride)
    else:
        w, s = synth.ternary_weights(rows, cols, 32, 7, 3, 1)
        h = hip.weights_upload_i2s(w, s, rows, cols, 32)
    assert hip.matmul_f16_supported(h)
    xh = (torch_.randn(m, cols, device="cuda") * 0.5).half()
    res = torch_.randn(m, rows, device="cuda")
    gout = torch_.from_numpy(rng.uniform(0.5, 1.5, rows).astype(np.float32)).cuda()
    outs = []
    for flags in (0, YH_QB32):
        y = torch_.full((m, rows), float("nan"), device="cuda")
        so = torch_.zeros(rows // 64 * m * 2, device="cuda")
        if flags:
            yh = torch_.zeros(hip.qb32_bytes(m, rows), dtype=torch_.uint8, device="cuda")
        else:
            yh = torch_.zeros(m, rows, dtype=torch_.float16, device="cuda")
        hip.matmul_f16_dev(h, xh, m, y=y, residual=res, flags=flags, yh=yh, gamma_out=gout, stats_out=so)
        torch_.cuda.synchronize()
        outs.append((y.cpu().numpy(), so.cpu().numpy(), yh.cpu().numpy()))
    (y0, s0, yh0), (y1, s1, qb) = outs
    assert not np.isnan(y1).any() and np.array_equal(y0, y1) and np.array_equal(s0, s1)
    pick = np.unique(np.r_[0, m - 1, rng.integers(0, m, 40)])
    dec, exps = qb32_ref.decode(qb, m, rows)
    want = (y1 * gout.cpu().numpy()[None, :]).astype(np.float32).astype(np.float64)
    step = qb32_ref.unit_lsb(exps)
    assert np.all(np.abs(dec[pick] - want[pick]) <= 0.5 * step[pick] * (1 + 1e-6))
    assert np.all(np.abs(dec - want) <= 0.5 * step * (1 + 1e-6))
    hip.weights_free(h)
