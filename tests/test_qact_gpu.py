"""GPU: the decode step's GEMV on activations quantised by their producer ("QAct", csrc/qact.hpp, kernels_gemvq.hip).

 1. the quantiser is integer work -> BIT-EXACT against the numpy restatement below (tests/qact_ref.py);
 2. the GEMV on a given QAct against the real-number product with the DEQUANTISED activations (f64): tight;
 3. the whole chain (quantise -> [LayerNorm after the product] -> GEMV -> silu*mul / residual -> next QAct) against the
    oracle's f32 chain (oracle.layernorm -> oracle.gemv_qk256 / i2s_matmul) at the 2B-4T shapes, with the gates SURVEY.md
    8d names for reduced-precision activations: cosine >= 0.99999 per output vector, and the reference's
    approx_eq_with_len where it applies."""
import numpy as np
import pytest

from tests.qact_ref import QREC, dequantize_qact, quantize_qact

pytestmark = pytest.mark.gpu


def cosine(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.fixture(scope="module")
def torch_():
    import torch

    return torch


def dev(torch_, a):
    return torch_.from_numpy(np.ascontiguousarray(a)).cuda()


def gpu_quantize(hip, torch_, x, gamma=None, want_stats=False):
    n = x.size
    q = torch_.zeros(hip.qact_bytes(n), dtype=torch_.uint8, device="cuda")
    st = torch_.zeros(n // 16 * 2, dtype=torch_.float64, device="cuda") if want_stats else None
    hip.quantize_act_dev(dev(torch_, x), None if gamma is None else dev(torch_, gamma), n, q, st)
    torch_.cuda.synchronize()
    return q, st


@pytest.mark.parametrize("n", [16, 256, 2560, 6912, 272])
def test_quantiser_bit_exact(hip, torch_, n):
    rng = np.random.default_rng(n)
    cases = [rng.normal(0, 1, n), rng.normal(3, 50, n), rng.uniform(-1e-3, 1e-3, n), np.zeros(n), rng.normal(0, 1, n) * (rng.random(n) < 0.1)]
    big = rng.normal(0, 1, n)
    big[::16] = 2.0 ** rng.integers(-20, 20, n // 16)  # group maxima exactly on powers of two
    big[5::32] = -1e15                                  # and huge ones (squares still inside f32: the statistics are f32 sums)
    cases.append(big)
    tiny = rng.normal(0, 1e-30, n)
    cases.append(tiny)
    for ci, x in enumerate(cases):
        x = x.astype(np.float32)
        gamma = None if ci % 2 else rng.uniform(0.5, 1.5, n).astype(np.float32) / 80
        q, st = gpu_quantize(hip, torch_, x, gamma, want_stats=True)
        want = quantize_qact(x, gamma)
        got = q.cpu().numpy()
        assert got.size == (n + 255) // 256 * QREC
        assert np.array_equal(got, want), (n, ci)
        s = st.cpu().numpy().reshape(-1, 2)
        x64 = x.astype(np.float64).reshape(-1, 16)
        # f32 sums over the 16 values (stored as f64): ~1e-7 of the sum of magnitudes
        assert np.all(np.abs(s[:, 0] - x64.sum(1)) <= 1e-6 * np.abs(x64).sum(1) + 1e-300) and np.allclose(s[:, 1], (x64 * x64).sum(1), rtol=4e-6, atol=1e-36)  # f32 sums: squares below 1e-38 flush
        # and the format holds every element to 2^-15 of its group's maximum
        u = x if gamma is None else x * gamma
        back = dequantize_qact(want, n)
        gmax = np.abs(u).reshape(-1, 16).max(1).repeat(16)
        gmax = np.maximum(gmax, 2.0 ** -94)  # groups below 2^-95 share that exponent (scales stay normal floats)
        assert np.all(np.abs(back - u.astype(np.float64)) <= gmax * 2.0 ** -14)  # 2^-15 of 2^(E+1) > max


def dense_qk256(qs, rows, cols):
    p = qs.reshape(rows, cols // 4)
    codes = np.stack([(p >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
    return np.array([-2, -1, 1, 2], np.float64)[codes]


def dense_ternary(packed, scales, rows, cols):
    p = packed.reshape(rows, cols // 4)
    codes = np.stack([(p >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
    return np.array([0, 1, 0, -1], np.float64)[codes] * np.repeat(scales.reshape(rows, cols // 32).astype(np.float64), 32, axis=1)


@pytest.mark.parametrize("rows,cols,fmt", [(16, 256, "qk256"), (48, 512, "qk256"), (640, 2560, "qk256"), (2560, 6912, "qk256"), (3840, 2560, "qk256"),
                                           (32, 256, "f16"), (640, 2560, "f16"), (2560, 6912, "f16"), (2560, 2560, "f32"), (144, 1280, "f32")])
def test_gemv_q_against_dequantised_product(hip, torch_, rows, cols, fmt):
    """No LayerNorm: y must be W . dequantise(QAct) up to f32 accumulation rounding, for every K split / ring depth
    these shapes select, all three scale forms."""
    rng = np.random.default_rng(rows + cols)
    if fmt == "qk256":
        qs = rng.integers(0, 256, rows * cols // 4, dtype=np.uint8)
        h = hip.weights_upload_qk256(qs, rows, cols, cols // 4)
        wd = dense_qk256(qs, rows, cols)
    else:
        codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(rows, cols), p=[0.5, 0.25, 0.25])
        packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8).reshape(-1)
        scales = rng.uniform(0.01, 2.0, rows * cols // 32).astype(np.float32)
        if fmt == "f16":
            scales = scales.astype(np.float16).astype(np.float32)
        h = hip.weights_upload_i2s(packed, scales, rows, cols, 32)
        wd = dense_ternary(packed, scales, rows, cols)
    assert hip.gemv_q_supported(h)
    x = (rng.normal(0, 1, cols) * rng.choice([0.01, 1.0, 30.0], cols)).astype(np.float32)
    res = rng.normal(0, 1, rows).astype(np.float32)
    q, _ = gpu_quantize(hip, torch_, x)
    xq = dequantize_qact(q.cpu().numpy(), cols)
    want = wd @ xq
    bound = 2e-6 * (np.abs(wd) @ np.abs(xq)) + 1e-30  # ~45 f32 roundings in the longest accumulation chain, worst case
    y = torch_.full((rows,), float("nan"), device="cuda")
    hip.gemv_q_dev(h, q, y=y)
    torch_.cuda.synchronize()
    assert np.all(np.abs(y.cpu().numpy() - want) <= bound), (rows, cols, fmt, np.max(np.abs(y.cpu().numpy() - want) / bound))
    # residual + QAct output of the result (what the next GEMV would read), bit-exact against the restated quantiser
    y2 = torch_.full((rows,), float("nan"), device="cuda")
    qo = torch_.zeros(hip.qact_bytes(rows), dtype=torch_.uint8, device="cuda")
    so = torch_.zeros(rows // 16 * 2, dtype=torch_.float64, device="cuda")
    gam = rng.uniform(0.5, 1.5, rows).astype(np.float32)
    hip.gemv_q_dev(h, q, y=y2, residual=dev(torch_, res), qact_out=qo, gamma_out=dev(torch_, gam), stats_out=so)
    torch_.cuda.synchronize()
    v = y2.cpu().numpy()
    assert np.array_equal(v, y.cpu().numpy() + res)
    assert np.array_equal(qo.cpu().numpy()[: (rows + 255) // 256 * QREC], quantize_qact(v, gam)[: (rows + 255) // 256 * QREC])
    s = so.cpu().numpy().reshape(-1, 2)
    v64 = v.astype(np.float64).reshape(-1, 16)
    assert np.all(np.abs(s[:, 0] - v64.sum(1)) <= 1e-6 * np.abs(v64).sum(1)) and np.allclose(s[:, 1], (v64 * v64).sum(1), rtol=4e-6)
    hip.weights_free(h)


def approx_eq_with_len(got, want, cols):
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = min(2e-4 * np.sqrt(cols / 256), 1e-3)
    rel = diff / np.maximum(np.maximum(np.abs(got), np.abs(want)), 1e-30)
    return (diff < tol) | (rel < 2e-2)


def test_qk256_chain_at_2b4t_shapes_vs_oracle(hip, oracle, torch_):
    """BASELINE configs[2] shapes: x -> [LN, q|k|v]; att -> [o + residual -> QAct(gamma_ffn) + stats] -> [LN, gate|up, silu*mul ->
    QAct] -> [down + residual], every stage against the oracle's f32 chain on the SAME f32 inputs."""
    rng = np.random.default_rng(11)
    H, F, QKV = 2560, 6912, 3840
    st = lambda cols: cols // 4
    mk = lambda r, c: rng.integers(0, 256, r * st(c), dtype=np.uint8)
    wq, wo, wg, wu, wdn = mk(QKV, H), mk(H, H), mk(F, H), mk(F, H), mk(H, F)
    g_attn = (rng.uniform(0.5, 1.5, H) / (1.58 * np.sqrt(H))).astype(np.float32)
    g_ffn = (rng.uniform(0.5, 1.5, H) / (1.58 * np.sqrt(H))).astype(np.float32)
    hq, ho = hip.weights_upload_qk256(wq, QKV, H, st(H)), hip.weights_upload_qk256(wo, H, H, st(H))
    hg, hu = hip.weights_upload_qk256(wg, F, H, st(H)), hip.weights_upload_qk256(wu, F, H, st(H))
    hgu = hip.weights_concat([hg, hu], interleave16=True)
    hd = hip.weights_upload_qk256(wdn, H, F, st(F))
    ga, gf = dev(torch_, g_attn), dev(torch_, g_ffn)
    hip.weights_bind_ln(hq, ga)
    hip.weights_bind_ln(hgu, gf)
    z = lambda n, dt=torch_.uint8: torch_.zeros(n, dtype=dt, device="cuda")
    # stage 1: LN + q|k|v from a QAct of x (as the embedding / previous down-projection leaves it)
    x = rng.normal(0.05, 1.0, H).astype(np.float32)
    qx, sx = z(hip.qact_bytes(H)), z(H // 16 * 2, torch_.float64)
    hip.quantize_act_dev(dev(torch_, x), ga, H, qx, sx)
    y = z(QKV, torch_.float32)
    hip.gemv_q_dev(hq, qx, y=y, stats_in=sx, ln_gamma=ga, ln_eps=1e-5)
    torch_.cuda.synchronize()
    want = oracle.gemv_qk256(wq, oracle.layernorm(x, g_attn, 1e-5), QKV, H, st(H))
    got = y.cpu().numpy()
    assert cosine(got, want) >= 0.99999 and np.max(np.abs(got - want)) <= 2e-4 * np.max(np.abs(want)), (cosine(got, want), np.max(np.abs(got - want)))
    # stage 2: o-projection of an attention output + residual, leaving QAct(gamma_ffn * x2) + stats
    att = rng.normal(0, 0.7, H).astype(np.float32)
    qa = z(hip.qact_bytes(H))
    hip.quantize_act_dev(dev(torch_, att), None, H, qa)
    x2d, qx2, sx2 = z(H, torch_.float32), z(hip.qact_bytes(H)), z(H // 16 * 2, torch_.float64)
    hip.gemv_q_dev(ho, qa, y=x2d, residual=dev(torch_, x), qact_out=qx2, gamma_out=gf, stats_out=sx2)
    torch_.cuda.synchronize()
    want_o = oracle.gemv_qk256(wo, att, H, H, st(H))
    x2 = x2d.cpu().numpy()
    # (the reference's element-wise approx_eq_with_len is for f32 activations; SURVEY.md 8d gates reduced-precision activations
    #  on the per-vector cosine: an output row that happens to be ~0 carries the vector's absolute error, not a relative one)
    assert cosine(x2 - x, want_o) >= 0.99999 and np.max(np.abs(x2 - x - want_o)) <= 2e-4 * np.max(np.abs(want_o))
    # stage 3: LN + gate|up + silu*mul on the QAct the o-projection left; QAct-only output
    qh = z(hip.qact_bytes(F))
    hd_f32 = z(F, torch_.float32)
    hip.gemv_q_dev(hgu, qx2, y=hd_f32, stats_in=sx2, ln_gamma=gf, ln_eps=1e-5, flags=1, qact_out=qh)
    torch_.cuda.synchronize()
    xn = oracle.layernorm(x2, g_ffn, 1e-5)  # the oracle continues from the device's f32 x2: stage errors do not add up in this check
    g, u = oracle.gemv_qk256(wg, xn, F, H, st(H)), oracle.gemv_qk256(wu, xn, F, H, st(H))
    want_h = (g / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32) * u
    hgot = hd_f32.cpu().numpy()
    assert cosine(hgot, want_h) >= 0.99999, cosine(hgot, want_h)
    # stage 4: down + residual from that QAct
    xo = z(H, torch_.float32)
    hip.gemv_q_dev(hd, qh, y=xo, residual=x2d)
    torch_.cuda.synchronize()
    want_d = oracle.gemv_qk256(wdn, hgot, H, F, st(F))
    got_d = xo.cpu().numpy() - x2
    assert cosine(got_d, want_d) >= 0.99999, cosine(got_d, want_d)
    assert np.max(np.abs(got_d - want_d)) <= 2e-4 * np.max(np.abs(want_d))
    for h in (hq, ho, hg, hu, hgu, hd):
        hip.weights_free(h)


def test_bitnet32_f16_chain_at_2b4t_shapes_vs_oracle(hip, pkg, oracle, torch_):
    """BASELINE configs[1] storage: the same four stages with ternary codes + f16 32-block scales -- the kernel instances
    bench.py times (k_gemv_q<8,5,2,3> for gate|up)."""
    import importlib

    synth = importlib.import_module("bitnet-rs_amd.synth")
    cfg = synth.ModelConfig(**synth.BITNET_2B_4T)
    lay = synth.make_layer(cfg, 1, fmt="i2s", block=32)
    H, F = cfg.hidden, cfg.ffn
    up = lambda n: hip.weights_upload_i2s(lay[n], lay[n + "_scales"], cfg.shapes()[n][0], cfg.shapes()[n][1], 32)
    om = lambda n, a: oracle.i2s_matmul(a, lay[n], lay[n + "_scales"], 1, cfg.shapes()[n][0], cfg.shapes()[n][1], 32)
    hs = {n: up(n) for n in ("q", "k", "v", "o", "gate", "up", "down")}
    hqkv = hip.weights_concat([hs["q"], hs["k"], hs["v"]])
    hgu = hip.weights_concat([hs["gate"], hs["up"]], interleave16=True)
    ga, gf = dev(torch_, lay["attn_norm"]), dev(torch_, lay["ffn_norm"])
    hip.weights_bind_ln(hqkv, ga)
    hip.weights_bind_ln(hgu, gf)
    rng = np.random.default_rng(5)
    z = lambda n, dt=torch_.uint8: torch_.zeros(n, dtype=dt, device="cuda")
    x = rng.normal(-0.02, 1.3, H).astype(np.float32)
    qx, sx = z(hip.qact_bytes(H)), z(H // 16 * 2, torch_.float64)
    hip.quantize_act_dev(dev(torch_, x), ga, H, qx, sx)
    y = z(3840, torch_.float32)
    hip.gemv_q_dev(hqkv, qx, y=y, stats_in=sx, ln_gamma=ga, ln_eps=cfg.eps)
    torch_.cuda.synchronize()
    xn = oracle.layernorm(x, lay["attn_norm"], cfg.eps)
    want = np.concatenate([om(n, xn) for n in ("q", "k", "v")])
    got = y.cpu().numpy()
    assert cosine(got, want) >= 0.99999 and np.max(np.abs(got - want)) <= 2e-4 * np.max(np.abs(want))
    att = rng.normal(0, 0.5, H).astype(np.float32)
    qa = z(hip.qact_bytes(H))
    hip.quantize_act_dev(dev(torch_, att), None, H, qa)
    x2d, qx2, sx2 = z(H, torch_.float32), z(hip.qact_bytes(H)), z(H // 16 * 2, torch_.float64)
    hip.gemv_q_dev(hs["o"], qa, y=x2d, residual=dev(torch_, x), qact_out=qx2, gamma_out=gf, stats_out=sx2)
    torch_.cuda.synchronize()
    x2 = x2d.cpu().numpy()
    assert cosine(x2 - x, om("o", att)) >= 0.99999 and np.max(np.abs(x2 - x - om("o", att))) <= 2e-4 * np.max(np.abs(om("o", att)))
    qh, hf = z(hip.qact_bytes(F)), z(F, torch_.float32)
    hip.gemv_q_dev(hgu, qx2, y=hf, stats_in=sx2, ln_gamma=gf, ln_eps=cfg.eps, flags=1, qact_out=qh)
    torch_.cuda.synchronize()
    xn2 = oracle.layernorm(x2, lay["ffn_norm"], cfg.eps)
    g, u = om("gate", xn2), om("up", xn2)
    want_h = (g / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32) * u
    hgot = hf.cpu().numpy()
    assert cosine(hgot, want_h) >= 0.99999 and np.max(np.abs(hgot - want_h)) <= 3e-4 * np.max(np.abs(want_h)), cosine(hgot, want_h)
    xo = z(H, torch_.float32)
    hip.gemv_q_dev(hs["down"], qh, y=xo, residual=x2d)
    torch_.cuda.synchronize()
    got_d, want_d = xo.cpu().numpy() - x2, om("down", hgot)
    assert cosine(got_d, want_d) >= 0.99999 and np.max(np.abs(got_d - want_d)) <= 2e-4 * np.max(np.abs(want_d))
    for h in list(hs.values()) + [hqkv, hgu]:
        hip.weights_free(h)


def test_gemv_q_argument_errors(hip, pkg, torch_):
    rng = np.random.default_rng(0)
    h = hip.weights_upload_qk256(rng.integers(0, 256, 16 * 128, dtype=np.uint8), 16, 300, 128)  # ragged K: not on the QAct path
    assert not hip.gemv_q_supported(h)
    q = torch_.zeros(hip.qact_bytes(300), dtype=torch_.uint8, device="cuda")
    y = torch_.zeros(16, device="cuda")
    with pytest.raises(pkg.BitNetHipError, match="not on the QAct path"):
        hip.gemv_q_dev(h, q, y=y)
    hip.weights_free(h)
    h = hip.weights_upload_qk256(rng.integers(0, 256, 32 * 64, dtype=np.uint8), 32, 256, 64)
    g = torch_.ones(256, device="cuda")
    with pytest.raises(pkg.BitNetHipError, match="bound with bitnet_hip_weights_bind_ln"):
        hip.gemv_q_dev(h, torch_.zeros(hip.qact_bytes(256), dtype=torch_.uint8, device="cuda"), y=torch_.zeros(32, device="cuda"), ln_gamma=g, ln_eps=1e-5,
                       stats_in=torch_.zeros(32, dtype=torch_.float64, device="cuda"))
    with pytest.raises(pkg.BitNetHipError, match="multiple of 16"):
        hip.quantize_act_dev(g, None, 100, torch_.zeros(1024, dtype=torch_.uint8, device="cuda"))
    hip.weights_free(h)


@pytest.mark.parametrize("n_heads,n_kv,fmt", [(8, 2, "qk256"), (4, 2, "qk256"), (20, 5, "f16"), (3, 3, "qk256")])
def test_merging_oproj_on_qact_path_equals_combine_then_project(hip, pkg, oracle, torch_, n_heads, n_kv, fmt):
    """Short contexts: bitnet_hip_attention_decode_partial_dev + bitnet_hip_gemv_attn_merge_q_dev (k_gemv_q merging the chunk
    records into its LDS image itself) against bitnet_hip_attention_decode_q_dev (combine kernel -> QAct) +
    bitnet_hip_gemv_q_dev, contexts of 1..4 chunks: same y, same QAct handed to the next GEMV."""
    D, max_pos = 128, 512
    cols, rows = n_heads * D, 640
    rng = np.random.default_rng(11 * n_heads + n_kv)
    if fmt == "qk256":
        w = hip.weights_upload_qk256(rng.integers(0, 256, rows * cols // 4, dtype=np.uint8), rows, cols, cols // 4) if cols % 256 == 0 else None
    else:
        codes = rng.integers(0, 256, rows * cols // 4, dtype=np.uint8)
        sc = rng.uniform(0.05, 1.0, rows * cols // 32).astype(np.float16).astype(np.float32)
        w = hip.weights_upload_i2s(codes, sc, rows, cols, 32)
    if w is None or not hip.gemv_q_supported(w):
        pytest.skip("shape not on the QAct path (K % 256 != 0)")
    sin, cos = oracle.rope_tables(D, max_pos, 10000.0)
    sin_d, cos_d = dev(torch_, sin), dev(torch_, cos)
    sb = hip.c.bitnet_hip_attention_scratch_bytes(n_kv, max_pos)
    gam = dev(torch_, rng.uniform(0.5, 1.5, rows).astype(np.float32))
    z = lambda n, dt=torch_.uint8: torch_.zeros(n, dtype=dt, device="cuda")
    for pos in (0, 1, 63, 64, 130, 255):
        kc = rng.normal(0, 1, n_kv * max_pos * D).astype(np.float32)
        vc = rng.normal(0, 1, n_kv * max_pos * D).astype(np.float32)
        qkv = dev(torch_, rng.normal(0, 1.5, (n_heads + 2 * n_kv) * D).astype(np.float32))
        res = dev(torch_, rng.normal(0, 1, rows).astype(np.float32))
        pos_d = torch_.tensor([pos], dtype=torch_.int32, device="cuda")
        k1, v1, s1 = dev(torch_, kc), dev(torch_, vc), torch_.zeros(sb // 4 + 16, device="cuda")
        qa, y1, q1, st1 = z(hip.qact_bytes(cols)), z(rows, torch_.float32), z(hip.qact_bytes(rows)), z(rows // 16 * 2, torch_.float64)
        hip.attention_decode_q_dev(qkv, sin_d, cos_d, k1, v1, n_heads, n_kv, D, max_pos, pos_d, s1, None, qa)
        hip.gemv_q_dev(w, qa, y=y1, residual=res, qact_out=q1, gamma_out=gam, stats_out=st1)
        k2, v2, s2 = dev(torch_, kc), dev(torch_, vc), torch_.zeros(sb // 4 + 16, device="cuda") + 3.0  # stale-but-finite records past the context
        y2, q2, st2 = z(rows, torch_.float32), z(hip.qact_bytes(rows)), z(rows // 16 * 2, torch_.float64)
        hip.attention_decode_partial_dev(qkv, sin_d, cos_d, k2, v2, n_heads, n_kv, D, max_pos, pos_d, s2)
        hip.gemv_attn_merge_q_dev(w, s2, n_heads, n_kv, max_pos, pos_d, y2, q2, residual=res, gamma_out=gam, stats_out=st2)
        torch_.cuda.synchronize()
        a, b = y1.cpu().numpy(), y2.cpu().numpy()
        assert np.max(np.abs(a - b)) <= 3e-5 * max(1.0, np.max(np.abs(a))), (n_heads, n_kv, pos, np.max(np.abs(a - b)))
        da, db = dequantize_qact(q1.cpu().numpy(), rows), dequantize_qact(q2.cpu().numpy(), rows)
        assert np.max(np.abs(da - db)) <= 2e-4 * max(1.0, np.max(np.abs(da)))
        assert np.array_equal(q2.cpu().numpy(), quantize_qact(b, gam.cpu().numpy()))  # its own output, quantised exactly as specified
        assert torch_.equal(k1, k2) and torch_.equal(v1, v2)
    hip.weights_free(w)
