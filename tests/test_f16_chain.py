"""GPU parity of the prompt forward's f16 activation chain (bitnet_hip_matmul_f16_dev, kernels_gemm.hip k_gemm_f16a<.., EPI = 1>; the
operators it replaces: T:683-691 per-row loop / K/cpu/quantized_matmul.rs:57-96, LayerNorm T:67-100, silu * up T:756-781, residual
T:1073) against the CPU oracle, through the C ABI.

Error model (written in the gates): an activation element is held as f16 (11-bit significand, rounded to nearest: relative error
uniform in +-2^-12, rms 2^-12 / sqrt(3)); weights (code map value x f16 block scale) are exact in f16 and products are exact in the
f32 accumulator, so an output errs by sigma = 2^-12 / sqrt(3) * sqrt(sum_k (w_k x_k)^2) plus f32 accumulation noise.  Gate: 7 sigma per
output (+ 1e-6 * sum |w_k x_k| for the accumulation order) and cosine >= 0.99999 per row (benches/qk256_gemv.rs:234)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LUT_QK = np.array([-2, -1, 1, 2], np.float64)
LUT_T = np.array([0, 1, 0, -1], np.float64)


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.fixture(scope="module")
def torch_():
    import torch

    return torch


RAW = {}  # handle -> the packed matrix as the ORACLE takes it (oracle_rows below)


def oracle_rows(oracle, h, xrows):
    """The same matrix through the oracle .so -- gemv_qk256 (Q/i2s_qk256.rs:346) / i2s_matmul_f32 (K/cpu/quantized_matmul.rs:57-96) -- on a few
    activation rows: the dense f64 products this file gates against are built in the test, and would not notice a quirk shared by the test's
    decoding and the kernel's (VERDICT r04 'what's weak' 2)."""
    kind, a, b, rows, cols = RAW[h]
    if kind == "qk256":
        return np.stack([oracle.gemv_qk256(a, r, rows, cols, cols // 256 * 64) for r in xrows]).astype(np.float64)
    return np.stack([oracle.i2s_matmul(r, a, b, 1, rows, cols, 32) for r in xrows]).astype(np.float64)


def make_matrix(hip, rng, fmt, rows, cols):
    """-> (handle, dense f64 matrix [rows, cols])"""
    if fmt == "qk256":
        stride = cols // 256 * 64
        qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
        pk = qs.reshape(rows, cols // 4)
        codes = np.stack([(pk >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
        h = hip.weights_upload_qk256(qs, rows, cols, stride)
        RAW[h] = ("qk256", qs, None, rows, cols)
        return h, LUT_QK[codes]
    codes = rng.choice(np.array([0, 1, 3], np.uint8), size=(rows, cols), p=[0.5, 0.25, 0.25])
    packed = (codes[:, 0::4] | codes[:, 1::4] << 2 | codes[:, 2::4] << 4 | codes[:, 3::4] << 6).astype(np.uint8)
    scales = (1.0 / ((np.arange(rows * (cols // 32)) % 100) + 1)).astype(np.float16).astype(np.float32)
    dense = LUT_T[codes] * np.repeat(scales.reshape(rows, cols // 32).astype(np.float64), 32, axis=1)
    h = hip.weights_upload_i2s(packed.reshape(-1), scales, rows, cols, 32)
    RAW[h] = ("i2s", packed.reshape(-1), scales, rows, cols)
    return h, dense


def gate(got, want, w, xa):
    """7 sigma of the f16 rounding of the activations + f32 accumulation slack; cosine per row"""
    sigma = 2.0 ** -12 / np.sqrt(3.0) * np.sqrt((xa.astype(np.float64) ** 2) @ (w ** 2).T)
    slack = 1e-6 * (np.abs(xa).astype(np.float64) @ np.abs(w).T) + 1e-7
    bad = np.abs(got - want) > 7.0 * sigma + slack
    assert not bad.any(), float(np.max(np.abs(got - want) / (7.0 * sigma + slack)))
    for i in range(got.shape[0]):
        assert cosine(got[i], want[i]) >= 0.99999, i


@pytest.mark.parametrize("fmt", ["qk256", "i2s"])
@pytest.mark.parametrize("rows,cols,m", [(256, 256, 1), (512, 2560, 100), (2560, 768, 64), (768, 6912, 300), (2560, 768, 4000)])
def test_f16_chain_plain_residual_and_handover(hip, oracle, torch_, fmt, rows, cols, m):
    """y = residual + W xh (in place), the f16 copy f16(gamma_out * y) and the LayerNorm partials the next projection reads"""
    rng = np.random.default_rng(rows + cols + m)
    h, w = make_matrix(hip, rng, fmt, rows, cols)
    assert hip.matmul_f16_supported(h)
    mp = -(-m // 64) * 64
    x = (rng.normal(0, 1, (m, cols)) * np.exp(rng.uniform(-2, 2, (m, 1)))).astype(np.float32)
    res = rng.normal(0, 1, (m, rows)).astype(np.float32)
    gout = rng.uniform(0.5, 1.5, rows).astype(np.float32)
    xh = torch_.zeros(mp, cols, dtype=torch_.float16, device="cuda")
    hip.rows_to_f16_dev(torch_.from_numpy(x).cuda(), None, m, cols, xh, None)
    torch_.cuda.synchronize()
    x16 = x.astype(np.float16)
    assert np.array_equal(xh[:m].cpu().numpy(), x16)  # round to nearest even, as numpy
    y = torch_.from_numpy(res).cuda()
    yh = torch_.full((mp, rows), float("nan"), dtype=torch_.float16, device="cuda")
    st = torch_.full((rows // 64, mp, 2), float("nan"), device="cuda")
    hip.matmul_f16_dev(h, xh, m, y=y, residual=y, yh=yh, gamma_out=torch_.from_numpy(gout).cuda(), stats_out=st)
    torch_.cuda.synchronize()
    got = y.cpu().numpy()
    want = res.astype(np.float64) + x16.astype(np.float64) @ w.T
    assert np.isfinite(got).all()
    # the activations ARE f16 values here (xh is the input): only accumulation order separates the kernel from the f64 product
    assert np.max(np.abs(got - want) / (np.abs(x16).astype(np.float64) @ np.abs(w).T + np.abs(res) + 1e-6)) <= 2e-6
    # ... and the oracle's own loop on the same f16-valued rows (f32, left to right: its accumulation error is the larger of the two)
    pick = np.unique(np.r_[0, m - 1, rng.integers(0, m, 3)])
    oref = oracle_rows(oracle, h, x16[pick].astype(np.float32))
    mag = np.abs(x16[pick]).astype(np.float64) @ np.abs(w).T + 1e-6
    assert np.max(np.abs(got[pick] - res[pick] - oref) / mag) <= 3e-5, float(np.max(np.abs(got[pick] - res[pick] - oref) / mag))
    # hand-over: the f16 copy is the f32 output times gamma_out, rounded once; partials are the 64-row slab sums of the f32 output
    assert np.array_equal(yh[:m].cpu().numpy(), (got * gout).astype(np.float16))
    stn = st.cpu().numpy()[:, :m, :]
    wr = hip.matmul_last_wave_rows()  # 64, or 80 in the 320-row workgroups a long 2560-row launch takes (4000 tokens: 504 workgroups, one round)
    assert wr == (80 if (rows, m) == (2560, 4000) else 64)
    if wr == 80:
        assert np.all(stn[rows // 80 :] == 0)  # the surplus entries of the [rows / 64] array are written as zero
        stn = stn[: rows // 80]
    slabs = got.reshape(m, rows // wr, wr).astype(np.float64)
    # (f32 sums of 64 terms in the kernel: 2^-24 relative per addition, against the sum of the magnitudes)
    assert np.all(np.abs(stn[:, :, 0].T - slabs.sum(axis=2)) <= 4e-6 * np.abs(slabs).sum(axis=2) + 1e-6)
    assert np.all(np.abs(stn[:, :, 1].T - (slabs ** 2).sum(axis=2)) <= 4e-6 * (slabs ** 2).sum(axis=2) + 1e-6)
    if mp > m:
        assert np.all(st.cpu().numpy()[:, m:, :] == 0)  # padding tokens: zero partials
    hip.weights_free(h)


@pytest.mark.parametrize("fmt", ["qk256", "i2s"])
def test_f16_chain_layernorm_after_product_and_silu(hip, oracle, torch_, fmt):
    """qkv-like (LayerNorm -> f32 rows) and gate|up-like (LayerNorm -> silu(gate) * up -> f16 rows) launches at the 2B-4T widths,
    LayerNorm statistics from one partial (the chain's entry) and from 40 (a producer's slabs), against the oracle's LayerNorm + product"""
    rng = np.random.default_rng(11)
    K, N, m = 2560, 512, 96
    mp = -(-m // 64) * 64
    ha, wa = make_matrix(hip, rng, fmt, N, K)
    hb, wb = make_matrix(hip, rng, fmt, N, K)
    x = rng.normal(0.15, 1.0, (m, K)).astype(np.float32)
    g = (rng.uniform(0.5, 1.5, K) / (80 if fmt == "qk256" else 4)).astype(np.float32)
    gd = torch_.from_numpy(g).cuda()
    xn = np.stack([oracle.layernorm(x[i], g, 1e-5) for i in range(m)]).astype(np.float64)
    ya, yb = xn @ wa.T, xn @ wb.T
    xd = torch_.from_numpy(x).cuda()
    xh = torch_.zeros(mp, K, dtype=torch_.float16, device="cuda")
    st1 = torch_.zeros(1, mp, 2, device="cuda")
    hip.rows_to_f16_dev(xd, gd, m, K, xh, st1)
    # the same statistics as 40 slab partials (what an o- / down-projection leaves)
    st40 = torch_.zeros(40, mp, 2, device="cuda")
    sl = x.reshape(m, 40, 64).astype(np.float64)
    st40[:, :m, 0] = torch_.from_numpy(sl.sum(axis=2).T.astype(np.float32)).cuda()
    st40[:, :m, 1] = torch_.from_numpy((sl ** 2).sum(axis=2).T.astype(np.float32)).cuda()
    hc = hip.weights_concat([ha, hb])
    hip.weights_bind_ln(hc, gd)
    xg = (x * g).astype(np.float64)  # what the f16 rounding acts on: sigma is taken over gamma * x, the mean term is exact
    wcat = np.concatenate([wa, wb], axis=0)
    for st, n_st in ((st1, 1), (st40, 40)):
        y = torch_.full((m, 2 * N), float("nan"), device="cuda")
        hip.matmul_f16_dev(hc, xh, m, stats_in=st, n_stats=n_st, ln_gamma=gd, ln_eps=1e-5, y=y)
        torch_.cuda.synchronize()
        got = y.cpu().numpy()
        denom = np.sqrt(x.astype(np.float64).var(axis=1, keepdims=True) + 1e-5)
        sigma = 2.0 ** -12 / np.sqrt(3.0) * np.sqrt((xg ** 2) @ (wcat ** 2).T) / denom
        want = np.concatenate([ya, yb], axis=1)
        slack = 2e-6 * (np.abs(xg) @ np.abs(wcat).T) / denom + 1e-6
        assert np.all(np.abs(got - want) <= 7.0 * sigma + slack), (n_st, float(np.max(np.abs(got - want) / (7.0 * sigma + slack))))
        for i in range(m):
            assert cosine(got[i], want[i]) >= 0.99999
    hg = hip.weights_concat([ha, hb], interleave16=True)
    hip.weights_bind_ln(hg, gd)
    hh = torch_.full((mp, N), float("nan"), dtype=torch_.float16, device="cuda")
    yf = torch_.full((m, N), float("nan"), device="cuda")
    hip.matmul_f16_dev(hg, xh, m, stats_in=st40, n_stats=40, ln_gamma=gd, ln_eps=1e-5, y=yf, flags=1, yh=hh)
    torch_.cuda.synchronize()
    want = ya / (1 + np.exp(-ya)) * yb
    got = yf.cpu().numpy()
    assert np.max(np.abs(got - want)) <= 2e-3 * np.max(np.abs(want))
    for i in range(m):
        assert cosine(got[i], want[i]) >= 0.99999
    assert np.array_equal(hh[:m].cpu().numpy(), got.astype(np.float16))
    with pytest.raises(Exception, match="bound gamma"):
        hip.matmul_f16_dev(hc, xh, m, stats_in=st1, n_stats=1, ln_gamma=torch_.from_numpy(g.copy()).cuda(), ln_eps=1e-5, y=yf)
    for h in (ha, hb, hc, hg):
        hip.weights_free(h)


def test_f16_chain_refuses_what_it_cannot_take(hip, torch_):
    rng = np.random.default_rng(3)
    n, k = 100, 512  # rows % 256 != 0
    qs = rng.integers(0, 256, n * (k // 256) * 64, dtype=np.uint8)
    h = hip.weights_upload_qk256(qs, n, k, k // 256 * 64)
    assert not hip.matmul_f16_supported(h)
    xh = torch_.zeros(64, k, dtype=torch_.float16, device="cuda")
    y = torch_.zeros(10, n, device="cuda")
    with pytest.raises(Exception, match="does not take the f16 chain"):
        hip.matmul_f16_dev(h, xh, 10, y=y)
    hip.weights_free(h)


def test_f16_hand_over_saturation_is_counted_and_the_decoder_falls_back(pkg, hip, oracle, torch_):
    """the f16 chain stores f16(gamma * x) with no row scale and clamped silently at +-65504.  Now every writer COUNTS a clamp
    (bitnet_hip_f16_saturations) and Decoder::prefill repeats a prompt that clamped anything on the 4-digit planes (30-bit fixed point behind a
    per-row power-of-two scale) before it takes the logits.  Op level: an in-range call leaves the counter at zero, an outlier channel raises it.
    Decoder level: a BitNet32-F16 model whose LAST feed-forward LayerNorm weight has one channel at 2e5 (gamma * x far beyond the f16 range on
    every token, and silu * up beyond it by orders of magnitude; the attention inputs stay in range -- the prompt attention multiplies f16
    operands whatever the projections do): the chain's result would be wrong; the decoder reports one fallback and matches the oracle."""
    import importlib

    synth = importlib.import_module("bitnet-rs_amd.synth")
    rng = np.random.default_rng(5)
    m, cols = 200, 512
    x = rng.normal(0.0, 1.0, (m, cols)).astype(np.float32)
    g = rng.uniform(0.5, 1.5, cols).astype(np.float32)
    xh = torch_.zeros(256, cols, dtype=torch_.float16, device="cuda")
    st = torch_.zeros(2 * 256, device="cuda")
    hip.f16_saturations(True)
    hip.rows_to_f16_dev(torch_.from_numpy(x).cuda(), torch_.from_numpy(g).cuda(), m, cols, xh, st)
    assert hip.f16_saturations(True) == 0
    g2 = g.copy()
    g2[17] = 3e5
    hip.rows_to_f16_dev(torch_.from_numpy(x).cuda(), torch_.from_numpy(g2).cuda(), m, cols, xh, st)
    n = hip.f16_saturations(True)
    assert 0 < n <= m and hip.f16_saturations(False) == 0  # one lane per row whose element 17 left the range; reading resets
    assert np.all(np.abs(xh.cpu().numpy()[:m, 17][np.abs(x[:, 17] * 3e5) > 65504]) == 65504.0)
    # ---- decoder level
    cfg = synth.ModelConfig(hidden=512, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, ffn=1024, vocab=2048, max_pos=320, eps=1e-5, rope_theta=10000.0)
    glob = synth.make_globals(cfg)
    layers = [synth.make_layer(cfg, l, fmt="i2s", block=32) for l in range(cfg.n_layers)]
    layers[-1]["ffn_norm"] = layers[-1]["ffn_norm"].copy()
    layers[-1]["ffn_norm"][33] = 2e5
    om = oracle.OracleModel(cfg, [dict(w, ternary=32) for w in layers], glob, n_threads=8)
    T = 256
    prompt = synth.prompt(T, cfg.vocab)
    want = None
    for t in prompt:
        _, want, _ = om.step(int(t), want_logits=True)
    om.close()
    dec = pkg.HostDecoder(cfg)
    for l, w in enumerate(layers):
        dec.set_layer_i2s(l, w, 32)
    dec.set_globals(glob)
    dec.reset()
    dec.feed(prompt)
    dec.prefill(T, with_logits=True, digits=2)
    assert dec.saturation_fallbacks() == 1 and dec.last_prefill_path() == 0
    got = dec.last_logits().astype(np.float64)
    c = float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want)))
    assert c >= 0.9999, c
    assert int(dec.history(T + 1)[T]) == oracle.argmax(want)
    dec.close()


_F16H_HASH_SCRIPT = r"""
import hashlib, importlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
hip = pkg.load(); hip.init(0)
import test_f16_chain as T
print("HASH", T.f16h_case(hip, torch, {fmt!r}, {kind!r})[0])
"""


def f16h_case(hip, torch_, fmt, kind):
    """The wide chain launches of a 4096-token prompt at the 2B-4T widths: kind 'gateup' (13824 x 2560, LayerNorm after the product, silu * up -> f16
    rows) or 'qkv' (3840 x 2560, LayerNorm -> f32 rows).  -> (sha256 of the output bytes, outputs, the pieces a checker needs)."""
    import hashlib

    rng = np.random.default_rng(123)
    K, m = 2560, 4096
    N = 6912 if kind == "gateup" else 1920
    ha, wa = make_matrix(hip, rng, fmt, N, K)
    hb, wb = make_matrix(hip, rng, fmt, N, K)
    x = rng.normal(0.15, 1.0, (m, K)).astype(np.float32)
    g = (rng.uniform(0.5, 1.5, K) / (80 if fmt == "qk256" else 4)).astype(np.float32)
    gd = torch_.from_numpy(g).cuda()
    xh = torch_.zeros(m, K, dtype=torch_.float16, device="cuda")
    st = torch_.zeros(1, m, 2, device="cuda")
    hip.rows_to_f16_dev(torch_.from_numpy(x).cuda(), gd, m, K, xh, st)
    h = hip.weights_concat([ha, hb], interleave16=(kind == "gateup"))
    hip.weights_bind_ln(h, gd)
    if kind == "gateup":
        out = torch_.full((m, N), float("nan"), dtype=torch_.float16, device="cuda")
        hip.matmul_f16_dev(h, xh, m, stats_in=st, n_stats=1, ln_gamma=gd, ln_eps=1e-5, flags=1, yh=out)
    else:
        out = torch_.full((m, 2 * N), float("nan"), device="cuda")
        hip.matmul_f16_dev(h, xh, m, stats_in=st, n_stats=1, ln_gamma=gd, ln_eps=1e-5, y=out)
    torch_.cuda.synchronize()
    tile = dict(hip.matmul_last_tile())
    o = out.cpu().numpy()
    for hh in (ha, hb, h):
        hip.weights_free(hh)
    return hashlib.sha256(o.tobytes()).hexdigest(), o, (x, g, wa, wb, tile)


@pytest.mark.parametrize("fmt", ["i2s", "qk256"])
@pytest.mark.parametrize("kind", ["gateup", "qkv"])
def test_f16h_wide_tile_instances_match_oracle_and_the_64_token_kernel(hip, oracle, torch_, fmt, kind):
    """k_gemm_f16h (round 5): the 64-row x 128-token wave tile the f16 chain's wide launches take at 4096 tokens -- gate|up split 48 + 6 row blocks
    (the six on k_gemm_f16a), q|k|v whole (480 workgroups: one round).  Tile asserted; sampled rows against the oracle's LayerNorm + product
    (+ silu * up); and the WHOLE output bit-identical to the launch with BITNET_HIP_GEMM_F16H=0 (k_gemm_f16a alone: same k-order, same f32
    accumulation per output element) -- the switch is read once per process, hence the child process."""
    import os
    import subprocess
    import sys

    h1, out, (x, g, wa, wb, tile) = f16h_case(hip, torch_, fmt, kind)
    assert tile["wave_tokens"] == 128 and tile["digits"] == 2 and tile["scale_mode"] == (4 if fmt == "i2s" else 5), tile
    assert np.isfinite(out.astype(np.float64)).all()
    rng = np.random.default_rng(9)
    for i in np.unique(np.r_[0, 4095, rng.integers(0, 4096, 6)]):
        xn = oracle.layernorm(x[i], g, 1e-5).astype(np.float64)
        a, b = xn @ wa.T, xn @ wb.T
        want = a / (1 + np.exp(-a)) * b if kind == "gateup" else np.concatenate([a, b])
        got = out[i].astype(np.float64)
        assert cosine(got, want) >= 0.99999, int(i)
        assert np.max(np.abs(got - want)) <= 2.5e-3 * np.max(np.abs(want)), int(i)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _F16H_HASH_SCRIPT.format(root=root, fmt=fmt, kind=kind)], env=dict(os.environ, BITNET_HIP_GEMM_F16H="0"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    h0 = [l.split()[1] for l in p.stdout.splitlines() if l.startswith("HASH ")][0]
    assert h0 == h1, "k_gemm_f16h's outputs differ from k_gemm_f16a's"
