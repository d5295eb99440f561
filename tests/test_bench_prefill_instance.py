"""The prefill instances bench.py TIMES (c4 / c5: `--digits 2`, 4096+ tokens at the bitnet-b1.58-2B-4T widths) against the
CPU oracle -- VERDICT r02 "what's weak" 1: the small 2-digit tests collapse the launcher's token tile to 16 / 32 tokens, a
different template instantiation from the 64-token tile `k_gemm_mfma<2, 4, 0, 2, 1>` the 25 ms number runs on.

  (a) every projection shape of the model x 4096 token rows x 2 digits, QK256 (64-token tile, K = 64 int8 MFMA on two digit planes)
      and BitNet32-F16 (64-token tile, f16 MFMA: weights +-s as f16, f16 activations -- k_gemm_f16w), with the fusions the prefill loop uses (LayerNorm prologue, silu*mul,
      residual); the launcher's choice is read back through bitnet_hip_matmul_last_tile and asserted; the oracle
      (gemv_qk256 Q/i2s_qk256.rs:196-321 per row as forward_qk256 does T:683-691; i2s_matmul_f32
      K/cpu/quantized_matmul.rs:57-96) runs on a sample of token rows, all output rows of them.
      Gate: cosine >= 0.99999 per row (benches/qk256_gemv.rs:234) and, per element, the rounding bound of the format: two digits
      hold an activation to a step of 2^-13 of its row's maximum (uniform error, sigma = step / sqrt(12)); a dot product over K
      weights of rms w_rms then errs by sigma * sqrt(K) * w_rms -- the gate is 7 sigma = 2^-12 * max|x| * sqrt(K) * w_rms; the
      BitNet32-F16 form holds activations as f16 (11 bits of the ELEMENT): sigma = 2^-11.5 / sqrt(3) * |x| * w_rms, gate 7 sigma.
  (b) whole prompt prefill(digits = 2) -> decode against the oracle's token-by-token model on the 2-layer model of the 2B-4T
      widths, 1024 tokens (the gate|up launch takes the wide tile there), both formats; plus the 4096-token prompt, where
      EVERY launch takes the wide tile, against the 4-digit prefill that (a) and test_prefill_parity pin to the oracle."""
import importlib
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M = 4096
SAMPLE = 64
WIDE = dict(hidden=2560, n_layers=2, n_heads=20, n_kv_heads=5, head_dim=128, ffn=6912, vocab=4096, max_pos=4224, eps=1e-5, rope_theta=500000.0)


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.fixture(scope="module")
def synth(pkg):
    return importlib.import_module("bitnet-rs_amd.synth")


@pytest.fixture(scope="module")
def torch_():
    import torch

    return torch


@pytest.fixture(scope="module")
def layers(synth):
    cfg = synth.ModelConfig(**synth.BITNET_2B_4T)
    return cfg, {"qk256": synth.make_layer(cfg, 0, fmt="qk256"), "i2s": synth.make_layer(cfg, 0, fmt="i2s", block=32)}


def upload(hip, lay, fmt, name, rows, cols):
    if fmt == "qk256":
        return hip.weights_upload_qk256(lay[name], rows, cols, cols // 256 * 64)
    return hip.weights_upload_i2s(lay[name], lay[name + "_scales"], rows, cols, 32)


def oracle_rows(oracle, lay, fmt, name, rows, cols, xs):
    """W x for every sampled row: the reference's own per-row loop (QK256) / i2s_matmul_f32 (ternary x block scale)."""
    if fmt == "qk256":
        return np.stack([oracle.gemv_qk256(lay[name], xs[i], rows, cols, cols // 256 * 64) for i in range(xs.shape[0])])
    return oracle.i2s_matmul(xs.reshape(-1), lay[name], lay[name + "_scales"], xs.shape[0], rows, cols, 32).reshape(xs.shape[0], rows)


EXPECT_TILE = {"qk256": dict(digits=2, wave_tokens=64, waves=4, scale_mode=0), "i2s": dict(digits=2, wave_tokens=64, waves=4, scale_mode=4)}
# the 2560-row launches (o, down) at 4096 tokens: 640 64-token tiles would leave the second round a quarter full, so the launcher
# takes 32-token tiles, three workgroups to a CU (kernels_gemm.hip gemm_token_tiles)
NARROW_CASES = ("o_residual", "down_residual")


def expect_tile(fmt, case):
    t = dict(EXPECT_TILE[fmt])
    if case in NARROW_CASES and fmt == "qk256":  # (the f16 form keeps 64: its 32-token tile measured slower)
        t["wave_tokens"] = 32
    return t


@pytest.mark.parametrize("fmt", ["qk256", "i2s"])
@pytest.mark.parametrize("case", ["qkv_ln", "o_residual", "gateup_ln_silu", "down_residual"])
def test_benchmarked_tile_matches_oracle(hip, oracle, torch_, layers, fmt, case):
    cfg, both = layers
    lay = both[fmt]
    H, F, QD, KD = cfg.hidden, cfg.ffn, cfg.n_heads * cfg.head_dim, cfg.n_kv_heads * cfg.head_dim
    rng = np.random.default_rng(zlib.crc32(f"{fmt}/{case}".encode()))
    sample = np.sort(rng.choice(M, SAMPLE, replace=False))
    sample[0], sample[-1] = 0, M - 1  # first and last row of the launch
    if case == "qkv_ln":
        names, K, gamma = ("q", "k", "v"), H, lay["attn_norm"]
    elif case == "gateup_ln_silu":
        names, K, gamma = ("gate", "up"), H, lay["ffn_norm"]
    elif case == "o_residual":
        names, K, gamma = ("o",), QD, None
    else:
        names, K, gamma = ("down",), F, None
    shapes = cfg.shapes()
    parts = [upload(hip, lay, fmt, n, *shapes[n]) for n in names]
    if len(parts) == 1:
        h = parts[0]
    else:
        h = hip.weights_concat(parts, interleave16=(case == "gateup_ln_silu"))
        for p in parts:
            hip.weights_free(p)
    # activations: per-row magnitudes over three decades (the fixed-point scale is per row), a non-zero mean for LayerNorm
    x = rng.normal(0.2 if gamma is not None else 0.0, 1.0, (M, K)).astype(np.float32)
    x *= np.exp(rng.uniform(np.log(0.05), np.log(50.0), (M, 1))).astype(np.float32)
    out_cols = F if case == "gateup_ln_silu" else sum(shapes[n][0] for n in names)
    res = rng.normal(0, 1, (M, out_cols)).astype(np.float32) if gamma is None else None
    wsb = hip.matmul_workspace_bytes(M, K, 4)  # sized for the 4-digit pass below; the 2-digit one needs less
    ws = torch_.empty(wsb, dtype=torch_.uint8, device="cuda")
    xd = torch_.from_numpy(x).cuda()
    yd = torch_.full((M, out_cols), float("nan"), device="cuda")
    gd = torch_.from_numpy(gamma).cuda() if gamma is not None else None
    rd = torch_.from_numpy(res).cuda() if res is not None else None
    hip.matmul_fused_dev(h, xd, yd, M, ws, wsb, ln_gamma=gd, ln_eps=cfg.eps if gamma is not None else 0.0, residual=rd,
                         flags=1 if case == "gateup_ln_silu" else 0, digits=2)
    torch_.cuda.synchronize()
    assert hip.matmul_last_tile() == expect_tile(fmt, case), hip.matmul_last_tile()  # the instance bench.py's c4 / c5 prefill runs
    got_all = yd.cpu().numpy()
    assert not np.isnan(got_all).any()
    got = got_all[sample]
    xs = x[sample]
    if gamma is not None:
        xs = np.stack([oracle.layernorm(xs[i], gamma, cfg.eps) for i in range(SAMPLE)])
    ys = [oracle_rows(oracle, lay, fmt, n, *shapes[n], xs) for n in names]
    if case == "gateup_ln_silu":
        g, u = ys
        want = (g / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32) * u  # T:756-781
        prod = want
    else:
        prod = np.concatenate(ys, axis=1)
        want = prod + (res[sample] if res is not None else 0.0)
    # the format's 7-sigma rounding bound (module docstring).  QK256: two digits = a step of 2^-13 of the row maximum, weights
    # uniform over {-2,-1,1,2} (rms 1.58).  BitNet32-F16: f16 activations, every element rounded to 11 bits, and the synthetic
    # block scales differ widely from output row to output row, so sigma is taken per (token, output): 2^-11.5 / sqrt(3) * sqrt(sum_k (w_k x_k)^2)
    # (an f16 value m 2^e is rounded to +-2^(e-11): uniform, rms 2^(e-11) / sqrt(3); relative to the value, averaged over m in [1, 2): x 2^-0.5)
    if gamma is not None:  # what the quantiser sees: the normalised row (f64 here: this only sizes the bound)
        x64 = x.astype(np.float64)
        xn_all = ((x64 - x64.mean(axis=1, keepdims=True)) / np.sqrt(x64.var(axis=1, keepdims=True) + cfg.eps) * gamma).astype(np.float32)
    else:
        xn_all = x
    if fmt == "qk256":
        sig = lambda rows_idx: (2.0 ** -12 / 7.0 * np.max(np.abs(xn_all[rows_idx]), axis=1) * np.sqrt(K) * 1.58)[:, None] * np.ones((1, 1))
        sig_parts = lambda rows_idx: [sig(rows_idx) for _ in names]
    else:
        def dense_sq(name):
            r_, c_ = shapes[name]
            pk = lay[name].reshape(r_, c_ // 4)
            codes = np.stack([(pk >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(r_, c_)
            return ((codes == 1) | (codes == 3)) * np.repeat(lay[name + "_scales"].reshape(r_, c_ // 32) ** 2, 32, axis=1)
        wsq = [dense_sq(n).astype(np.float32) for n in names]
        sig_parts = lambda rows_idx: [2.0 ** -11.5 / np.sqrt(3.0) * np.sqrt((xn_all[rows_idx] ** 2) @ wq.T) for wq in wsq]
    sp = sig_parts(sample)
    if case == "gateup_ln_silu":  # d(silu(g) u) <= |u| * max|silu'| * dg + |silu(g)| * du
        tol = 7.0 * (1.1 * np.abs(ys[1]) * sp[0] + np.abs(ys[0]) * sp[1])
    else:
        tol = 7.0 * (np.concatenate([np.broadcast_to(a, (SAMPLE, shapes[n][0])) for a, n in zip(sp, names)], axis=1))
    err = np.abs(got - want)
    assert np.all(err <= tol + 1e-6), (fmt, case, float(np.max(err / (tol + 1e-30))))
    for i in range(SAMPLE):
        gp = got[i] - (res[sample[i]] if res is not None else 0.0)
        assert cosine(gp, prod[i]) >= 0.99999, (fmt, case, int(sample[i]))
    # 4 digits (the oracle-pinned form of tests/test_gemm_parity.py) over ALL 4096 rows: 2 digits may differ from it by the
    # same rounding bound (silu * mul: its operands are not separately visible, the sampled check above covers it)
    if case != "gateup_ln_silu":
        hip.matmul_fused_dev(h, xd, yd, M, ws, wsb, ln_gamma=gd, ln_eps=cfg.eps if gamma is not None else 0.0, residual=rd, flags=0, digits=4)
        torch_.cuda.synchronize()
        y4 = yd.cpu().numpy()
        if res is not None:
            got_all, y4 = got_all - res, y4 - res
        every = np.arange(M)
        tol_rows = 7.0 * np.concatenate([np.broadcast_to(a, (M, shapes[n][0])) for a, n in zip(sig_parts(every), names)], axis=1)
        assert np.all(np.abs(got_all - y4) <= tol_rows + 1e-6), float(np.max(np.abs(got_all - y4) / (tol_rows + 1e-30)))
    hip.weights_free(h)


@pytest.mark.parametrize("fmt", ["qk256", "i2s"])
@pytest.mark.parametrize("case", ["o", "down"])
def test_benchmarked_f16_tile_of_the_2560_row_launches_matches_oracle(hip, oracle, torch_, layers, fmt, case):
    """The o- / down-projection of the timed 4096-token prompt, BOTH formats (round 4): their inputs are f16 rows already (the attention
    output, silu * up), so they go to the f16 matrix cores as they stand -- bitnet_hip_matmul_f16_dev, k_gemm_f16a in 320-row
    workgroups (five row tiles per wave: 8 x 64 = 512 workgroups, one round of the chip), x = x + W h in place, plus the chain's
    hand-over outputs (f16(gamma * x), LayerNorm partials).  Against the oracle's rows on the SAME f16 values: nothing is rounded
    but the f32 accumulation, so the gate is 2e-6 of sum |w x| (the reference's own loops round about as much)."""
    cfg, both = layers
    lay = both[fmt]
    name = case
    rows, K = cfg.shapes()[name]
    assert rows == 2560
    rng = np.random.default_rng(zlib.crc32(f"f16/{fmt}/{case}".encode()))
    sample = np.sort(rng.choice(M, SAMPLE, replace=False))
    sample[0], sample[-1] = 0, M - 1
    h = upload(hip, lay, fmt, name, rows, K)
    assert hip.matmul_f16_supported(h)
    x16 = (rng.normal(0.0, 1.0, (M, K)) * np.exp(rng.uniform(np.log(0.05), np.log(50.0), (M, 1)))).astype(np.float16)
    res = rng.normal(0, 1, (M, rows)).astype(np.float32)
    gout = rng.uniform(0.5, 1.5, rows).astype(np.float32)
    xh = torch_.from_numpy(x16).cuda()
    y = torch_.from_numpy(res).cuda()
    yh = torch_.full((M, rows), float("nan"), dtype=torch_.float16, device="cuda")
    st = torch_.full((rows // 64, M, 2), float("nan"), device="cuda")
    hip.matmul_f16_dev(h, xh, M, y=y, residual=y, yh=yh, gamma_out=torch_.from_numpy(gout).cuda(), stats_out=st)
    torch_.cuda.synchronize()
    assert hip.matmul_last_tile() == dict(digits=2, wave_tokens=64, waves=4, scale_mode=5 if fmt == "qk256" else 4)
    assert hip.matmul_last_wave_rows() == 80
    got_all = y.cpu().numpy()
    assert np.isfinite(got_all).all()
    xs = x16[sample].astype(np.float32)
    prod = oracle_rows(oracle, lay, fmt, name, rows, K, xs)
    want = prod + res[sample]
    if fmt == "qk256":
        mag = np.abs(xs).astype(np.float64).sum(axis=1, keepdims=True) * 2.0  # |w| <= 2
    else:
        pk = lay[name].reshape(rows, K // 4)
        codes = np.stack([(pk >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, K)
        wabs = ((codes == 1) | (codes == 3)) * np.repeat(np.abs(lay[name + "_scales"]).reshape(rows, K // 32), 32, axis=1)
        mag = np.abs(xs).astype(np.float64) @ wabs.T.astype(np.float64)
    err = np.abs(got_all[sample] - want)
    assert np.all(err <= 2e-6 * mag + 1e-6 * np.abs(res[sample]) + 1e-6), float(np.max(err / (2e-6 * mag + 1e-6)))
    for i in range(SAMPLE):
        assert cosine(got_all[sample[i]] - res[sample[i]], prod[i]) >= 0.999999, int(sample[i])
    # hand-over: f16(gamma_out * y) rounded once; the rows / 64 partial entries add up to the row's (sum, sum of squares) -- with
    # 320-row workgroups the first rows / 80 entries are the waves' 80-row sums and the rest are zero
    assert np.array_equal(yh.cpu().numpy(), (got_all * gout).astype(np.float16))
    stn = st.cpu().numpy().astype(np.float64)
    g64 = got_all.astype(np.float64)
    assert np.all(np.abs(stn[:, :, 0].sum(axis=0) - g64.sum(axis=1)) <= 4e-6 * np.abs(g64).sum(axis=1) + 1e-6)
    assert np.all(np.abs(stn[:, :, 1].sum(axis=0) - (g64 ** 2).sum(axis=1)) <= 4e-6 * (g64 ** 2).sum(axis=1) + 1e-6)
    slabs = g64.reshape(M, rows // 80, 80)
    assert np.all(np.abs(stn[: rows // 80, :, 0].T - slabs.sum(axis=2)) <= 4e-6 * np.abs(slabs).sum(axis=2) + 1e-6)
    assert np.all(stn[rows // 80 :] == 0)
    hip.weights_free(h)


def _models(synth, fmt, cfg):
    glob = synth.make_globals(cfg)
    if fmt == "qk256":
        layers = [synth.make_layer(cfg, l) for l in range(cfg.n_layers)]
        return glob, layers, layers
    layers = [synth.make_layer(cfg, l, fmt="i2s", block=32) for l in range(cfg.n_layers)]
    tmap = np.array([0, 1, 0, -1], np.float32)
    olayers = []
    for lay in layers:
        d = {"attn_norm": lay["attn_norm"], "ffn_norm": lay["ffn_norm"], "dense": True}
        for name, (rows, cols) in cfg.shapes().items():
            pk = lay[name].reshape(rows, cols // 4)
            codes = np.stack([(pk >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
            d[name] = tmap[codes] * np.repeat(lay[name + "_scales"].reshape(rows, cols // 32), 32, axis=1)
        olayers.append(d)
    return glob, layers, olayers


@pytest.mark.parametrize("fmt", ["qk256", "i2s"])
@pytest.mark.parametrize("name", ["q", "down"])
def test_full_size_matmul_properties_are_bit_exact(hip, torch_, layers, fmt, name):
    """Size-independent properties of the benchmarked launches at m = 4096 (q: 64-token tiles; down: the narrow-tile rule), checked
    on EVERY output, bit for bit:
      * scaling the activations by a power of two scales the outputs by it (row scales are powers of two, digits / f16 mantissas
        do not change) -- any rounding that depended on the tile position or the launch shape would break it;
      * permuting the token rows permutes the output rows (a row's result does not depend on which tile, wave or round it is in);
      * the first 1000 rows alone (another grid, narrower tiles) give the same rows."""
    cfg, both = layers
    lay = both[fmt]
    rows, cols = cfg.shapes()[name]
    h = upload(hip, lay, fmt, name, rows, cols)
    rng = np.random.default_rng(zlib.crc32(f"prop/{fmt}/{name}".encode()))
    x = (rng.normal(0, 1, (M, cols)) * np.exp(rng.uniform(-3, 3, (M, 1)))).astype(np.float32)
    wsb = hip.matmul_workspace_bytes(M, cols, 2)
    ws = torch_.empty(wsb, dtype=torch_.uint8, device="cuda")

    def run(xa, m=M):
        xd = torch_.from_numpy(np.ascontiguousarray(xa)).cuda()
        yd = torch_.full((m, rows), float("nan"), device="cuda")
        hip.matmul_fused_dev(h, xd, yd, m, ws, wsb, digits=2)
        torch_.cuda.synchronize()
        return yd.cpu().numpy()

    y = run(x)
    assert np.isfinite(y).all()
    assert np.array_equal(run(x * np.float32(4.0)), y * np.float32(4.0))
    assert np.array_equal(run(x * np.float32(0.125)), y * np.float32(0.125))
    perm = rng.permutation(M)
    assert np.array_equal(run(x[perm]), y[perm])
    assert np.array_equal(run(x[:1000], 1000), y[:1000])
    hip.weights_free(h)


@pytest.mark.parametrize("fmt", ["qk256", "i2s"])
def test_prefill_2_digits_then_decode_matches_oracle_1k_and_4k(pkg, hip, oracle, synth, fmt):
    """c4's regime end to end against the ORACLE: the oracle's token-by-token model (oracle/transformer_oracle.c,
    T:398-543 attention over an f32 cache T:1171-1202, 16 host threads dealing heads / rows: bit-identical to one thread,
    tests/test_oracle_threads.py) walks the 4096-token prompt once; on the way it leaves the logits at position 1023 and, at the end, the
    logits of 3 greedy tokens from position 4095 on.  The HIP decoder is then held to them:
      * prefill(1024, digits = 2) (gate|up on the wide tile, the 2560-row launches on narrower ones);
      * prefill(4096, digits = 2) -- EVERY launch on the benchmarked tile -- then 2 decode steps at 4097 / 4098 keys, f32 KV cache;
      * the same with the f16 KV cache (bench.py's c4 default): the decode steps run k_attn_partial<2, true> + k_attn_combine<5>,
        the 128-position f16 form the decoder takes beyond 256 chunk records (5 KV heads x 65 chunks);
    logits cosine >= 0.9999 each, the same greedy tokens."""
    cfg = synth.ModelConfig(**WIDE)
    glob, layers, olayers = _models(synth, fmt, cfg)
    T, n_new = 4096, 3
    prompt = synth.prompt(T, cfg.vocab)
    om = oracle.OracleModel(cfg, olayers, glob, n_threads=16)
    seq = list(prompt)
    o_logits, o_1k = [], None
    for p in range(T + n_new - 1):
        _, logits, _ = om.step(seq[p], want_logits=p >= T - 1 or p == 1023)
        if p == 1023:
            o_1k = logits.copy()
        if p >= T - 1:
            o_logits.append(logits.copy())
            seq.append(oracle.argmax(logits))
    om.close()
    dec = pkg.HostDecoder(cfg)
    for l, w in enumerate(layers):
        dec.set_layer_qk256(l, w) if fmt == "qk256" else dec.set_layer_i2s(l, w, 32)
    dec.set_globals(glob)
    dec.reset()
    dec.feed(prompt[:1024])
    dec.prefill(1024, with_logits=True, digits=2)
    t = hip.matmul_last_tile()  # the last launch of the loop is a 2560-row down-projection: at 1024 rows a narrower tile, and QK256 keeps the
    assert t["digits"] == 2 and t["scale_mode"] == (0 if fmt == "qk256" else 4)  # int8 planes there (the hybrid forward starts at 2497 tokens: Decoder::hybrid_applies)
    assert dec.position() == 1024
    c = cosine(dec.last_logits(), o_1k)
    assert c >= 0.9999, c
    hidden = {}
    for kv16 in (False, True):
        dec.reset()
        dec.set_kv_f16(kv16)
        dec.feed(seq)  # every token forced to the oracle's
        dec.prefill(T, with_logits=True, digits=2)
        tile = hip.matmul_last_tile()
        # the prompt's last matmul is a down-projection: k_gemm_f16a on f16 silu * up rows in 320-row workgroups, both formats
        assert tile == dict(digits=2, wave_tokens=64, waves=4, scale_mode=5 if fmt == "qk256" else 4) and hip.matmul_last_wave_rows() == 80, tile
        assert dec.position() == T
        c = cosine(dec.last_logits(), o_logits[0])
        assert c >= 0.9999, (kv16, c)
        hidden[kv16] = dec.last_hidden().copy()
        for i in range(1, n_new):
            dec.run(1, with_logits=True, use_graph=True)  # keys 4097, 4098: the wide (128-position) attention form
            c = cosine(dec.last_logits(), o_logits[i])
            assert c >= 0.9999, (kv16, i, c)
        assert list(dec.history(T + n_new)) == [int(t) for t in seq], kv16
    assert cosine(hidden[False], hidden[True]) >= 0.99999  # the prompt attention is f16 on the matrix cores either way
    dec.close()
    if fmt != "qk256":
        return
    # ---- the QB32 chain (opt-in, round 5: no row quantiser launch; q|k|v and gate|up on producer-quantised block-scaled rows, LayerNorm after
    # the product): the same prompt, the same oracle logits, the same greedy tokens
    import os

    os.environ["BITNET_HOST_PREFILL_QB32"] = "1"  # read by a Decoder at its first prefill
    try:
        dec = pkg.HostDecoder(cfg)
        for l, w in enumerate(layers):
            dec.set_layer_qk256(l, w)
        dec.set_globals(glob)
        dec.reset()
        dec.set_kv_f16(True)
        dec.feed(seq)
        dec.prefill(T, with_logits=True, digits=2)
        assert dec.last_prefill_path() == 2
        c = cosine(dec.last_logits(), o_logits[0])
        assert c >= 0.9999, c
        assert cosine(dec.last_hidden(), hidden[True]) >= 0.9999
        for i in range(1, n_new):
            dec.run(1, with_logits=True, use_graph=True)
            assert cosine(dec.last_logits(), o_logits[i]) >= 0.9999, i
        assert list(dec.history(T + n_new)) == [int(t) for t in seq]
        dec.reset()
        dec.feed(prompt[:1024])
        dec.prefill(1024, with_logits=True, digits=2)  # short prompts keep the quantiser launches (the producers' 64-token tiles need the hybrid forward)
        assert dec.last_prefill_path() == 0 and cosine(dec.last_logits(), o_1k) >= 0.9999
        dec.close()
    finally:
        del os.environ["BITNET_HOST_PREFILL_QB32"]
