"""GPU parity at the shapes and storage format BASELINE.json's metric is quoted on
(configs[1]: bitnet-b1.58-2B-4T, BitNet32-F16 = ternary codes + one f16 scale per 32 weights):
every fused GEMV of the decode step at hidden 2560 / ffn 6912 against the oracle chain
(oracle.layernorm -> oracle.i2s_matmul [K/cpu/quantized_matmul.rs:57-96] -> silu*mul [T:756-781]),
i.e. the kernel instances bench.py times (k_gemv_mfma<8,5,..> for the paired gate|up at K = 2560,
RING 2 for q|k|v and o, RING 4 for down), then whole decode tokens of a 2-layer model of those
widths in that format -- through set_layer_i2s and through the inline-f16 GGUF loader."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROJ = ("q", "k", "v", "o", "gate", "up", "down")
WIDE = dict(hidden=2560, n_layers=2, n_heads=20, n_kv_heads=5, head_dim=128, ffn=6912, vocab=4096, max_pos=48, eps=1e-5, rope_theta=500000.0)


def cosine(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.fixture(scope="module")
def synth(pkg):
    return importlib.import_module("bitnet-rs_amd.synth")


@pytest.fixture(scope="module")
def torch_():
    import torch

    return torch


def approx_eq_with_len(got, want, cols):
    """crates/bitnet-models/tests/helpers/qk256_tolerance.rs: abs 2e-4*sqrt(cols/256) (<= 1e-3) OR rel 2e-2."""
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = min(2e-4 * np.sqrt(cols / 256), 1e-3)
    rel = diff / np.maximum(np.maximum(np.abs(got), np.abs(want)), 1e-30)
    return (diff < tol) | (rel < 2e-2)


def dense_of(packed, scales, rows, cols):
    """t(code) * scale[block] as dense f32 [rows, cols]; t: 0->0, 1->+1, 2->0, 3->-1 (K/cuda/quantized_matmul.rs:19-27)."""
    p = packed.reshape(rows, cols // 4)
    codes = np.stack([(p >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
    return np.array([0, 1, 0, -1], np.float32)[codes] * np.repeat(scales.reshape(rows, cols // 32), 32, axis=1)


@pytest.fixture(scope="module")
def layer0(synth):
    cfg = synth.ModelConfig(**synth.BITNET_2B_4T)
    return cfg, synth.make_layer(cfg, 0, fmt="i2s", block=32)


def test_gateup_ln_silu_2b4t_bitnet32_f16(hip, oracle, torch_, layer0):
    """The bench's dominant kernel instance: interleaved gate|up 13824x2560, f16-exact 32-block scales,
    LayerNorm bound (applied after the product) + silu(gate)*up."""
    cfg, lay = layer0
    K, F = cfg.hidden, cfg.ffn
    for name in ("gate", "up"):
        s = lay[name + "_scales"]
        assert np.array_equal(s.astype(np.float16).astype(np.float32), s)  # f16-exact -> the f16 scale tiles
    hg = hip.weights_upload_i2s(lay["gate"], lay["gate_scales"], F, K, 32)
    hu = hip.weights_upload_i2s(lay["up"], lay["up_scales"], F, K, 32)
    hgu = hip.weights_concat([hg, hu], interleave16=True)
    gamma = lay["ffn_norm"]
    gd = torch_.from_numpy(gamma).cuda()
    hip.weights_bind_ln(hgu, gd)
    rng = np.random.default_rng(3)
    for mean, std in ((0.0, 1.0), (0.3, 6.0)):
        x = rng.normal(mean, std, K).astype(np.float32)
        xn = oracle.layernorm(x, gamma, cfg.eps)
        g = oracle.i2s_matmul(xn, lay["gate"], lay["gate_scales"], 1, F, K, 32)
        u = oracle.i2s_matmul(xn, lay["up"], lay["up_scales"], 1, F, K, 32)
        want = (g / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32) * u
        yd = torch_.full((F,), float("nan"), device="cuda")
        hip.gemv_fused_dev(hgu, torch_.from_numpy(x).cuda(), yd, 1, ln_gamma=gd, ln_eps=cfg.eps, flags=1)
        torch_.cuda.synchronize()
        got = yd.cpu().numpy()
        assert np.max(np.abs(got - want)) <= 3e-5 * np.max(np.abs(want)) + 1e-6, np.max(np.abs(got - want))
        assert cosine(got, want) >= 0.99999
    for h in (hg, hu, hgu):
        hip.weights_free(h)


def test_qkv_o_down_2b4t_bitnet32_f16(hip, oracle, torch_, layer0):
    """q|k|v concat 3840x2560 with the bound LayerNorm, o 2560x2560 + residual, down 2560x6912 + residual."""
    cfg, lay = layer0
    K, F = cfg.hidden, cfg.ffn
    rng = np.random.default_rng(4)
    x = rng.normal(0.1, 2.0, K).astype(np.float32)
    # q|k|v
    hs = [hip.weights_upload_i2s(lay[n], lay[n + "_scales"], cfg.shapes()[n][0], K, 32) for n in ("q", "k", "v")]
    hqkv = hip.weights_concat(hs)
    gamma = lay["attn_norm"]
    gd = torch_.from_numpy(gamma).cuda()
    hip.weights_bind_ln(hqkv, gd)
    xn = oracle.layernorm(x, gamma, cfg.eps)
    want = np.concatenate([oracle.i2s_matmul(xn, lay[n], lay[n + "_scales"], 1, cfg.shapes()[n][0], K, 32) for n in ("q", "k", "v")])
    yd = torch_.full((want.size,), float("nan"), device="cuda")
    hip.gemv_fused_dev(hqkv, torch_.from_numpy(x).cuda(), yd, 1, ln_gamma=gd, ln_eps=cfg.eps)
    torch_.cuda.synchronize()
    got = yd.cpu().numpy()
    assert np.max(np.abs(got - want)) <= 3e-5 * np.max(np.abs(want)) + 1e-6
    assert cosine(got, want) >= 0.99999
    for h in hs + [hqkv]:
        hip.weights_free(h)
    # o and down: plain f32 activations + residual
    for name, cols in (("o", K), ("down", F)):
        a = rng.normal(0, 1.5, cols).astype(np.float32)
        res = rng.normal(0, 1, K).astype(np.float32)
        h = hip.weights_upload_i2s(lay[name], lay[name + "_scales"], K, cols, 32)
        want = oracle.i2s_matmul(a, lay[name], lay[name + "_scales"], 1, K, cols, 32)
        yd = torch_.full((K,), float("nan"), device="cuda")
        hip.gemv_fused_dev(h, torch_.from_numpy(a).cuda(), yd, 1, residual=torch_.from_numpy(res).cuda())
        torch_.cuda.synchronize()
        got = yd.cpu().numpy() - res
        assert np.all(approx_eq_with_len(got, want, cols)), name
        assert np.max(np.abs(got - want)) <= 3e-5 * np.max(np.abs(want)) + 2e-6, name
        hip.weights_free(h)


def _oracle_dense_layers(cfg, layers):
    dense = []
    for lay in layers:
        d = {"attn_norm": lay["attn_norm"], "ffn_norm": lay["ffn_norm"], "dense": True}
        for name in PROJ:
            rows, cols = cfg.shapes()[name]
            d[name] = dense_of(lay[name], lay[name + "_scales"], rows, cols)
        dense.append(d)
    return dense


def _decode_against(om, dec, oracle, prompt, n_new, graph_every=2):
    n_prompt = len(prompt)
    seq = list(prompt)
    dec.reset()
    dec.feed(prompt)
    worst = 1.0
    for p in range(n_prompt + n_new - 1):
        _, logits, _ = om.step(seq[p])
        if p + 1 >= n_prompt:
            seq.append(oracle.argmax(logits))
        dec.run(1, with_logits=True, use_graph=(p % graph_every == 0))
        got = dec.last_logits()
        c = cosine(got, logits)
        worst = min(worst, c)
        assert c >= 0.9999, (p, c)
        assert np.max(np.abs(got - logits)) <= 2e-3 * np.max(np.abs(logits)), p
    assert list(dec.history(n_prompt + n_new)) == [int(t) for t in seq], worst
    return seq


def test_wide_decode_bitnet32_f16(pkg, hip, oracle, synth):
    """Whole decode tokens at hidden 2560 / ffn 6912 in the BitNet32-F16 storage (set_layer_i2s, block 32): the
    step graph's kernel instances are the 2B-4T ones; oracle = the restated step on dense scale*t(code) matrices."""
    cfg = synth.ModelConfig(**WIDE)
    layers = [synth.make_layer(cfg, l, fmt="i2s", block=32) for l in range(cfg.n_layers)]
    glob = synth.make_globals(cfg)
    om = oracle.OracleModel(cfg, _oracle_dense_layers(cfg, layers), glob, n_threads=8)
    dec = pkg.HostDecoder(cfg)
    for l, w in enumerate(layers):
        dec.set_layer_i2s(l, w, 32)
    dec.set_globals(glob)
    seq = _decode_against(om, dec, oracle, synth.prompt(4, cfg.vocab), 6)
    assert len(set(seq[4:])) > 1  # the synthetic model is not a fixed point of its own embedding
    dec.close()
    om.close()


def test_wide_decode_inline_f16_gguf(pkg, hip, oracle, synth, tmp_path):
    """The same widths through the GGUF loader's inline-f16 flavour (10 B per 32 weights: 8 B codes + f16 scale,
    M/quant/i2s.rs:66-140), code map {-2,-1,0,1}... as the loader assigns it; oracle fed by oracle/gguf_oracle.py."""
    from oracle import gguf_oracle as G
    from tests import gguf_util as W

    cfg = synth.ModelConfig(**dict(WIDE, vocab=2048))
    glob = synth.make_globals(cfg)
    tensors = [("token_embd.weight", (cfg.vocab, cfg.hidden), W.F16, glob["embed_f16"].tobytes()),
               ("output_norm.weight", (cfg.hidden,), W.F32, glob["final_norm"].tobytes())]
    rng = np.random.default_rng(21)
    for l in range(cfg.n_layers):
        lay = synth.make_layer(cfg, l, fmt="i2s", block=32)
        tensors.append((f"blk.{l}.attn_norm.weight", (cfg.hidden,), W.F32, lay["attn_norm"].tobytes()))
        tensors.append((f"blk.{l}.ffn_norm.weight", (cfg.hidden,), W.F32, lay["ffn_norm"].tobytes()))
        for name in PROJ:
            rows, cols = cfg.shapes()[name]
            nb = rows * cols // 32
            codes = rng.integers(0, 256, (nb, 8), dtype=np.uint8)
            scales = (rng.uniform(0.2, 1.0, nb) * 0.17).astype(np.float16)
            tensors.append((f"blk.{l}.{W.BLK[name]}.weight", (rows, cols), W.I2_S, W.inline_f16_blocks(codes, scales)))
    data = W.write_gguf(W.model_kvs(cfg), tensors)
    path = tmp_path / "wide_inline_f16.gguf"
    path.write_bytes(data)
    g = G.parse(data)
    ocfg = G.extract_config(g)
    f32 = lambda t: np.frombuffer(g.tensor_bytes(t), "<f4", count=int(np.prod(t.shape))).astype(np.float32)
    olayers = []
    for l in range(cfg.n_layers):
        d = {"attn_norm": f32(g.info(f"blk.{l}.attn_norm.weight")), "ffn_norm": f32(g.info(f"blk.{l}.ffn_norm.weight")), "dense": True}
        for name in PROJ:
            r = G.load_i2s(g, g.info(f"blk.{l}.{W.BLK[name]}.weight"), ocfg)
            assert r[0] == "f32" and r[1].shape == cfg.shapes()[name]
            d[name] = r[1]
        olayers.append(d)
    emb = np.frombuffer(g.tensor_bytes(g.info("token_embd.weight")), np.uint16, count=cfg.vocab * cfg.hidden)
    om = oracle.OracleModel(cfg, olayers, {"embed_f16": emb, "final_norm": glob["final_norm"]}, n_threads=8)
    f = pkg.GgufFile(path=str(path))
    dec = pkg.HostDecoder(cfg)
    dec.load_gguf(f)
    f.close()
    _decode_against(om, dec, oracle, synth.prompt(3, cfg.vocab), 4, graph_every=1)
    dec.close()
    om.close()


@pytest.mark.parametrize("fmt", ["i2s", "qk256"])
def test_fast_step_matches_unfused_exact_kernel_step(pkg, hip, synth, fmt):
    """Decoder::run_reference = the same decode step unfused, in the reference's op order (T:977-1134), every projection on
    the reference-order kernel (bit-identical to the scalar CPU loops, tests/test_gpu_parity.py).  The fast step (fused
    epilogues, LayerNorm after the product, hipGraph) must give the same logits and the same greedy tokens; bench.py
    repeats this check at the full 30-layer size next to the timed run."""
    cfg = synth.ModelConfig(**dict(WIDE, n_layers=3))
    dec = pkg.HostDecoder(cfg)
    for l in range(cfg.n_layers):
        w = synth.make_layer(cfg, l, fmt=fmt, block=32)
        dec.set_layer_i2s(l, w, 32) if fmt == "i2s" else dec.set_layer_qk256(l, w)
    dec.set_globals(synth.make_globals(cfg))
    prompt = synth.prompt(5, cfg.vocab)
    runs = []
    for ref in (True, False):
        dec.reset()
        dec.feed(prompt)
        logits = []
        for p in range(5 + 5 - 1):
            dec.run_reference(1, with_logits=True) if ref else dec.run(1, with_logits=True, use_graph=True)
            logits.append(dec.last_logits().copy())
        runs.append((logits, list(dec.history(10))))
    for p, (a, b) in enumerate(zip(*[r[0] for r in runs])):
        assert cosine(a, b) >= 0.99999, (fmt, p, cosine(a, b))
        assert np.max(np.abs(a - b)) <= 1e-3 * np.max(np.abs(a)), (fmt, p)
    assert runs[0][1] == runs[1][1]
    dec.close()


def test_two_host_threads_share_the_library(hip, oracle, torch_):
    """The provider trait is Send + Sync (K/lib.rs:39): upload / gemv / free from two host threads at once, each on its own
    stream, one of them freeing a handle the other is still launching on (reference-counted handles: the launch
    finishes on a live matrix, later calls see 'unknown weights handle')."""
    import threading

    rows, cols = 640, 2560
    stride = cols // 256 * 64
    rng = np.random.default_rng(77)
    qs = [rng.integers(0, 256, rows * stride, dtype=np.uint8) for _ in range(2)]
    x = rng.uniform(-1, 1, cols).astype(np.float32)
    want = [oracle.gemv_qk256(q, x, rows, cols, stride) for q in qs]
    xd = torch_.from_numpy(x).cuda()
    shared = hip.weights_upload_qk256(qs[0], rows, cols, stride)
    errors = []

    def worker(i):
        try:
            s = torch_.cuda.Stream()
            for it in range(40):
                h = hip.weights_upload_qk256(qs[i], rows, cols, stride)
                with torch_.cuda.stream(s):
                    # (the output's zero fill rides on the SAME stream as the launch: torch's side streams do not synchronise with the
                    # default stream, and a fill left there may land after the product -- seen once as a zero vector under two threads)
                    y = torch_.zeros(rows, device="cuda")
                    hip.matmul_kernel_dev(h, xd, y, 1, 3 if it % 2 else 0, stream=s.cuda_stream)
                s.synchronize()
                if not np.all(approx_eq_with_len(y.cpu().numpy(), want[i], cols)):
                    errors.append(("value", i, it))
                hip.weights_free(h)
                if i == 0:  # hammer the shared handle while the other thread frees it half way
                    try:
                        hip.gemv_dev(shared, xd, y, stream=s.cuda_stream)
                        s.synchronize()
                    except Exception as e:  # noqa: BLE001
                        if "unknown weights handle" not in str(e):
                            errors.append(("shared", str(e)))
                elif it == 20:
                    hip.weights_free(shared)
        except Exception as e:  # noqa: BLE001
            errors.append(("exc", i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_one_copy_of_the_weights_on_the_device(hip, pkg, oracle, torch_, layer0):
    """A handle keeps the streaming layout only (both layouts = 2x the model's bytes).  The reference-order
    kernels rebuild the reference layout on first use -- bit for bit: the exact kernel still equals the scalar oracle --
    and bitnet_hip_weights_trim drops it again."""
    cfg, lay = layer0
    K = cfg.hidden
    rows = cfg.shapes()["k"][0]
    h = hip.weights_upload_i2s(lay["k"], lay["k_scales"], rows, K, 32)
    algo = hip.weights_info(h)[2]
    assert algo == rows * K // 4 + 2 * rows * K // 32
    assert hip.weights_device_bytes(h) == algo  # 1 KiB code tiles + f16 scale tiles, nothing else
    x = np.random.default_rng(8).uniform(-10, 10, K).astype(np.float32)
    xd = torch_.from_numpy(x).cuda()
    want = oracle.i2s_matmul(x, lay["k"], lay["k_scales"], 1, rows, K, 32)
    y = torch_.empty(rows, device="cuda")
    hip.matmul_kernel_dev(h, xd, y, 1, pkg.KERNEL_EXACT)
    torch_.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), want)                 # rebuilt codes + scales are the uploaded ones
    assert hip.weights_device_bytes(h) > 2 * algo                 # reference layout + f32 scales are back
    hip.weights_trim(h)
    assert hip.weights_device_bytes(h) == algo
    hip.matmul_kernel_dev(h, xd, y, 1, pkg.KERNEL_VALU)          # and once more from the trimmed state
    torch_.cuda.synchronize()
    assert np.all(approx_eq_with_len(y.cpu().numpy(), want, K))
    # concat of trimmed parts (tile path) then the exact kernel on the fused matrix
    h2 = hip.weights_upload_i2s(lay["v"], lay["v_scales"], rows, K, 32)
    hip.weights_trim(h)
    hc = hip.weights_concat([h, h2])
    assert hip.weights_device_bytes(hc) == 2 * algo
    yc = torch_.empty(2 * rows, device="cuda")
    hip.matmul_kernel_dev(hc, xd, yc, 1, pkg.KERNEL_EXACT)
    torch_.cuda.synchronize()
    want2 = oracle.i2s_matmul(x, lay["v"], lay["v_scales"], 1, rows, K, 32)
    assert np.array_equal(yc.cpu().numpy(), np.concatenate([want, want2]))
    # QK256: codes only
    qs = np.random.default_rng(9).integers(0, 256, rows * K // 4, dtype=np.uint8)
    hq = hip.weights_upload_qk256(qs, rows, K, K // 4)
    assert hip.weights_device_bytes(hq) == rows * K // 4
    for hh in (h, h2, hc, hq):
        hip.weights_free(hh)
