"""The oracle's host threads only DEAL independent work (attention heads, rows of a dense projection, logits rows): every
output element is computed by one thread in the single-thread order, so the threaded oracle (what the long-context GPU parity
tests run: 4096-token prompts, tests/test_bench_prefill_instance.py) equals the one-thread restatement bit for bit."""
import importlib

import numpy as np

SMALL = dict(hidden=512, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, ffn=1024, vocab=2048, max_pos=320, eps=1e-5, rope_theta=10000.0)


def test_threaded_oracle_is_bit_identical_to_one_thread():
    synth = importlib.import_module("bitnet-rs_amd.synth")
    from oracle import oracle

    cfg = synth.ModelConfig(**SMALL)
    glob = synth.make_globals(cfg)
    qk = [synth.make_layer(cfg, l) for l in range(cfg.n_layers)]
    tmap = np.array([0, 1, 0, -1], np.float32)
    dense = []
    for l in range(cfg.n_layers):  # the dense-matrix form the BitNet32-F16 tests hand the oracle
        lay = synth.make_layer(cfg, l, fmt="i2s", block=32)
        d = {"attn_norm": lay["attn_norm"], "ffn_norm": lay["ffn_norm"], "dense": True}
        for name, (rows, cols) in cfg.shapes().items():
            pk = lay[name].reshape(rows, cols // 4)
            codes = np.stack([(pk >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
            d[name] = tmap[codes] * np.repeat(lay[name + "_scales"].reshape(rows, cols // 32), 32, axis=1)
        dense.append(d)
    prompt = synth.prompt(260, cfg.vocab)  # attention deals heads to threads from 128 keys on
    for layers in (qk, dense):
        outs = []
        for nt in (1, 5):
            om = oracle.OracleModel(cfg, layers, glob, n_threads=nt)
            got = []
            for p in range(260):
                hidden, logits, _ = om.step(int(prompt[p]), want_logits=p in (100, 259))
                if p in (100, 259):
                    got.append((hidden.copy(), logits.copy()))
            om.close()
            outs.append(got)
        for (h1, l1), (h5, l5) in zip(*outs):
            assert np.array_equal(h1, h5) and np.array_equal(l1, l5)


def test_ternary_layer_kind_is_the_dense_kind_bit_for_bit():
    """OracleModel's "ternary" layers (packed codes + f32 block scales, W[r, c] = t(code) * scale multiplied on the fly in the loop
    of i2s_matmul_f32, K/cpu/quantized_matmul.rs:57-96) against the dense f32 form the 32-element flavours take in the reference
    (dequantised at load, M/gguf_simple.rs:1260-1285, then a plain Linear): same values in the same order -> identical hidden
    states and logits.  The ternary kind is what the full-depth (30-layer) GPU parity runs hand the oracle: 26 MB per layer
    instead of 278 MB.  Also: one projection of that kind against bo_i2s_matmul_f32 itself."""
    synth = importlib.import_module("bitnet-rs_amd.synth")
    from oracle import oracle

    cfg = synth.ModelConfig(**SMALL)
    glob = synth.make_globals(cfg)
    tmap = np.array([0, 1, 0, -1], np.float32)
    tern, dense = [], []
    for l in range(cfg.n_layers):
        lay = synth.make_layer(cfg, l, fmt="i2s", block=32)
        tern.append(dict(lay, ternary=32))
        d = {"attn_norm": lay["attn_norm"], "ffn_norm": lay["ffn_norm"], "dense": True}
        for name, (rows, cols) in cfg.shapes().items():
            pk = lay[name].reshape(rows, cols // 4)
            codes = np.stack([(pk >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
            d[name] = tmap[codes] * np.repeat(lay[name + "_scales"].reshape(rows, cols // 32), 32, axis=1)
        dense.append(d)
    prompt = synth.prompt(12, cfg.vocab)
    outs = []
    for layers, nt in ((dense, 1), (tern, 1), (tern, 7)):
        om = oracle.OracleModel(cfg, layers, glob, n_threads=nt)
        outs.append([tuple(a.copy() for a in om.step(int(t), want_trace=True)) for t in prompt])
        om.close()
    for a, b, c in zip(*outs):
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(x, z)
    # the first projection of the first token, by the a11 function itself
    lay = tern[0]
    x = oracle.layernorm(glob["embed_f16"].view(np.float16).reshape(cfg.vocab, cfg.hidden)[int(prompt[0])].astype(np.float32), lay["attn_norm"], cfg.eps)
    rows, cols = cfg.shapes()["gate"]
    want = oracle.i2s_matmul(x, lay["gate"], lay["gate_scales"], 1, rows, cols, 32)
    acc = np.zeros(rows, np.float32)
    for c in range(cols):  # left-to-right f32 accumulation, no fused multiply-add: the dense loop
        acc = (acc + x[c] * dense[0]["gate"][:, c]).astype(np.float32)
    assert np.array_equal(want, acc)


def test_pool_keeps_its_workers():
    """bo_parallel_for's workers are created once per process (the per-call pthread_create of the earlier threaded GEMV made 'all
    cores' lose to 4 threads: bench.py cpu_baseline)."""
    from oracle import oracle

    rows, cols = 640, 2560
    rng = np.random.default_rng(1)
    qs = rng.integers(0, 256, rows * cols // 4, dtype=np.uint8)
    x = rng.uniform(-1, 1, cols).astype(np.float32)
    if not oracle.have_avx2():
        return
    one = oracle.gemv_qk256(qs, x, rows, cols, cols // 4, impl="avx2")
    for nt in (3, 8, 3, 8):
        assert np.array_equal(oracle.gemv_qk256(qs, x, rows, cols, cols // 4, impl="avx2_mt", threads=nt), one)
    n = oracle.lib().bo_pool_workers()
    assert 7 <= n <= 63
    oracle.gemv_qk256(qs, x, rows, cols, cols // 4, impl="avx2_mt", threads=8)
    assert oracle.lib().bo_pool_workers() == n
