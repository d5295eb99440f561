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
