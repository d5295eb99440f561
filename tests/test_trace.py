"""The activation trace hook (reference: crates/bitnet-trace/src/lib.rs).  CPU: the BLAKE3 the records carry
against the specification's published test vectors (input byte i = i % 251).  GPU: one traced decode step leaves one record
per stage in the reference's JSON form, consistent with what the decoder itself reports."""
import importlib
import json
import os

import numpy as np
import pytest

# official BLAKE3 test vectors (test_vectors.json of the specification's reference implementation), `hash` field, first 32 bytes
VECTORS = {
    0: "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
    1: "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
    1023: "10108970eeda3eb932baac1428c7a2163b0e924c9a9e25b35bba72b28f70bd11",
    1024: "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
    1025: "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
    2048: "e776b6028c7cd22a4d0ba182a8bf62205d2ef576467e838ed6f2529b85fba24a",
    2049: "5f4d72f40d7a5f82b15ca2b2e44b1de3c2ef86c426c95c1af0b6879522563030",
    3072: "b98cb0ff3623be03326b373de6b9095218513e64f1ee2edd2525c7ad1e5cffd2",
    3073: "7124b49501012f81cc7f11ca069ec9226cecb8a2c850cfe644e327d22d3e1cd3",
    4096: "015094013f57a5277b59d8475c0501042c0b642e531b0a1c8f58d2163229e969",
    8192: "aae792484c8efe4f19e2ca7d371d8c467ffb10748d8a5a1ae579948f718a2a63",
}


def test_blake3_matches_the_published_vectors(pkg):
    for n, want in VECTORS.items():
        data = bytes(i % 251 for i in range(n))
        assert pkg.blake3_hex(data) == want, n
    assert pkg.blake3_hex(b"abc") == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"


@pytest.mark.gpu
@pytest.mark.parametrize("act_mode", [1, 0])
def test_trace_step_records(pkg, hip, tmp_path, act_mode):
    synth = importlib.import_module("bitnet-rs_amd.synth")
    cfg = synth.ModelConfig(hidden=512, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, ffn=1024, vocab=2048, max_pos=64, eps=1e-5, rope_theta=10000.0)
    dec = pkg.HostDecoder(cfg)
    for l in range(cfg.n_layers):
        dec.set_layer_qk256(l, synth.make_layer(cfg, l))
    dec.set_globals(synth.make_globals(cfg))
    dec.set_act_mode(act_mode)
    prompt = synth.prompt(3, cfg.vocab)
    dec.reset()
    dec.feed(prompt)
    dec.run(2, with_logits=False)
    d = str(tmp_path / f"trace{act_mode}")
    dec.trace_step(d, with_logits=True)  # position 2
    assert dec.position() == 3
    files = sorted(os.listdir(d))
    want = {"t2_embeddings.trace", "t2_all_layers_out.trace", "t2_logits.trace"}
    for l in range(cfg.n_layers):
        want |= {f"t2_blk{l}_{s}.trace" for s in ("q_proj", "k_proj", "v_proj", "attn_out", "attn_residual", "ffn_hidden", "ffn_out")}
    assert set(files) == want
    rec = {f: json.load(open(os.path.join(d, f))) for f in files}
    for f, r in rec.items():
        assert set(r) == {"name", "shape", "dtype", "blake3", "rms", "num_elements", "seq", "layer", "stage"}  # TraceRecord, bitnet-trace/src/lib.rs:46-70
        assert r["dtype"] == "F32" and r["shape"] == [1, r["num_elements"]] and r["seq"] == 2 and len(r["blake3"]) == 64
        assert r["name"].replace("/", "_") + ".trace" == f  # sanitize_filename
        assert np.isfinite(r["rms"]) and r["rms"] > 0
    logits = dec.last_logits()
    lr = rec["t2_logits.trace"]
    assert lr["name"] == "t2/logits" and lr["layer"] == -1 and lr["stage"] == "logits" and lr["num_elements"] == cfg.vocab
    assert lr["blake3"] == pkg.blake3_hex(logits.astype("<f4").tobytes())
    assert abs(lr["rms"] - float(np.sqrt(np.mean(logits.astype(np.float64) ** 2)))) <= 1e-12 * max(1.0, lr["rms"])
    hid = dec.last_hidden()
    assert rec["t2_all_layers_out.trace"]["blake3"] == pkg.blake3_hex(hid.astype("<f4").tobytes())
    assert rec["t2_blk1_ffn_out.trace"]["blake3"] == rec["t2_all_layers_out.trace"]["blake3"]
    assert rec["t2_blk0_q_proj.trace"]["num_elements"] == cfg.n_heads * cfg.head_dim and rec["t2_blk1_ffn_hidden.trace"]["num_elements"] == cfg.ffn
    # the traced step is the production step (it only takes the two-kernel attention form, whose output the merging
    # o-projection of short contexts never materialises): an untraced run from the same state gives the same logits
    dec.reset()
    dec.feed(prompt)
    dec.run(2, with_logits=False)
    dec.run(1, with_logits=True, use_graph=False)
    a, b = dec.last_logits().astype(np.float64), logits.astype(np.float64)
    assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) >= 0.999999
    dec.close()
