"""Prefill time of the BitNet32-F16 (c2) storage: 32-element block scales take the many-rows GEMV
(one launch per matrix, grid.y = rows) instead of the tiled matmul."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
hip = pkg.load(); hip.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = synth.ModelConfig(**dict(synth.BITNET_2B_4T, max_pos=n + 64))
dec = pkg.HostDecoder(cfg)
for l in range(cfg.n_layers):
    dec.set_layer_i2s(l, synth.make_layer(cfg, l, fmt="i2s", block=32), 32)
dec.set_globals(synth.make_globals(cfg))
p = synth.prompt(n, cfg.vocab)
for _ in range(2):
    dec.reset(); dec.feed(p)
    ms = dec.prefill(n, with_logits=True, digits=3)
print(f"c2-format prefill of {n} tokens: {ms:.1f} ms = {n / ms * 1e3:.0f} tok/s", flush=True)
