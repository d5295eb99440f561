"""Logits GEMV time vs workgroup count (BITNET_HOST_LOGITS_WGS)."""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
hip = pkg.load(); hip.init(0)
cfg = synth.ModelConfig(**dict(synth.BITNET_2B_4T, n_layers=1, max_pos=256))
dec = pkg.HostDecoder(cfg)
dec.set_layer_qk256(0, synth.make_layer(cfg, 0)); dec.set_globals(synth.make_globals(cfg))
dec.reset(); dec.feed(synth.prompt(8, cfg.vocab)); dec.run(4, with_logits=True)
us, b = dec.probe_kernel(5, 30)
print(os.environ.get("BITNET_HOST_LOGITS_WGS", "1024"), f"{us:.1f} us  {b / us / 1e3:.0f} GB/s", flush=True)
