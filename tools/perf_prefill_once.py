"""One 4096-token prefill per format at the default digit count (for rocprofv3 --kernel-trace --stats)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
hip = pkg.load(); hip.init(0)
fmts = sys.argv[1:] or ["qk256", "i2s"]
for fmt in fmts:
    cfg = synth.ModelConfig(**synth.BITNET_2B_4T); cfg.max_pos = 4200
    dec = pkg.HostDecoder(cfg)
    for l in range(cfg.n_layers):
        w = synth.make_layer(cfg, l, fmt=fmt, block=32)
        dec.set_layer_qk256(l, w) if fmt == "qk256" else dec.set_layer_i2s(l, w, 32)
    dec.set_globals(synth.make_globals(cfg))
    T = 4096
    prompt = synth.prompt(T, cfg.vocab)
    for rep in range(3):
        dec.reset(); dec.feed(prompt)
        ms = dec.prefill(T, with_logits=True, digits=2)
    print(fmt, "prefill ms", round(ms, 2), flush=True)
    dec.close()
