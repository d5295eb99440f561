import importlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
hip = pkg.load(); hip.init(0)
for fmt in ("qk256", "i2s"):
    cfg = synth.ModelConfig(**synth.BITNET_2B_4T); cfg.max_pos = 4200
    dec = pkg.HostDecoder(cfg)
    for l in range(cfg.n_layers):
        w = synth.make_layer(cfg, l, fmt=fmt, block=32)
        dec.set_layer_qk256(l, w) if fmt == "qk256" else dec.set_layer_i2s(l, w, 32)
    dec.set_globals(synth.make_globals(cfg))
    T = 4096
    prompt = synth.prompt(T, cfg.vocab)
    res = {}
    for d in (4, 3, 2):
        dec.reset(); dec.feed(prompt)
        dec.prefill(T, with_logits=True, digits=d)
        dec.reset(); dec.feed(prompt)
        ms = dec.prefill(T, with_logits=True, digits=d)
        res[d] = (dec.last_logits().astype(np.float64), ms, int(dec.history(T + 1)[T]))
    for d in (3, 2):
        a, b = res[4][0], res[d][0]
        print(fmt, "digits", d, "ms", round(res[d][1], 2), "cos vs 4 digits", a @ b / np.linalg.norm(a) / np.linalg.norm(b), "max rel", np.max(np.abs(a - b)) / np.max(np.abs(a)), "token same", res[d][2] == res[4][2])
    print(fmt, "digits 4 ms", round(res[4][1], 2))
    dec.close()
