"""4096-token prefills at the default digit count (for rocprofv3 --kernel-trace --stats and the --pmc passes).

    python3 tools/perf_prefill_once.py [qk256|i2s] [repetitions = 3] [layers = 30] [tokens = 4096]

The counter passes of tools/profile_round.sh run ONE repetition on 4 layers: the same launches per layer at the same
shapes, few enough dispatches for rocprofv3's counter collection."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bitnet-rs_amd"); synth = importlib.import_module("bitnet-rs_amd.synth")
hip = pkg.load(); hip.init(0)
fmt = sys.argv[1] if len(sys.argv) > 1 else "qk256"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cfg = synth.ModelConfig(**synth.BITNET_2B_4T); cfg.max_pos = 4200; cfg.n_layers = layers
dec = pkg.HostDecoder(cfg)
for l in range(cfg.n_layers):
    w = synth.make_layer(cfg, l, fmt=fmt, block=32)
    dec.set_layer_qk256(l, w) if fmt == "qk256" else dec.set_layer_i2s(l, w, 32)
dec.set_globals(synth.make_globals(cfg))
T = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
prompt = synth.prompt(T, cfg.vocab)
for rep in range(reps):
    dec.reset(); dec.feed(prompt)
    ms = dec.prefill(T, with_logits=True, digits=2)
print(fmt, "layers", layers, "tokens", T, "prefill ms", round(ms, 2), "tile", hip.matmul_last_tile(), flush=True)
dec.close()
