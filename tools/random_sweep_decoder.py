"""Developer tool (GPU box): the whole host decoder (embedding -> layers -> logits -> greedy token) on random small model
configurations, both storage formats, random prompt lengths, prefill or step-by-step prompt, f32 / f16 KV cache, against the oracle's
restated reference transformer (teacher-forced on the oracle's greedy tokens): logits cosine and greedy tokens.
python tools/random_sweep_decoder.py [n] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
synth = importlib.import_module("bitnet-rs_amd.synth")
from oracle import oracle  # noqa: E402
hip = pkg.load(); hip.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
cos = lambda a, b: float(np.dot(a.astype(np.float64), b.astype(np.float64)) / (np.linalg.norm(a.astype(np.float64)) * np.linalg.norm(b.astype(np.float64)) + 1e-300))
bad = 0
for case in range(n_cases):
    hidden = int(rng.choice([512, 1024, 1536]))
    n_heads = hidden // 128
    n_kv = int(rng.choice([d for d in (1, 2, 3, 4, 6, 12) if n_heads % d == 0 and n_heads // d in (1, 2, 4)]))
    ffn = 256 * int(rng.integers(2, 9))
    fmt = str(rng.choice(["qk256", "i2s"]))
    n_prompt, n_new = int(rng.choice([1, 2, 7, 63, 64, 65, 130, 200])), int(rng.integers(2, 6))
    cfg = synth.ModelConfig(hidden=hidden, n_layers=int(rng.integers(1, 4)), n_heads=n_heads, n_kv_heads=n_kv, head_dim=128, ffn=ffn,
                            vocab=int(rng.choice([1024, 2048, 5000])), max_pos=n_prompt + n_new + 8, eps=1e-5, rope_theta=float(rng.choice([10000.0, 500000.0])))
    use_prefill, kv16 = bool(rng.integers(0, 2)) and n_prompt >= 2, bool(rng.integers(0, 2))
    tag = (hidden, n_heads, n_kv, ffn, cfg.n_layers, cfg.vocab, fmt, n_prompt, n_new, "prefill" if use_prefill else "steps", "kv16" if kv16 else "kv32")
    layers = [synth.make_layer(cfg, l, fmt=fmt, block=32) for l in range(cfg.n_layers)]
    glob = synth.make_globals(cfg)
    if fmt == "qk256":
        olayers = layers
    else:  # the oracle model takes dense f32 matrices for the ternary format (as tests/test_prefill_parity.py does)
        tmap = np.array([0, 1, 0, -1], np.float32)
        olayers = []
        for lay in layers:
            d = {"attn_norm": lay["attn_norm"], "ffn_norm": lay["ffn_norm"], "dense": True}
            for name, (rows, cols) in cfg.shapes().items():
                p = lay[name].reshape(rows, cols // 4)
                codes = np.stack([(p >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(rows, cols)
                d[name] = tmap[codes] * np.repeat(lay[name + "_scales"].reshape(rows, cols // 32), 32, axis=1)
            olayers.append(d)
    om = oracle.OracleModel(cfg, olayers, glob, n_threads=8)
    prompt = synth.prompt(n_prompt, cfg.vocab)
    seq, o_logits = list(prompt), []
    for p in range(n_prompt + n_new - 1):
        _, logits, _ = om.step(seq[p])
        o_logits.append(logits)
        if p + 1 >= n_prompt:
            seq.append(oracle.argmax(logits))
    om.close()
    dec = pkg.HostDecoder(cfg)
    try:
        for l, w in enumerate(layers):
            dec.set_layer_qk256(l, w) if fmt == "qk256" else dec.set_layer_i2s(l, w, 32)
        dec.set_globals(glob)
        dec.reset()
        if kv16:
            dec.set_kv_f16(True)
        dec.feed(np.asarray(seq, np.int32))  # teacher-forced on the oracle's own greedy tokens: a near-tie must not fork the sequences
        worst = 1.0
        picks_equal, gap_note = True, ""
        if use_prefill:
            dec.prefill(n_prompt, with_logits=True, digits=int(rng.choice([2, 3, 4])))
            worst = min(worst, cos(dec.last_logits(), o_logits[n_prompt - 1]))
            start = n_prompt
        else:
            start = 0
        # teacher-force the oracle's tokens (a near-tie must not make the sequences diverge): feed the whole oracle sequence
        dec2_tokens = [int(t) for t in seq]
        for p in range(start, n_prompt + n_new - 1):
            dec.run(1, with_logits=True, use_graph=bool(p & 1))
            lg = dec.last_logits()
            worst = min(worst, cos(lg, o_logits[p]))
            if p + 1 >= n_prompt and int(np.argmax(lg)) != dec2_tokens[p + 1]:
                picks_equal = False
                top = np.sort(o_logits[p])[-2:]
                gap_note = f"oracle top-2 gap {float(top[1] - top[0]):.3g} of max |logit| {float(np.max(np.abs(o_logits[p]))):.3g}; max |device - oracle| {float(np.max(np.abs(lg - o_logits[p]))):.3g}"
        floor = 0.9995 if kv16 or use_prefill else 0.9999
        same = picks_equal
        if worst < floor:
            bad += 1
            print("FAIL", tag, "worst cosine", worst, "tokens equal" if same else "tokens differ", flush=True)
        elif not same:
            print("note ", tag, "worst cosine", round(worst, 7), "greedy pick differs from the oracle's:", gap_note, flush=True)
    except pkg.BitNetHipError as e:
        bad += 1
        print("FAIL", tag, repr(e), flush=True)
    dec.close()
print(f"{n_cases - bad}/{n_cases} configurations agree", flush=True)
sys.exit(1 if bad else 0)
