"""Worker for tests/test_dist_device.py: two PROCESSES = two ranks on the one GPU, torch.distributed backend gloo, driving
the real device path of the token-parallel prefill (Decoder::prefill_sharded, C++ host loop) with the all-gather carried by
torch.distributed (bitnet-rs_amd/prefill_parallel.torch_gather)."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, T = sys.argv[1], int(sys.argv[2])
    import torch

    pkg = importlib.import_module("bitnet-rs_amd")
    synth = importlib.import_module("bitnet-rs_amd.synth")
    dist_ = importlib.import_module("bitnet-rs_amd.dist")
    tp = importlib.import_module("bitnet-rs_amd.prefill_parallel")
    os.environ["BITNET_DIST_BACKEND"] = "gloo"
    r = dist_.init("gloo")
    torch.cuda.set_device(0)
    pkg.load().init(0)
    cfg = synth.ModelConfig(hidden=512, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, ffn=1024, vocab=2048, max_pos=640, eps=1e-5, rope_theta=10000.0)
    dec = pkg.HostDecoder(cfg)
    for l in range(cfg.n_layers):
        dec.set_layer_qk256(l, synth.make_layer(cfg, l))
    dec.set_globals(synth.make_globals(cfg))
    prompt = synth.prompt(T, cfg.vocab)
    dec.reset()
    dec.feed(prompt)
    dec.prefill_sharded(T, r.rank, r.world, gather=tp.torch_gather(r.world) if r.world > 1 else None, with_logits=True, digits=3, wire_f16=False)
    pos = dec.position()
    dec.run(2, with_logits=True)  # every rank's cache holds all T positions: every rank can go on decoding
    res = {"rank": r.rank, "world": r.world, "pos": pos, "tokens": [int(t) for t in dec.history(T + 3)[T:]]}
    if r.rank == 0:
        np.save(os.path.join(out_dir, "logits.npy"), dec.last_logits())
    with open(os.path.join(out_dir, f"rank{r.rank}.json"), "w") as f:
        json.dump(res, f)
    dec.close()
    dist_.finalize(r)


if __name__ == "__main__":
    main()
