"""Worker for tests/test_dist_gloo.py::test_eight_rank_gloo_plan_gather_order_and_c5_line: world 8 on CPU (gloo), the exact
helpers bench.py's N > 1 prefill part uses (bitnet-rs_amd/prefill_parallel.py: zigzag plan, the rank-major gathered buffer read
through zz_row, ranks_seen, the phases object, the prefill_c5 line) with marker rows instead of k|v and made-up phase times."""
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
    import torch.distributed as dist

    dist_ = importlib.import_module("bitnet-rs_amd.dist")
    tp = importlib.import_module("bitnet-rs_amd.prefill_parallel")
    synth = importlib.import_module("bitnet-rs_amd.synth")
    r = dist_.init("gloo")
    plan = tp.zigzag_plan(T, r.world)
    mine = tp.local_positions(plan[r.rank])
    # this rank's k|v rows: [position, 3 * position + 1] -- what the per-layer all-gather moves (rows in LOCAL order)
    send = torch.from_numpy(np.stack([mine, 3 * mine + 1], axis=1).astype(np.float32))
    recv = torch.empty(r.world * len(mine), 2)
    dist.all_gather_into_tensor(recv, send)
    # the attention kernel never un-permutes: it reads position p at row zz_row(p) of the gathered buffer
    rows = np.array([tp.zz_row(p, T, r.world) for p in range(T)])
    got = recv.numpy()[rows]
    order_ok = bool(np.array_equal(got[:, 0], np.arange(T, dtype=np.float32)) and np.array_equal(got[:, 1], 3 * np.arange(T, dtype=np.float32) + 1))
    # and the explicit scatter gives the same absolute order
    kv_all = torch.full((T, 2), -1.0)
    tp.scatter_gathered(recv.view(r.world, len(mine), 2), plan, kv_all)
    order_ok = order_ok and bool(np.array_equal(kv_all.numpy(), got))
    causal_work = int(sum(int(p) + 1 for p in mine))  # keys each of this rank's queries attends to: the zigzag's balance
    seen = tp.count_ranks(r.world)
    # made-up per-layer phase medians: rank 5 is the slowest by its compute phases (its large gather figure must not decide)
    ph = {"matmul_us": 400.0 + r.rank, "attention_us": 100.0 + (50.0 if r.rank == 5 else 0.0), "gather_wait_us": 3.0, "gather_us": 30.0 + (500.0 if r.rank == 2 else 0.0)}
    phases = tp.assemble_phases(r.rank, r.world, ph)
    elapsed = dist_.timed_region(r, lambda: None) + 0.0123 * 2
    res = {"rank": r.rank, "world": r.world, "order_ok": order_ok, "causal_work": causal_work, "seen": seen, "phases": phases, "n_local": int(len(mine))}
    if r.rank == 0:
        cfg = synth.ModelConfig(**synth.BITNET_2B_4T)
        res["line"] = tp.c5_line(r.world, T, 2, 0.0246, cfg, 2, seen, 4242, {"same_first_token": True}, phases,
                                 "torch.distributed all_gather_into_tensor (gloo, host-synchronised)", "2.22.3")
    with open(os.path.join(out_dir, f"rank{r.rank}.json"), "w") as f:
        json.dump(res, f)
    dist_.finalize(r)


if __name__ == "__main__":
    main()
