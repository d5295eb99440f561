"""Worker for tests/test_dist_gloo.py: launched with torch.distributed.run, backend gloo.
Exercises the exact helpers bench.py uses for N > 1 (barrier, max-over-ranks timing, whole-job
aggregation, row sharding) with a fake per-rank workload of known duration."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dist_ = importlib.import_module("bitnet-rs_amd.dist")


def main():
    out_dir = sys.argv[1]
    r = dist_.init("gloo")
    steps = 10
    sleep_s = 0.05 * (r.rank + 1)  # rank 1 is slower: the job time must be ITS time

    elapsed = dist_.timed_region(r, lambda: time.sleep(sleep_s))
    value = dist_.aggregate_throughput(r, steps, elapsed)
    lo, hi = dist_.shard_rows(6912, r.world, r.rank, multiple=16)
    lo256, hi256 = dist_.shard_rows(27 * 256, r.world, r.rank, multiple=256)
    # the one collective of the token-parallel prefill: all-gather of every rank's k|v rows, then
    # scatter into absolute order (bitnet-rs_amd/prefill_parallel.py), here with position markers
    import numpy as np
    import torch
    import torch.distributed as dist

    tp = importlib.import_module("bitnet-rs_amd.prefill_parallel")
    T = 512
    plan = tp.zigzag_plan(T, r.world)
    mine = tp.local_positions(plan[r.rank])
    send = torch.from_numpy(np.stack([mine, mine * 2 + 1], axis=1).astype(np.float32))
    recv = torch.empty(r.world * len(mine), 2)
    if r.world > 1:
        dist.all_gather_into_tensor(recv, send)
    else:
        recv.copy_(send)
    kv_all = torch.full((T, 2), -1.0)
    tp.scatter_gathered(recv.view(r.world, len(mine), 2), plan, kv_all)
    gather_ok = bool(torch.equal(kv_all[:, 0], torch.arange(T, dtype=torch.float32)) and torch.equal(kv_all[:, 1], torch.arange(T, dtype=torch.float32) * 2 + 1))
    res = {"rank": r.rank, "world": r.world, "gather_ok": gather_ok, "elapsed": elapsed, "value": value, "rows": [lo, hi], "rows256": [lo256, hi256],
           "local_sleep": sleep_s}
    with open(os.path.join(out_dir, f"rank{r.rank}.json"), "w") as f:
        json.dump(res, f)
    dist_.finalize(r)


if __name__ == "__main__":
    main()
