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
    res = {"rank": r.rank, "world": r.world, "elapsed": elapsed, "value": value, "rows": [lo, hi], "rows256": [lo256, hi256],
           "local_sleep": sleep_s}
    with open(os.path.join(out_dir, f"rank{r.rank}.json"), "w") as f:
        json.dump(res, f)
    dist_.finalize(r)


if __name__ == "__main__":
    main()
