"""The N > 1 path of bench.py on CPU: world_size 2, backend gloo, 127.0.0.1 rendezvous.
Replicas only: no data-path collective, just the barrier-bracketed timing,
MAX over ranks, and whole-job aggregation the driver's contract asks for."""
import importlib
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_timing_and_aggregation(tmp_path):
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path),
    ]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [r["world"] for r in res] == [2, 2]
    # the all-gather + scatter of the token-parallel prefill puts every k|v row at its absolute position
    assert all(r["gather_ok"] for r in res)
    # every rank reports the SAME elapsed time = the slowest rank's (max over ranks)
    assert abs(res[0]["elapsed"] - res[1]["elapsed"]) < 1e-9
    assert res[0]["elapsed"] >= 0.1 - 1e-3  # rank 1 slept 0.1 s
    # whole-job value = world * steps / max time
    assert abs(res[0]["value"] - 2 * 10 / res[0]["elapsed"]) < 1e-6
    # row shards tile the range without overlap, on the requested multiples
    assert res[0]["rows"][0] == 0 and res[0]["rows"][1] == res[1]["rows"][0] and res[1]["rows"][1] == 6912
    assert all(v % 16 == 0 for r in res for v in r["rows"])
    assert res[0]["rows256"][1] == res[1]["rows256"][0] and res[1]["rows256"][1] == 27 * 256
    assert all(v % 256 == 0 for r in res for v in r["rows256"])


def test_single_rank_helpers():
    dist_ = importlib.import_module("bitnet-rs_amd.dist")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r = dist_.init()
    assert (r.world, r.rank) == (1, 0)
    assert dist_.max_over_ranks(r, 1.5) == 1.5
    assert dist_.aggregate_throughput(r, 10, 2.0) == 5.0
    spans = [dist_.shard_rows(2560, 8, i, 16) for i in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 2560 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
