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


def test_eight_rank_gloo_plan_gather_order_and_c5_line(tmp_path):
    """BASELINE configs[4] rehearsed at its real world size on CPU: 8 ranks over gloo, 8192 positions.  The zigzag plan tiles the
    prompt and balances the causal work, the rank-major gathered buffer read through zz_row (the kernel's map,
    kernels_prefill_attn.hip) is the absolute order, ranks_seen = 8, the phases object names the slowest rank by its compute
    phases with every rank's sum beside it, and rank 0's prefill_c5 object is assembled (tensor_parallel.rs:123-290 is the
    reference's in-process counterpart)."""
    T, W = 8192, 8
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(W),
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_dist_c5_worker.py"), str(tmp_path), str(T),
    ]
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(W)]
    assert [r["world"] for r in res] == [W] * W and all(r["seen"] == W for r in res)
    assert all(r["order_ok"] for r in res)
    assert all(r["n_local"] == T // W for r in res)
    work = [r["causal_work"] for r in res]
    assert sum(work) == T * (T + 1) // 2
    assert max(work) - min(work) <= (T // (2 * W)) ** 2  # zigzag: equal up to the triangle inside a chunk (a contiguous split: 15x between first and last)
    for r in res:  # every rank computed the same object
        ph = r["phases"]
        assert ph["rank"] == 5 and ph["attention_us"] == 150.0 and ph["matmul_us"] == 405.0 and ph["per_layer_us"] == 558.0
        assert ph["per_layer_us_by_rank"] == [503.0 + i + (50.0 if i == 5 else 0.0) for i in range(W)]
    line = res[0]["line"]
    assert line["ranks_seen"] == W and line["rccl_version"] == "2.22.3" and line["phases"]["rank"] == 5 and line["scaling"] == "strong"
    assert line["tokens_per_s"] == round(T * 2 / 0.0246, 1) and line["ms_per_prompt"] == 12.3
    assert "8xMI355X token-parallel prefill, 8192-token prompt" in line["workload"] and "all-gather of k|v rows" in line["collective"]
    assert line["collective"].startswith(f"all-gather of k|v rows (f16 on the wire) per layer: {T * 2 * 5 * 128 * 2} B x 30 layers")


def test_single_rank_helpers():
    dist_ = importlib.import_module("bitnet-rs_amd.dist")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r = dist_.init()
    assert (r.world, r.rank) == (1, 0)
    assert dist_.max_over_ranks(r, 1.5) == 1.5
    assert dist_.aggregate_throughput(r, 10, 2.0) == 5.0
    tp = importlib.import_module("bitnet-rs_amd.prefill_parallel")
    assert tp.count_ranks(1) == 1
    ph = tp.assemble_phases(0, 1, {"matmul_us": 1.0, "attention_us": 2.0, "gather_wait_us": 0.5, "gather_us": 9.0})
    assert ph["per_layer_us"] == 3.5 and ph["rank"] == 0
    assert [tp.zz_row(p, 512, 2) for p in (0, 127, 128, 255, 256, 384, 511)] == [0, 127, 256, 383, 384, 128, 255]
    spans = [dist_.shard_rows(2560, 8, i, 16) for i in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 2560 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
