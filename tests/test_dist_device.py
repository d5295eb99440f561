"""GPU: the N > 1 paths with two real processes on the one GPU of the box (torch.distributed backend gloo; RCCL needs one
GPU per rank, which the driver's 8-GPU node has and this box has not): the token-parallel prefill through the C++ host loop,
and bench.py --gpus 2 end to end (self-launch, replicas line + the prefill_c5 object in ONE json line)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_process_sharded_prefill_on_the_device(pkg, hip, tmp_path):
    T = 256
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tests", "_dist_prefill_worker.py"), str(tmp_path), str(T)]
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [r["world"] for r in res] == [2, 2] and all(r["pos"] == T for r in res)
    # the same prompt on one rank, in this process
    synth = importlib.import_module("bitnet-rs_amd.synth")
    cfg = synth.ModelConfig(hidden=512, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128, ffn=1024, vocab=2048, max_pos=640, eps=1e-5, rope_theta=10000.0)
    dec = pkg.HostDecoder(cfg)
    for l in range(cfg.n_layers):
        dec.set_layer_qk256(l, synth.make_layer(cfg, l))
    dec.set_globals(synth.make_globals(cfg))
    dec.reset()
    dec.feed(synth.prompt(T, cfg.vocab))
    dec.prefill(T, with_logits=True, digits=3)
    dec.run(2, with_logits=True)
    want = [int(t) for t in dec.history(T + 3)[T:]]
    want_logits = dec.last_logits()
    dec.close()
    assert res[0]["tokens"] == want          # rank 0 sampled the first token and decoded on
    assert res[1]["tokens"][1:] != [] and res[1]["pos"] == T
    got = np.load(tmp_path / "logits.npy")
    c = float(got.astype(np.float64) @ want_logits.astype(np.float64) / (np.linalg.norm(got) * np.linalg.norm(want_logits)))
    assert c >= 0.99999


def test_bench_two_ranks_one_line_with_prefill_c5(tmp_path):
    """python bench.py --gpus 2 without a launcher: starts its own two ranks, prints ONE line with the replicas value
    (n_gpus 2) and the token-parallel prefill object (ranks_seen 2)."""
    env = dict(os.environ, BITNET_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "8", "--warmup", "2", "--c5-prompt", "512",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "2xMI355X" in d["config"]["workload"] and d["config"]["parallelism"] == "replicas x2"
    c5 = d["prefill_c5"]
    assert c5["ranks_seen"] == 2 and c5["tokens"] == 512 and c5["tokens_per_s"] > 0 and 0 <= c5["first_sampled_token"] < 128256
    # a world size that disagrees with --gpus is refused instead of silently benchmarking one GPU
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stdout + p.stderr)


def test_rccl_single_rank_allgather_through_the_c_entry(pkg, hip):
    """The RCCL route of the token-parallel prefill with the one rank this box can host: a communicator created as
    bitnet-rs_amd/rccl.py (and a Rust host) does it, ncclAllGather issued by the C entry bitnet_host_rccl_allgather on a stream
    of ours -- the dlopen / dlsym resolution, the argument order and the datatype code are what this pins (with one rank the
    gathered buffer is the send buffer)."""
    import ctypes as C

    import torch

    rccl = importlib.import_module("bitnet-rs_amd.rccl")
    comm = rccl.Comm(0, 1)
    assert comm.handle != 0
    C.CDLL(pkg.LIB_PATH, mode=C.RTLD_GLOBAL)
    host = C.CDLL(pkg.HOST_LIB_PATH)
    host.bitnet_host_rccl_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    host.bitnet_host_rccl_allgather.restype = C.c_int
    n = 1 << 20
    send = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    recv = torch.zeros(n, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    rc = host.bitnet_host_rccl_allgather(C.c_void_p(comm.handle), C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), n, C.c_void_p(stream.cuda_stream))
    assert rc == 0
    stream.synchronize()
    assert torch.equal(send, recv)
    assert host.bitnet_host_rccl_allgather(None, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), n, None) != 0  # no communicator
    comm.close()
