"""Token-parallel prefill of one long prompt over the GPUs of a node (BASELINE
configs[4]): one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Partition.  Rows of the activation matrix = prompt tokens.  The prompt is cut into 2*world equal
chunks (multiples of 64 tokens); rank r owns chunks r and 2*world-1-r ("zigzag"), so every rank
does the same amount of causal attention work (a contiguous split would give the last rank
(2*world-1)x the first rank's).  Weights are replicated (521 MB): every projection and the FFN
are collective-free.  The ONE exchange per layer is an all-gather of the raw k|v rows
([T, 2*kv_dim] f32, 42 MB at T = 8192 for the 2B-4T shapes): attention needs the keys/values of
all earlier tokens.  After the gather every rank applies RoPE to all keys and fills its own
KV cache for all T positions (bitnet_hip_attention_prefill_sharded_dev), so any rank can go on
decoding.  No all-reduce: nothing on this path is a partial sum.

The layer loop itself lives in C++ (Decoder::prefill_sharded, bitnet-rs_amd/host/decoder.cpp; the collective is
ncclAllGather issued from that loop, bitnet_host_rccl_allgather).  This module keeps only what Python callers need: the
partition plan (tests, CPU rehearsals over gloo) and a torch.distributed carrier for the loop's all-gather callback.
"""
from __future__ import annotations

import numpy as np

BLOCK = 64  # attention query / key block (kernels_prefill_attn.hip)


def zigzag_plan(n_tokens: int, world: int):
    """-> per rank [(start, length), (start, length)]: chunks r and 2*world-1-r of 2*world equal chunks."""
    if world < 1 or n_tokens <= 0 or n_tokens % (2 * world * BLOCK) != 0:
        raise ValueError(f"prompt length {n_tokens} must be a positive multiple of {2 * world * BLOCK} (2 * world * {BLOCK}) for world {world}")
    c = n_tokens // (2 * world)
    return [[(r * c, c), ((2 * world - 1 - r) * c, c)] for r in range(world)]


def local_positions(plan_r) -> np.ndarray:
    return np.concatenate([np.arange(s, s + n) for s, n in plan_r]).astype(np.int64)


def scatter_gathered(gathered, plan, out) -> None:
    """gathered[r] holds rank r's rows in its local order; out[position] receives them (absolute order).
    Works on torch tensors or numpy arrays."""
    for r, chunks in enumerate(plan):
        row = 0
        for start, n in chunks:
            out[start:start + n] = gathered[r][row:row + n]
            row += n


def zz_row(position: int, n_tokens: int, world: int) -> int:
    """Row of absolute position `position` in the gathered (rank-major, zigzag) k|v buffer: the map the attention kernel applies
    where it reads the rows (kernels_prefill_attn.hip zz_row) -- chunk c = position / chunk belongs to rank c (first half of the
    chunks) or 2 world - 1 - c (second half), as that rank's first or second chunk."""
    chunk = n_tokens // (2 * world)
    c, off = divmod(position, chunk)
    first = c < world
    r = c if first else 2 * world - 1 - c
    return r * 2 * chunk + (0 if first else chunk) + off


# ---- assembling the `prefill_c5` object of bench.py's line (every rank calls these: they hold collectives) -----------------

PHASE_KEYS = ("matmul_us", "attention_us", "gather_wait_us", "gather_us")
PHASE_NOTE = ("medians over the layers, slowest rank: matmul = q|k|v + pack + o + gate|up + down, attention = query-side phase + k/v phase, "
              "gather_wait = what the compute stream waited for the collective beyond the query-side phase, gather = the collective on its own stream")


def count_ranks(world: int, device: str = "cpu") -> int:
    """Ranks that reached this point (an all-reduce of ones): `ranks_seen` of the line."""
    if world <= 1:
        return 1
    import torch
    import torch.distributed as dist

    t = torch.ones(1, device=device)
    dist.all_reduce(t)
    return int(t.item())


def assemble_phases(rank: int, world: int, ph: dict, device: str = "cpu") -> dict:
    """Every rank's per-layer phase medians (Decoder::phase_times) -> the object rank 0 prints: the SLOWEST rank's figures (the
    one whose compute phases add up to the most), its rank, and every rank's sum -- whichever communicator carried the gather."""
    mine = [float(ph[k]) for k in PHASE_KEYS]
    if world <= 1:
        return dict({k: round(v, 1) for k, v in zip(PHASE_KEYS, mine)}, rank=rank, per_layer_us=round(sum(mine[:3]), 1), note=PHASE_NOTE)
    import torch
    import torch.distributed as dist

    allp = torch.empty(world * 4, device=device)
    dist.all_gather_into_tensor(allp, torch.tensor(mine, device=device))
    allp = allp.cpu().reshape(world, 4).numpy()
    sums = allp[:, :3].sum(axis=1)
    slow = int(np.argmax(sums))
    out = {k: round(float(allp[slow, i]), 1) for i, k in enumerate(PHASE_KEYS)}
    out.update(rank=slow, per_layer_us=round(float(sums[slow]), 1), per_layer_us_by_rank=[round(float(x), 1) for x in sums], note=PHASE_NOTE)
    return out


def c5_line(world: int, prompt_len: int, steps: int, elapsed: float, cfg, digits: int, seen: int, token: int, check, phases: dict, how: str, rccl_version=None) -> dict:
    """The `prefill_c5` object (bench.py: rank 0).  Pure assembly: tested on CPU at world 8 (tests/test_dist_gloo.py)."""
    flops = 2.0 * 2_084_044_800 * (cfg.n_layers / 30) * prompt_len + 4.0 * prompt_len * prompt_len / 2 * cfg.n_heads * cfg.head_dim * cfg.n_layers
    kv_bytes = prompt_len * 2 * cfg.n_kv_heads * cfg.head_dim * 2
    return {
        "workload": f"bitnet-b1.58-2B-4T I2_S QK256 blocks, {world}xMI355X token-parallel prefill, {prompt_len}-token prompt",
        "tokens": prompt_len, "steps": steps, "ms_per_prompt": round(elapsed / steps * 1e3, 3), "tokens_per_s": round(prompt_len * steps / elapsed, 1),
        "eff_TFLOPs": round(flops * steps / elapsed / 1e12, 1), "digits": digits, "ranks_seen": seen, "first_sampled_token": token,
        "prefill_check": check, "phases": phases, "rccl_version": rccl_version,
        "parallelism": f"token-parallel x{world} (zigzag chunks), weights replicated",
        "collective": f"all-gather of k|v rows (f16 on the wire) per layer: {kv_bytes} B x {cfg.n_layers} layers; {how}" if world > 1 else how,
        "scaling": "strong",
    }


# ---- the collective for Decoder::prefill_sharded (C++ host loop, bitnet-rs_amd/host/decoder.cpp) ---------------------------

class _DevBytes:
    """__cuda_array_interface__ view of raw device memory, so torch can wrap the host loop's buffers."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def torch_gather(world: int):
    """gather(send_ptr, recv_ptr, bytes_per_rank, stream) over torch.distributed (backend nccl = RCCL, or gloo in the
    one-GPU rehearsals): the device buffers are the C++ loop's own.  Host-synchronises around the collective (torch's
    communicator streams are not the decoder's); the pure-C path (rccl.py + bitnet_host_rccl_allgather) has no such stop."""
    import torch
    import torch.distributed as dist

    def gather(send, recv, nbytes, stream):
        torch.cuda.synchronize()  # the pack kernel on the decoder's stream
        s = torch.as_tensor(_DevBytes(send, nbytes), device="cuda")
        r = torch.as_tensor(_DevBytes(recv, nbytes * world), device="cuda")
        if dist.get_backend() == "gloo":  # host-staged
            rc = torch.empty(nbytes * world, dtype=torch.uint8)
            dist.all_gather_into_tensor(rc, s.cpu())
            r.copy_(rc)
        else:
            dist.all_gather_into_tensor(r, s)
        torch.cuda.synchronize()
        return 0

    return gather
