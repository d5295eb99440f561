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
