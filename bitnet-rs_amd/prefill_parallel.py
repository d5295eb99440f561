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

This module is host plumbing only (buffers, launch order, the collective); every device
operation is a bitnet_hip_* call.  The collective is injected (`gather`) so that the
partition logic can be driven by virtual ranks on one GPU (tests) or gloo on CPU.
"""
from __future__ import annotations

import ctypes as C

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


class TokenParallelPrefill:
    """One rank's share of a sharded prompt forward.  Phases per layer: front (LayerNorm + q|k|v
    projection of the local rows -> the k|v rows to contribute), [all-gather], back (attention of the
    local queries against the whole context, o-proj, FFN)."""

    def __init__(self, dec, hip, rank: int, world: int, digits: int = 3):
        import torch

        self.torch = torch
        self.dec, self.hip, self.rank, self.world, self.digits = dec, hip, rank, world, digits
        cfg = dec.cfg
        self.H, self.D = cfg.hidden, cfg.head_dim
        self.QD, self.KD, self.F = cfg.n_heads * cfg.head_dim, cfg.n_kv_heads * cfg.head_dim, cfg.ffn
        L = dec.c
        L.bitnet_host_layer_objects.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        L.bitnet_host_layer_objects.restype = None
        L.bitnet_host_global_objects.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.bitnet_host_global_objects.restype = None
        L.bitnet_host_finish_prefill.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        self.layers = []
        for l in range(cfg.n_layers):
            h, p = (C.c_uint64 * 4)(), (C.c_void_p * 4)()
            L.bitnet_host_layer_objects(dec.h, l, h, p)
            self.layers.append(([int(v) for v in h], [int(v or 0) for v in p]))
        g = (C.c_void_p * 7)()
        L.bitnet_host_global_objects(dec.h, g)
        self.embed, self.final_norm, self.rope_sin, self.rope_cos = (int(g[i] or 0) for i in range(4))

    # -- phases ---------------------------------------------------------------------------
    def begin(self, tokens) -> None:
        """tokens: the WHOLE prompt (host int array); this rank embeds its own rows."""
        t = self.torch
        cfg = self.dec.cfg
        T = len(tokens)
        if T > cfg.max_pos - 1:
            raise ValueError("KV cache overflow")
        self.T = T
        self.plan = zigzag_plan(T, self.world)
        pos = local_positions(self.plan[self.rank])
        self.nq = nq = len(pos)
        dev = "cuda"
        self.block_pos = t.from_numpy(pos[::BLOCK].astype(np.int32)).to(dev)
        tok = t.from_numpy(np.asarray(tokens, np.int32)[pos]).to(dev)
        f32 = dict(dtype=t.float32, device=dev)
        self.x = t.empty(nq, self.H, **f32)
        self.qkv = t.empty(nq, self.QD + 2 * self.KD, **f32)
        self.att = t.empty(nq, self.QD, **f32)
        self.h = t.empty(nq, self.F, **f32)
        self.kv_send = t.empty(nq, 2 * self.KD, **f32)
        self.kv_all = t.empty(T, 2 * self.KD, **f32)
        self.gemm_wsb = self.hip.matmul_workspace_bytes(nq, max(self.H, self.F), self.digits)
        self.gemm_ws = t.empty(self.gemm_wsb, dtype=t.uint8, device=dev)
        self.attn_wsb = self.hip.attention_prefill_sharded_workspace_bytes(cfg.n_heads, cfg.n_kv_heads, nq, T)
        self.attn_ws = t.empty(self.attn_wsb, dtype=t.uint8, device=dev)
        self.stream = t.cuda.current_stream().cuda_stream
        self.hip.embed_f16_dev(self.embed, tok, self.x, nq, self.H, cfg.vocab, stream=self.stream)

    def _mm(self, handle, x, y, **kw):
        self.hip.matmul_fused_dev(handle, x, y, self.nq, self.gemm_ws, self.gemm_wsb, digits=self.digits, stream=self.stream, **kw)

    def layer_front(self, l: int):
        (qkv_h, _, _, _), (attn_norm, _, _, _) = self.layers[l]
        self._mm(qkv_h, self.x, self.qkv, ln_gamma=attn_norm, ln_eps=self.dec.cfg.eps)
        self.kv_send.copy_(self.qkv[:, self.QD:])  # the raw (pre-RoPE) k|v rows this rank contributes
        return self.kv_send

    def layer_back(self, l: int, gathered) -> None:
        cfg = self.dec.cfg
        (_, o_h, gu_h, down_h), (_, ffn_norm, kcache, vcache) = self.layers[l]
        scatter_gathered(gathered, self.plan, self.kv_all)
        self.hip.attention_prefill_sharded_dev(self.qkv, self.QD + 2 * self.KD, self.block_pos, self.nq, self.kv_all, 2 * self.KD, self.T,
                                               self.rope_sin, self.rope_cos, kcache, vcache, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim,
                                               cfg.max_pos, self.attn_ws, self.attn_wsb, self.att, stream=self.stream)
        self._mm(o_h, self.att, self.x, residual=self.x)
        self._mm(gu_h, self.x, self.h, ln_gamma=ffn_norm, ln_eps=cfg.eps, flags=1)
        self._mm(down_h, self.h, self.x, residual=self.x)

    def finish(self, with_logits: bool = True) -> None:
        """The last prompt position is the last local row of rank 0 (chunk 2*world-1)."""
        self.torch.cuda.synchronize()
        last = self.x[self.nq - 1].data_ptr() if self.rank == 0 else None
        rc = self.dec.c.bitnet_host_finish_prefill(self.dec.h, self.T, last, int(with_logits))
        self.dec._check(rc)

    # -- the real thing: one process per GPU -------------------------------------------------
    def run(self, tokens, with_logits: bool = True) -> None:
        import torch.distributed as dist

        t = self.torch
        self.begin(tokens)
        # concatenated layout [world * nq, cols] (the form both the RCCL and the gloo backend take)
        recv = t.empty(self.world * self.nq, 2 * self.KD, dtype=t.float32, device="cuda")
        for l in range(self.dec.cfg.n_layers):
            send = self.layer_front(l)
            if self.world > 1:
                dist.all_gather_into_tensor(recv, send)  # RCCL over xGMI; the only collective of the path
            else:
                recv.copy_(send)
            self.layer_back(l, recv.view(self.world, self.nq, 2 * self.KD))
        self.finish(with_logits)


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
