"""Minimal GGUF v3 writer for the tests (data generation only): standard layout -- header,
KVs, tensor infos, `alignment`-aligned data section, offsets relative to it."""
import struct

import numpy as np

F32, F16, F64, I2_S = 0, 1, 4, 36


def _s(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


def kv_u32(k, v):
    return _s(k.encode()) + struct.pack("<II", 4, v)


def kv_i32(k, v):
    return _s(k.encode()) + struct.pack("<Ii", 5, v)


def kv_f32(k, v):
    return _s(k.encode()) + struct.pack("<If", 6, v)


def kv_str(k, v):
    return _s(k.encode()) + struct.pack("<I", 8) + _s(v.encode())


def kv_str_array(k, n):
    return _s(k.encode()) + struct.pack("<IIQ", 9, 8, n) + b"\0" * (8 * n)


def kv_u32_array(k, vals):
    return _s(k.encode()) + struct.pack("<IIQ", 9, 4, len(vals)) + b"".join(struct.pack("<I", v) for v in vals)


def write_gguf(kvs, tensors, alignment=32, version=3, std_v3_header=False) -> bytes:
    """tensors: list of (name, shape tuple, type id, payload bytes).  std_v3_header adds the
    (alignment u32, data_offset u64) fields the reference's reader also accepts after kv_count."""
    buf = bytearray(b"GGUF" + struct.pack("<IQQ", version, len(tensors), len(kvs)))
    hdr_fix = None
    if std_v3_header:
        buf += struct.pack("<I", alignment)
        hdr_fix = len(buf)
        buf += struct.pack("<Q", 0)
    for kv in kvs:
        buf += kv
    fix = []
    for name, shape, ttype, _ in tensors:
        buf += _s(name.encode()) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", d) for d in shape) + struct.pack("<I", ttype)
        fix.append(len(buf))
        buf += struct.pack("<Q", 0)
    buf += b"\0" * ((alignment - len(buf) % alignment) % alignment)
    data_start = len(buf)
    if hdr_fix is not None:
        buf[hdr_fix:hdr_fix + 8] = struct.pack("<Q", data_start)
    for (name, shape, ttype, payload), pos in zip(tensors, fix):
        buf += b"\0" * ((alignment - len(buf) % alignment) % alignment)
        buf[pos:pos + 8] = struct.pack("<Q", len(buf) - data_start)
        buf += payload
    return bytes(buf)


def model_kvs(cfg, prefix="bitnet-b1.58"):
    return [
        kv_str("general.architecture", "bitnet"),
        kv_str_array("tokenizer.ggml.tokens", cfg.vocab),
        kv_u32(f"{prefix}.embedding_length", cfg.hidden),
        kv_u32(f"{prefix}.block_count", cfg.n_layers),
        kv_u32(f"{prefix}.attention.head_count", cfg.n_heads),
        kv_u32(f"{prefix}.attention.head_count_kv", cfg.n_kv_heads),
        kv_u32(f"{prefix}.feed_forward_length", cfg.ffn),
        kv_f32(f"{prefix}.rope.freq_base", cfg.rope_theta),
        kv_f32(f"{prefix}.attention.layer_norm_rms_epsilon", cfg.eps),
    ]


BLK = {"q": "attn_q", "k": "attn_k", "v": "attn_v", "o": "attn_output", "gate": "ffn_gate", "up": "ffn_up", "down": "ffn_down"}


def inline_f16_blocks(codes: np.ndarray, scales_f16: np.ndarray) -> bytes:
    """codes u8 [blocks, 8], scales f16 [blocks] -> 10-byte blocks."""
    out = np.zeros((codes.shape[0], 10), np.uint8)
    out[:, :8] = codes
    out[:, 8:] = scales_f16.astype("<f2").view(np.uint8).reshape(-1, 2)
    return out.tobytes()
