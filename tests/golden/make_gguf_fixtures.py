#!/usr/bin/env python3
"""Regenerates the three GGUF v3 fixtures the reference pins by SHA-256
(ci/fixtures/qk256/SHA256SUMS; the files themselves are not in the reference tree, only
their generator's format description, crates/bitnet-models/tests/helpers/qk256_fixtures.rs:62-273,
and their sizes, ci/fixtures/qk256/QUICK_REFERENCE.md:8-12).

Format written here (GGUF v3, little endian): magic, version 3, 2 tensors, 8 KVs
(general.name = "fixture_seed_<seed>", general.architecture = "bitnet", tokenizer.ggml.tokens =
1000 empty strings, bitnet-b1.58.{embedding_length=512, block_count=1, attention.head_count=8,
attention.head_count_kv=8, feed_forward_length=2048}), tensor infos for
tok_embeddings.weight (I2_S = type 36) and output.weight (F16 = type 1), both [rows, cols],
32-byte aligned data section, offsets relative to it.  The I2_S payload repeats one byte
(code = seed % 4 in all four 2-bit fields); BitNet32-F16 blocks are 8 code bytes + f16 1.0.

    python tests/golden/make_gguf_fixtures.py        # writes tests/golden/*.gguf, verifies SHA-256
"""
import hashlib
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ALIGN = 32
I2S, F16 = 36, 1

# the reference's committed checksums (ci/fixtures/qk256/SHA256SUMS) and sizes (QUICK_REFERENCE.md:8-12)
EXPECTED = {
    "bitnet32_2x64.gguf": ("c1568a0a08e38ef2865ce0816bfd2c617e5589c113114cd731e4c5014b7fbb20", 8832),
    "qk256_3x300.gguf": ("6e5a4f21607c0064affbcb86133627478eb34d812b59807a7123ff386c63bd3e", 10696),
    "qk256_4x256.gguf": ("a41cc62c893bcf1d4c03c30ed3da12da03c339847c4d564e9e5794b5d4c6932a", 10816),
}


def _s(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


def _kv_str(k, v):
    return _s(k.encode()) + struct.pack("<I", 8) + _s(v.encode())


def _kv_u32(k, v):
    return _s(k.encode()) + struct.pack("<II", 4, v)


def _kv_str_array(k, n):
    return _s(k.encode()) + struct.pack("<IIQ", 9, 8, n) + b"\0" * (8 * n)


def _pad(buf: bytearray):
    buf.extend(b"\0" * ((ALIGN - len(buf) % ALIGN) % ALIGN))


def build(rows, cols, payload: bytes, seed: int) -> bytes:
    buf = bytearray(b"GGUF" + struct.pack("<IQQ", 3, 2, 8))
    buf += _kv_str("general.name", f"fixture_seed_{seed}")
    buf += _kv_str("general.architecture", "bitnet")
    buf += _kv_str_array("tokenizer.ggml.tokens", 1000)
    buf += _kv_u32("bitnet-b1.58.embedding_length", 512)
    buf += _kv_u32("bitnet-b1.58.block_count", 1)
    buf += _kv_u32("bitnet-b1.58.attention.head_count", 8)
    buf += _kv_u32("bitnet-b1.58.attention.head_count_kv", 8)
    buf += _kv_u32("bitnet-b1.58.feed_forward_length", 2048)
    offs = []
    for name, typ in (("tok_embeddings.weight", I2S), ("output.weight", F16)):
        buf += _s(name.encode()) + struct.pack("<IQQI", 2, rows, cols, typ)
        offs.append(len(buf))
        buf += struct.pack("<Q", 0)
    _pad(buf)
    data_start = len(buf)
    buf[offs[0]:offs[0] + 8] = struct.pack("<Q", len(buf) - data_start)
    buf += payload
    _pad(buf)
    buf[offs[1]:offs[1] + 8] = struct.pack("<Q", len(buf) - data_start)
    buf += np.full(rows * cols, (seed % 256) / 256.0, dtype=np.float16).tobytes()
    return bytes(buf)


def qk256(rows, cols, seed):
    c = seed % 4
    byte = c | c << 2 | c << 4 | c << 6
    return build(rows, cols, bytes([byte]) * (rows * (-(-cols // 256)) * 64), seed)


def bitnet32(rows, cols, seed):
    c = seed % 4
    byte = c | c << 2 | c << 4 | c << 6
    block = bytes([byte]) * 8 + b"\x00\x3c"
    return build(rows, cols, block * (rows * (-(-cols // 32))), seed)


FIXTURES = {
    "qk256_4x256.gguf": lambda: qk256(4, 256, 42),
    "bitnet32_2x64.gguf": lambda: bitnet32(2, 64, 43),
    "qk256_3x300.gguf": lambda: qk256(3, 300, 44),
}


def main():
    for name, gen in FIXTURES.items():
        data = gen()
        sha, size = EXPECTED[name]
        got = hashlib.sha256(data).hexdigest()
        assert len(data) == size, (name, len(data), size)
        assert got == sha, (name, got, sha)
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(data)
        print(f"{name}: {len(data)} B sha256 {got} OK")


if __name__ == "__main__":
    main()
