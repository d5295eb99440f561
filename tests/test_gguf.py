"""GGUF I2_S ingestion: the CPU oracle's restatement against the
reference's own known answers and SHA-256-pinned fixtures, and the product C++ reader
(bitnet-rs_amd/host/gguf.cpp, through its C shim) against the oracle.  No GPU needed."""
import hashlib
import importlib
import os
import struct

import numpy as np
import pytest

from oracle import gguf_oracle as G
from tests import gguf_util as W
from tests.golden import make_gguf_fixtures as fx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# (nelems, available, has_sibling) -> flavour | None (= fail closed).
# crates/bitnet-models/tests/i2s_flavor_detection.rs and i2s_close_match_priority.rs
FLAVOR_KATS = [
    (64, 20, False, G.BITNET32F16), (64, 16, True, G.SPLIT32), (160, 40, False, G.SPLIT32), (512, 128, False, G.QK256),
    (64, 24, False, G.BITNET32F16), (2048, 640, False, G.BITNET32F16), (2080, 520, True, G.SPLIT32), (100, 40, False, G.BITNET32F16),
    (96, 50, False, None), (32, 10, False, G.BITNET32F16), (800, 256, False, G.QK256), (64, 28, False, G.BITNET32F16),
    (96, 39, False, None), (512, 128, True, G.QK256), (520, 190, False, G.QK256), (264, 70, True, G.SPLIT32),
    (264, 130, True, G.QK256), (10240, 2560, False, G.QK256), (10240, 2560, True, G.QK256), (32, 64, False, G.QK256),
]


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("bitnet-rs_amd")


@pytest.fixture(scope="module")
def reader(pkg):
    pkg.build()
    return pkg.GgufFile()  # no file: only the stateless helpers


@pytest.mark.parametrize("nelems,avail,sib,want", FLAVOR_KATS)
def test_flavor_kats_oracle_and_product(reader, nelems, avail, sib, want):
    if want is None:
        with pytest.raises(G.GgufError, match="no valid flavor detected") as e:
            G.detect_i2s_flavor(avail, nelems, sib)
        assert f"available: {avail}" in str(e.value) and "split_need" in str(e.value) and "inline_need" in str(e.value)
        with pytest.raises(Exception, match="no valid flavor detected") as e2:
            reader.detect_i2s_flavor(avail, nelems, sib)
        assert f"available: {avail}" in str(e2.value)
    else:
        assert G.detect_i2s_flavor(avail, nelems, sib) == want
        assert reader.detect_i2s_flavor(avail, nelems, sib) == want


def test_tolerance_and_orientation_kats(reader):
    # crates/bitnet-quantization/src/lib.rs:80-84
    assert [G.qk256_tolerance_bytes(n) for n in (1_000_000, 131_072, 100_000, 1_000, 20)] == [1000, 132, 100, 8, 8]
    # qk256_utils.rs:106-179
    assert G.expected_qk256_shape("model.layers.0.self_attn.k_proj.weight", 2560, 20, 5, 6912) == (640, 2560)
    assert G.expected_qk256_shape("model.layers.0.mlp.down_proj.weight", 2560, 20, 5, 6912) == (2560, 6912)
    assert G.expected_qk256_shape("model.layers.0.mlp.gate_proj.weight", 2560, 20, 5, 6912) == (6912, 2560)
    assert G.expected_qk256_shape("model.embed_tokens.weight", 2560, 20, 5, 6912) is None
    assert G.detect_qk256_orientation_by_bytes((2560, 2560), (2560, 2560), 1_638_400) == (2560, 2560)
    assert G.detect_qk256_orientation_by_bytes((640, 2560), (2560, 640), 409_600) == (640, 2560)
    assert G.detect_qk256_orientation_by_bytes((640, 2560), (2560, 640), 491_520) == (2560, 640)
    # strict mode: 8-byte window (types.rs:885-893)
    assert G.detect_i2s_flavor(2560 + 8, 10240, False, strict=True) == G.QK256
    for strict in (False, True):
        assert reader.detect_i2s_flavor(2560 + 8, 10240, False, strict) == G.detect_i2s_flavor(2560 + 8, 10240, False, strict)
    # random agreement sweep, product vs oracle
    rng = np.random.default_rng(5)
    for _ in range(2000):
        nelems = int(rng.integers(1, 5000))
        avail = int(rng.integers(0, nelems // 2 + 80))
        sib, strict = bool(rng.integers(2)), bool(rng.integers(2))
        try:
            want = G.detect_i2s_flavor(avail, nelems, sib, strict)
        except G.GgufError:
            want = None
        try:
            got = reader.detect_i2s_flavor(avail, nelems, sib, strict)
        except Exception:
            got = None
        assert got == want, (nelems, avail, sib, strict)
        shape = (int(rng.integers(1, 40)), int(rng.integers(1, 700)))
        assert reader.loader_is_qk256(shape, avail) == G.loader_is_qk256(shape, avail)


def test_reference_fixture_sha256_pins():
    """The generator reproduces the reference's committed checksums bit for bit
    (ci/fixtures/qk256/SHA256SUMS) -- so everything below runs on the reference's own files."""
    for name, gen in fx.FIXTURES.items():
        data = gen()
        sha, size = fx.EXPECTED[name]
        assert len(data) == size and hashlib.sha256(data).hexdigest() == sha
        with open(os.path.join(GOLD, name), "rb") as f:
            assert f.read() == data  # the committed copy is that file


FIXTURE_FACTS = {
    # name: (shape, size from offsets, loader says QK256?)   expectations of
    # crates/bitnet-models/tests/qk256_dual_flavor_tests.rs:79-190
    "qk256_4x256.gguf": ((4, 256), 256, True),
    "bitnet32_2x64.gguf": ((2, 64), 64, False),  # 40 payload bytes + 24 alignment bytes
    "qk256_3x300.gguf": ((3, 300), 384, True),
}


@pytest.mark.parametrize("name", sorted(FIXTURE_FACTS))
def test_fixtures_parse_oracle_and_product(pkg, name):
    shape, size, is_qk = FIXTURE_FACTS[name]
    data = open(os.path.join(GOLD, name), "rb").read()
    g = G.parse(data)
    assert g.version == 3 and g.alignment == 32 and g.data_start % 32 == 0
    t = g.tensors[0]
    assert (t.name, t.shape, t.ttype, t.offset, t.size) == ("tok_embeddings.weight", shape, G.I2_S, 0, size)
    assert g.tensors[1].name == "output.weight" and g.tensors[1].ttype == G.F16 and g.tensors[1].size == 2 * shape[0] * shape[1]
    cfg = G.extract_config(g)
    assert (cfg.vocab, cfg.hidden, cfg.n_layers, cfg.n_heads, cfg.n_kv, cfg.inter) == (1000, 512, 1, 8, 8, 2048)
    assert G.loader_is_qk256(t.shape, t.size) == is_qk
    kind = G.load_i2s(g, t, cfg)
    seed = {"qk256_4x256.gguf": 42, "bitnet32_2x64.gguf": 43, "qk256_3x300.gguf": 44}[name]
    code = seed % 4
    if is_qk:
        assert kind[0] == "qk256" and (kind[1], kind[2]) == shape
        assert len(kind[3]) == shape[0] * -(-shape[1] // 256) * 64 and set(kind[3]) == {code * 0x55}
    else:
        # 40 payload bytes + 24 alignment bytes = 64 = ceil(128/256)*64: pass 1 of the reference
        # loader takes it for QK256 and skips it, pass 2 (per-row count 128) rejects it
        assert kind == ("dropped",)
        # the block bytes themselves: 8 code bytes + f16 1.0, value = (code - 2) * scale
        raw = g.tensor_bytes(t)[:40]
        assert raw == (bytes([code * 0x55]) * 8 + b"\x00\x3c") * 4
    # product reader: same records
    f = pkg.GgufFile(data=data)
    ts = f.tensors()
    assert f.data_start == g.data_start
    assert [(x["name"], x["shape"], x["type"], x["offset"], x["size"]) for x in ts] == [(x.name, x.shape, x.ttype, x.offset, x.size) for x in g.tensors]
    c = f.config()
    assert (c["vocab"], c["hidden"], c["n_layers"], c["n_heads"], c["n_kv_heads"], c["ffn"]) == (1000, 512, 1, 8, 8, 2048)
    assert c["rope_theta"] is None and c["eps"] is None
    assert f.loader_is_qk256(shape, size) == is_qk
    f.close()
    # and from disk through mmap
    f2 = pkg.GgufFile(path=os.path.join(GOLD, name))
    assert len(f2.tensors()) == 2
    f2.close()


def test_header_variants_and_metadata(pkg):
    """Both v3 header forms the reference's reader accepts (types.rs:156-330), v2, value types,
    llama.* fallback keys, layer discovery from names."""
    t = [("blk.0.attn_q.weight", (4, 256), W.I2_S, bytes(256)), ("blk.3.ffn_norm.weight", (8,), W.F32, bytes(32))]
    kvs = [W.kv_str("general.name", "x"), W.kv_u32_array("some.array", [1, 2, 3]), W.kv_str_array("tokenizer.ggml.tokens", 7),
           W.kv_i32("llama.embedding_length", 64), W.kv_f32("llama.rope.freq_base", 500000.0), W.kv_u32("llama.attention.head_count", 4)]
    for std, ver, align in ((False, 3, 32), (True, 3, 64), (False, 2, 32)):
        data = W.write_gguf(kvs, t, alignment=align, version=ver, std_v3_header=std)
        g = G.parse(data)
        f = pkg.GgufFile(data=data)
        assert g.data_start == f.data_start and g.data_start % align == 0
        assert [(x.name, x.shape, x.offset, x.size) for x in g.tensors] == [(x["name"], x["shape"], x["offset"], x["size"]) for x in f.tensors()]
        c, oc = f.config(), G.extract_config(g)
        assert (c["vocab"], c["hidden"], c["n_layers"], c["n_heads"]) == (7, 64, 4, 4) == (oc.vocab, oc.hidden, oc.n_layers, oc.n_heads)
        assert c["rope_theta"] == 500000.0 == oc.rope_theta
        f.close()
    # the last tensor's size runs to the end of the file (reader.rs:167-169)
    g = G.parse(W.write_gguf(kvs, t) + bytes(100))
    assert g.tensors[1].size == 132


def test_malformed_files_fail_closed(pkg):
    good = W.write_gguf([W.kv_str_array("tokenizer.ggml.tokens", 3), W.kv_u32("llama.embedding_length", 8)],
                        [("a.weight", (2, 4), W.F32, bytes(32))])
    cases = {
        "magic": b"GGUX" + good[4:],
        "tiny": good[:12],
        "version": good[:4] + struct.pack("<I", 7) + good[8:],
        "tensor_count": good[:8] + struct.pack("<Q", 1 << 40) + good[16:],
        "kv_count": good[:16] + struct.pack("<Q", 1 << 40) + good[24:],
        "truncated_kv": good[:40],
    }
    for label, data in cases.items():
        with pytest.raises(G.GgufError):
            G.parse(data)
        with pytest.raises(Exception):
            pkg.GgufFile(data=data)
    # zero dimension, unknown tensor type, offset past the end
    for bad in (W.write_gguf([], [("z", (0, 4), W.F32, b"")]), W.write_gguf([], [("z", (2,), 99, bytes(8))])):
        with pytest.raises(G.GgufError):
            G.parse(bad)
        with pytest.raises(Exception):
            pkg.GgufFile(data=bad)
    # config: missing vocab / hidden (gguf_simple.rs:752-764)
    f = pkg.GgufFile(data=W.write_gguf([W.kv_u32("llama.embedding_length", 8)], []))
    with pytest.raises(Exception, match="vocab_size"):
        f.config()
    f.close()
    f = pkg.GgufFile(data=W.write_gguf([W.kv_str_array("tokenizer.ggml.tokens", 3)], []))
    with pytest.raises(Exception, match="hidden_size"):
        f.config()
    f.close()


def test_oracle_32_element_flavours_to_f32():
    """Inline f16 and split+sibling decode to (code-2)*scale (M/gguf_simple.rs:1128-1285).
    Sizes chosen so the reference's 128-byte slack separates the two (2 B x blocks > 160)
    and the whole-tensor QK256 size (first pass) is more than 0.1 % away."""
    rng = np.random.default_rng(3)
    rows, cols, nb = 41, 96, 123  # split 984 B, inline 1230 B, whole-tensor qk256 1024 B, per-row 2624 B
    codes = rng.integers(0, 256, (nb, 8), dtype=np.uint8)
    scales = rng.uniform(0.01, 2.0, nb).astype(np.float16)
    c4 = np.stack([(codes >> (2 * i)) & 3 for i in range(4)], axis=-1).reshape(nb, 32).astype(np.int32)
    want = ((c4 - 2).astype(np.float32) * scales.astype(np.float32)[:, None]).reshape(rows, cols)
    kv = [W.kv_str_array("tokenizer.ggml.tokens", 3), W.kv_u32("llama.embedding_length", 8)]
    cfg = G.Config(vocab=3, hidden=8)
    g = G.parse(W.write_gguf(kv, [("blk.0.attn_q.weight", (rows, cols), W.I2_S, W.inline_f16_blocks(codes, scales)),
                                  ("pad", (1,), W.F32, bytes(4))]))
    t = g.tensors[0]
    assert t.size == 1248 and not G.loader_is_qk256(t.shape, t.size) and not G.first_pass_is_qk256(rows * cols, t.size)
    kind = G.load_i2s(g, t, cfg)
    assert kind[0] == "f32" and np.array_equal(kind[1], want)
    for name, st, payload in (("blk.0.attn_q.scale", W.F32, scales.astype(np.float32).tobytes()),
                              ("blk.0.attn_q.scales", W.F16, scales.tobytes())):
        g = G.parse(W.write_gguf(kv, [("blk.0.attn_q.weight", (rows, cols), W.I2_S, codes.tobytes()), (name, (nb,), st, payload)]))
        assert g.tensors[0].size == 992
        kind = G.load_i2s(g, g.tensors[0], cfg)
        assert kind[0] == "f32" and np.array_equal(kind[1], want)
    # small tensors: alignment padding is inside the 128-byte slack of BOTH sizes; the reference tries
    # split first and then fails the inline length check (gguf_simple.rs:1181-1233)
    g = G.parse(W.write_gguf(kv, [("blk.0.attn_q.weight", (6, 96), W.I2_S, W.inline_f16_blocks(codes[:18], scales[:18])),
                                  ("pad", (1,), W.F32, bytes(4))]))
    with pytest.raises(G.GgufError, match="expected 180 bytes for inline f16, got 144"):
        G.i2s32_to_f32(g, g.tensors[0])


def test_truncated_last_tensor_is_refused_not_read_past_the_file(pkg):
    """The 32-element flavours accept a byte count within +-128 B of the expected one, but the loader then reads the EXPECTED
    count: for the file's last tensor that lies past the end of the file.  The reference refuses ("insufficient file data",
    gguf_simple.rs:1196-1207); a tensor in the middle of the file reads on into its successor, as the reference does."""
    rng = np.random.default_rng(3)
    rows, cols = 64, 256  # 512 blocks: split (4096 B) and inline (5120 B) sizes are further apart than the slack
    nb = rows * cols // 32
    payload = W.inline_f16_blocks(rng.integers(0, 256, (nb, 8), dtype=np.uint8), rng.uniform(0.5, 1.0, nb).astype(np.float16))
    assert len(payload) == nb * 10
    tail = ("pad.weight", (16,), W.F32, bytes(64))
    for short in (1, 100, 128):
        # last tensor, `short` bytes missing: inside the +-128 slack, outside the file
        f = pkg.GgufFile(data=W.write_gguf([], [tail, ("blk.0.attn_q.weight", (rows, cols), W.I2_S, payload[:-short])]))
        with pytest.raises(pkg.BitNetHipError, match="insufficient file data"):
            f.check_projection(1, rows, cols)
        f.close()
    # the same truncation in the MIDDLE of the file (the aligned successor's bytes are there): accepted
    f = pkg.GgufFile(data=W.write_gguf([], [("blk.0.attn_q.weight", (rows, cols), W.I2_S, payload[:-20]), ("pad.weight", (64,), W.F32, bytes(256))]))
    f.check_projection(0, rows, cols)
    f.close()
    # complete file: accepted; beyond the slack: the size mismatch message
    f = pkg.GgufFile(data=W.write_gguf([], [tail, ("blk.0.attn_q.weight", (rows, cols), W.I2_S, payload)]))
    f.check_projection(1, rows, cols)
    f.close()
    f = pkg.GgufFile(data=W.write_gguf([], [tail, ("blk.0.attn_q.weight", (rows, cols), W.I2_S, payload[:-129])]))
    with pytest.raises(pkg.BitNetHipError, match="don't match BitNet split"):
        f.check_projection(1, rows, cols)
    f.close()
    # QK256 last tensor cut short by less than the slack: I2SQk256NoScale::new would accept the size, gemv_qk256 not the data
    qrows, qcols = 4, 256
    f = pkg.GgufFile(data=W.write_gguf([], [tail, ("blk.0.attn_k.weight", (qrows, qcols), W.I2_S, bytes(qrows * 64))]))
    f.check_projection(1, qrows, qcols)
    f.close()


def test_llama_cpp_labelled_qk256_is_taken_in_the_configured_orientation(pkg):
    """k_proj of the Microsoft 2B file is labelled [2560, 640] (ne[0] = in first; the reference's tests/gqa_shapes.rs:22) while
    its bytes are 640 rows of 10 blocks.  Read as labelled (2560 rows x 3 blocks) the per-row byte count is wrong, and the
    reference's two passes leave such a tensor in neither map (restated in oracle/gguf_oracle.py).  The product loader takes
    it in the orientation the model configuration names -- the bytes are not moved -- and keeps refusing a size that fits
    neither orientation."""
    rows, cols = 128, 640  # out, in: 3 blocks per row; as labelled (640, 128) it would be 640 rows of 1 block
    good = bytes(rows * 3 * 64)
    f = pkg.GgufFile(data=W.write_gguf([], [("blk.0.attn_k.weight", (cols, rows), W.I2_S, good), ("pad.weight", (16,), W.F32, bytes(64))]))
    f.check_projection(0, rows, cols)
    f.close()
    f = pkg.GgufFile(data=W.write_gguf([], [("blk.0.attn_k.weight", (rows, cols), W.I2_S, good), ("pad.weight", (16,), W.F32, bytes(64))]))
    f.check_projection(0, rows, cols)  # labelled [out, in]: the reference's own convention
    f.close()
    # the oracle (= the reference's two passes) drops the llama.cpp-labelled tensor
    g = G.parse(W.write_gguf([], [("blk.0.attn_k.weight", (cols, rows), W.I2_S, good), ("pad.weight", (16,), W.F32, bytes(64))]))
    t = g.info("blk.0.attn_k.weight")
    assert not G.loader_is_qk256(t.shape, t.size)
    # bytes that fit neither orientation
    f = pkg.GgufFile(data=W.write_gguf([], [("blk.0.attn_k.weight", (cols, rows), W.I2_S, bytes(rows * 3 * 64 + 4096)), ("pad.weight", (16,), W.F32, bytes(64))]))
    with pytest.raises(pkg.BitNetHipError):
        f.check_projection(0, rows, cols)
    f.close()
