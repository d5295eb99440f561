"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, validates arguments like the reference does, and fails loudly (no
fallback) when there is no GPU.  No GPU compute here."""
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib(pkg):
    pkg.build()
    return pkg.load()


def test_library_exports_every_declared_symbol(pkg, lib):
    declared = pkg.declared_symbols()
    assert len(declared) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared if s not in exported]
    assert not missing, f"declared in include/bitnet_hip.h but not exported: {missing}"
    for s in declared:
        assert hasattr(lib.c, s)


def test_no_torch_or_oracle_linked(lib):
    """The boundary is plain C: no torch types, and the oracle is not linked in."""
    out = subprocess.check_output(["ldd", lib.path], text=True)
    assert "torch" not in out and "bitnet_oracle" not in out
    assert "amdhip64" in out


def test_argument_validation_matches_reference_wording(lib, pkg):
    """Q/i2s_qk256.rs:691-742 substrings; checks run before any device work."""
    z64, z128 = np.zeros(64, np.uint8), np.zeros(128, np.uint8)
    with pytest.raises(pkg.BitNetHipError, match="x length") as e:
        lib.gemv_qk256(z64, np.ones(246, np.float32), 1, 256, 64)
    assert e.value.code == pkg.ERR_INVALID_ARGUMENT and e.value.kind == "InvalidArguments"
    with pytest.raises(pkg.BitNetHipError, match="too short"):
        lib.gemv_qk256(z64, np.ones(256, np.float32), 2, 256, 64)
    with pytest.raises(pkg.BitNetHipError, match="y_out length"):
        lib.gemv_qk256(z128, np.ones(256, np.float32), 2, 256, 64, y_len=1)
    with pytest.raises(pkg.BitNetHipError, match="row bytes mismatch"):
        lib.gemv_qk256(z128, np.ones(256, np.float32), 1, 256, 128)
    # K/cpu/quantized_matmul.rs:705-742
    a4, p4, s4 = np.ones(4, np.float32), np.zeros(4, np.uint8), np.ones(4, np.float32)
    for m, n, k in [(0, 2, 2), (2, 0, 2), (2, 2, 0)]:
        with pytest.raises(pkg.BitNetHipError, match="dimensions must be > 0"):
            lib.i2s_matmul_f32(a4, p4, s4, m, n, k, 32, out_len=4)
    with pytest.raises(pkg.BitNetHipError, match="block_size must be > 0"):
        lib.i2s_matmul_f32(a4, p4[:2], s4[:2], 2, 2, 2, 0)
    with pytest.raises(pkg.BitNetHipError, match="activations too small"):
        lib.i2s_matmul_f32(a4[:2], p4, s4, 2, 2, 4, 32)
    with pytest.raises(pkg.BitNetHipError, match="output too small"):
        lib.i2s_matmul_f32(a4, p4[:2], s4[:2], 2, 2, 2, 32, out_len=1)
    # K/cpu/fallback.rs:320-331, :351-360
    with pytest.raises(pkg.BitNetHipError, match="dimension mismatch"):
        lib.matmul_i2s([1, 2], [1, 0], 2, 2, 2, c_len=4)
    with pytest.raises(pkg.BitNetHipError, match="too small"):
        lib.quantize(np.ones(32, np.float32), out_len=1)
    with pytest.raises(pkg.BitNetHipError, match="Invalid quantization type"):
        lib.quantize(np.ones(32, np.float32), qtype=7)
    with pytest.raises(pkg.BitNetHipError) as e:
        lib.quantize(np.ones(32, np.float32), qtype=pkg.QTYPE_TL1)
    assert e.value.kind == "UnsupportedHardware"
    # K/cuda/qk256_gemv.rs:60-68
    with pytest.raises(pkg.BitNetHipError, match="multiple of 256"):
        lib.qk256_gemv(np.zeros(64, np.uint8), np.ones(1, np.float32), np.ones(255, np.float32), 1, 1, 255)
    # Q/i2s_qk256.rs:499-541 size tolerance, surfaced at upload
    with pytest.raises(pkg.BitNetHipError, match="data size mismatch"):
        lib.weights_upload_qk256(np.zeros(512 * 256 + 129, np.uint8), 512, 1024, 256)
    with pytest.raises(pkg.BitNetHipError, match="unknown weights handle"):
        lib.weights_free(12345)


def test_fails_loudly_without_gpu(lib, pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert not lib.is_available() and lib.device_count() == 0
    with pytest.raises(pkg.BitNetHipError) as e:
        lib.init(0)
    assert e.value.kind == "GpuError"
    # a valid call must NOT silently compute on the CPU
    with pytest.raises(pkg.BitNetHipError) as e:
        lib.gemv_qk256(np.full(64, 0xAA, np.uint8), np.ones(256, np.float32), 1, 256, 64)
    assert e.value.kind == "GpuError"
    with pytest.raises(pkg.BitNetHipError):
        lib.matmul_i2s([1, 2, 3, 4], [1, 0, 0, 1], 2, 2, 2)


def test_missing_library_raises(pkg, tmp_path):
    with pytest.raises(FileNotFoundError, match="no fallback"):
        pkg.HipLib(str(tmp_path / "nope.so"))


def test_overflowing_dimensions_are_refused_before_any_multiplication(lib):
    """rows*stride etc. would wrap size_t: the entry points say so instead of computing with them."""
    import ctypes as C

    lib = C.CDLL(lib.path)
    lib.bitnet_hip_get_last_error.restype = C.c_char_p
    buf = (C.c_uint8 * 64)()
    fl = (C.c_float * 64)()
    h = C.c_uint64(0)
    big = C.c_size_t(1 << 62)
    lib.bitnet_hip_weights_upload_qk256.argtypes = [C.c_void_p, C.c_size_t] + [C.c_size_t] * 3 + [C.c_void_p]
    rc = lib.bitnet_hip_weights_upload_qk256(buf, 64, big, 256, 64, C.byref(h))
    assert rc != 0 and b"too large" in lib.bitnet_hip_get_last_error()
    lib.bitnet_hip_gemv_qk256.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t] + [C.c_size_t] * 3
    rc = lib.bitnet_hip_gemv_qk256(buf, 64, fl, 64, fl, 1 << 62, big, 4, 4)
    assert rc != 0 and b"too large" in lib.bitnet_hip_get_last_error()
    lib.bitnet_hip_matmul_i2s.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t] + [C.c_size_t] * 3
    rc = lib.bitnet_hip_matmul_i2s(buf, 0, buf, 0, fl, 0, 1 << 32, 1 << 32, 0)
    assert rc != 0 and b"too large" in lib.bitnet_hip_get_last_error()


def test_new_entry_points_validate_arguments_without_a_gpu(lib):
    import ctypes as C

    c = C.CDLL(lib.path)
    c.bitnet_hip_get_last_error.restype = C.c_char_p
    c.bitnet_hip_attention_merge_max_keys.restype = C.c_size_t
    assert c.bitnet_hip_attention_merge_max_keys() == 256
    c.bitnet_hip_gemv_attn_merge_dev.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert c.bitnet_hip_gemv_attn_merge_dev(12345, None, 8, 2, 512, None, None, None, None) != 0
    assert b"unknown weights handle" in c.bitnet_hip_get_last_error()
    c.bitnet_hip_hbm_read_ceiling.argtypes = [C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    best, mean = C.c_double(), C.c_double()
    assert c.bitnet_hip_hbm_read_ceiling(16, 1, C.byref(best), C.byref(mean), None) != 0  # below 1 MiB
    assert b"hbm_read_ceiling" in c.bitnet_hip_get_last_error()
    assert c.bitnet_hip_hbm_read_ceiling(1 << 30, 1, None, None, None) != 0
    c.bitnet_hip_attention_decode_partial_dev.argtypes = [C.c_void_p] * 5 + [C.c_size_t] * 4 + [C.c_void_p] * 3
    assert c.bitnet_hip_attention_decode_partial_dev(None, None, None, None, None, 8, 2, 128, 512, None, None, None) != 0
    assert b"Null pointer" in c.bitnet_hip_get_last_error()


def test_rejected_config_gives_a_dead_decoder_that_refuses_every_call(pkg):
    """a configuration the kernels cannot take (here hidden % 512 != 0, as untrusted GGUF metadata could say)
    used to leave an object whose layer table was empty while its layer count was not -- set_layer_* indexed past it.
    Now the object is dead: error() says why, every other entry returns an error, nothing is dereferenced."""
    import ctypes as C

    c = C.CDLL(pkg.HOST_LIB_PATH)
    c.bitnet_host_create.restype = C.c_void_p
    c.bitnet_host_create.argtypes = [C.POINTER(pkg.HostConfig)]
    c.bitnet_host_error.restype = C.c_char_p
    c.bitnet_host_error.argtypes = [C.c_void_p]
    hc = pkg.HostConfig(hidden=100, n_layers=4, n_heads=4, n_kv_heads=2, head_dim=128, ffn=256, vocab=64, max_pos=32, eps=1e-5, rope_theta=1e4)
    d = c.bitnet_host_create(C.byref(hc))
    assert d and b"multiple of 512" in c.bitnet_host_error(d)
    fl = (C.c_float * 128)()
    u8 = (C.c_uint8 * 128)()
    c.bitnet_host_set_layer_qk256.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 9
    for layer in (0, 3, -1, 1 << 20):
        assert c.bitnet_host_set_layer_qk256(d, layer, fl, fl, u8, u8, u8, u8, u8, u8, u8) != 0
    for name in ("bitnet_host_reset", "bitnet_host_position"):
        getattr(c, name).argtypes = [C.c_void_p]
        assert getattr(c, name)(d) != 0
    c.bitnet_host_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    assert c.bitnet_host_run(d, 1, 1, 0, None) != 0
    c.bitnet_host_layer_objects.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    hs, ps = (C.c_uint64 * 4)(1, 1, 1, 1), (C.c_void_p * 4)()
    c.bitnet_host_layer_objects(d, 0, hs, ps)
    assert list(hs) == [0, 0, 0, 0]
    c.bitnet_host_destroy.argtypes = [C.c_void_p]
    c.bitnet_host_destroy(d)
    # the Python wrapper raises and does not leak the object
    import importlib

    synth = importlib.import_module("bitnet-rs_amd.synth")
    with pytest.raises(pkg.BitNetHipError, match="multiple of 512"):
        pkg.HostDecoder(synth.ModelConfig(hidden=100, n_layers=1, n_heads=4, n_kv_heads=2, head_dim=128, ffn=256, vocab=64, max_pos=32))


def test_integration_md_declares_every_entry_point_for_rust():
    """INTEGRATION.md section 2 is the `extern "C"` block a reference maintainer pastes into crates/bitnet-kernels/src/rocm/sys.rs: every symbol the
    header declares must be spelt out there (round 4's review counted 44 of 70), with the argument count of the C prototype
    (tools/gen_rust_extern.py is the mechanical map both sides are held to)."""
    import importlib.util
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rust_extern", os.path.join(root, "tools", "gen_rust_extern.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = dict(gen.prototypes(open(os.path.join(root, "include", "bitnet_hip.h")).read()))
    assert len(want) >= 77
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    have = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (bitnet_hip_[a-z0-9_]+)\(([^;]*?)\)\s*(?:->[^;]*)?;", text, re.S)}
    missing = sorted(set(want) - set(have))
    assert not missing, missing

    def n_args(params: str) -> int:
        return 0 if not params.strip() else params.count(":")

    for name, line in want.items():
        assert n_args(have[name]) == n_args(line[line.index("(") + 1:line.rindex(")")]), name
