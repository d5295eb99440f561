"""GGUF reader under AddressSanitizer + UBSan (CPU build): mutated copies of the committed fixtures
must be rejected with an error or parsed consistently -- never fault.  tools/fuzz_gguf.{cpp,sh}."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_gguf_reader_survives_mutated_files():
    lib = os.path.join(ROOT, "bitnet-rs_amd", "libbitnet_hip.so")
    if not os.path.exists(lib):
        import importlib

        importlib.import_module("__graft_entry__").build()
    out = subprocess.run([os.path.join(ROOT, "tools", "fuzz_gguf.sh"), "30000"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "no faults" in out.stdout
