"""Toolchain guard for the hand-scheduled prefill kernels (CPU only: hipcc cross-compiles gfx950 without a GPU).

k_prefill_attn stages its K / V^T tiles with LDS-DMA (`global_load_lds_dwordx4` from inline asm) and orders them with
hand-counted `s_waitcnt vmcnt(4)`: the count is right only while the COMPILER emits no vector-memory instruction and no
vmcnt wait of its own inside the tile loop (VERDICT r03 "what's weak" 8).  hipcc's own bookkeeping does not see the DMA
requests, so a stray load added to the loop -- by an edit or by a new compiler's scheduling -- would turn `vmcnt(4)` into a
race on the tiles without failing to build.  This test compiles the two prefill sources to ISA and asserts, for the
instantiations bench.py runs:
  * zero scratch, and the register ceilings that give two waves per SIMD (<= 256; the validated counts are 226 / 230 / 189);
  * in the attention tile loop: every global load / LDS-DMA and every `vmcnt` wait sits inside an inline-asm block
    (;;#ASMSTART .. ;;#ASMEND), none is compiler-emitted, no stores, no scratch traffic, and the per-iteration DMA count is
    the 8 pieces the `vmcnt(4)` arithmetic assumes (4 K pieces + 4 V^T pieces per wave and tile);
  * in the matmul's K loop: the expected number of loads, MFMAs and barriers per step, no stores.
It fails if someone adds a stray global load to the loop.  bitnet-rs_amd/build.py prints a warning when `hipcc --version`
differs from the version these counts were validated on (VALIDATED_HIPCC)."""
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bitnet-rs_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

ATTN = "_ZN10bitnet_hip14k_prefill_attnILi4ELi4ELi2EEEvNS_11PrefillArgsE"
GEMM64 = "_ZN10bitnet_hip11k_gemm_mfmaILi2ELi4ELi0ELi2ELi1EEEvNS_8GemmArgsE"
GEMM32 = "_ZN10bitnet_hip11k_gemm_mfmaILi2ELi2ELi0ELi2ELi1EEEvNS_8GemmArgsE"
GEMMF16 = "_ZN10bitnet_hip11k_gemm_f16aILi1ELi4ELi1ELi4EEEvNS_8GemmArgsEj"  # the f16 chain instance (BitNet32-F16, 64-token tile, chain epilogues)
# the 320-row workgroups of the 2560-row launches (five row tiles per wave), BitNet32-F16 and QK256 (the hybrid prompt forward's o / down)
GEMMF16_R5 = "_ZN10bitnet_hip11k_gemm_f16aILi1ELi4ELi1ELi5EEEvNS_8GemmArgsEj"
GEMMF16_R5_QK = "_ZN10bitnet_hip11k_gemm_f16aILi0ELi4ELi1ELi5EEEvNS_8GemmArgsEj"
GEMMFP6 = "_ZN10bitnet_hip10k_gemm_fp6ILi4ELi4ELi1ELi0EEEvNS_8GemmArgsEj"  # the fp6 x fp4 form on the resident fp4 image, 64-token tile (QK256 q|k|v, gate|up: round 5's default)
GEMMFP6_X = "_ZN10bitnet_hip10k_gemm_fp6ILi4ELi4ELi0ELi0EEEvNS_8GemmArgsEj"  # ... expanding the 2-bit tiles in its K loop (BITNET_HIP_FUSE_FP6_EXPAND)
GEMMFP6_QB = "_ZN10bitnet_hip10k_gemm_fp6ILi4ELi4ELi1ELi1EEEvNS_8GemmArgsEj"  # ... on QB32 rows with the chain epilogue (opt-in)
GEMMF16_QB = "_ZN10bitnet_hip11k_gemm_f16aILi0ELi4ELi2ELi5EEEvNS_8GemmArgsEj"  # the o- / down-projection handing QB32 rows over (opt-in)


def _compile(src, tmp):
    out = os.path.join(tmp, src.replace(".hip", ".s"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}", "-S", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-o", out]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    usage, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = usage.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return open(out).read().split("\n"), usage


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    tmp = str(tmp_path_factory.mktemp("isa"))
    return {src: _compile(src, tmp) for src in ("kernels_prefill_attn.hip", "kernels_gemm.hip")}


def body_of(lines, name):
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def classify(body):
    """-> list of (index, in_inline_asm, text) for every instruction line"""
    out, inasm = [], False
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            inasm = True
            continue
        if "#ASMEND" in l:
            inasm = False
            continue
        t = l.split(";")[0].strip()
        if t and not t.startswith("."):
            out.append((i, inasm, t))
    return out


def inner_loops(body):
    """(first, last) line index of every innermost loop: from an `Inner Loop Header` label to the branch back to it"""
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if m:
            back = [j for j in range(i, len(body)) if re.search(r"s_c?branch\S*\s+" + re.escape(m.group(1)) + r"\b", body[j])]
            assert back, "loop without a back edge: " + m.group(1)
            loops.append((i, back[-1]))
    return loops


VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)")


def test_resources_of_the_benchmarked_instantiations(isa):
    _, ua = isa["kernels_prefill_attn.hip"]
    _, ug = isa["kernels_gemm.hip"]
    for name, usage, ceiling in ((ATTN, ua, 230), (GEMM64, ug, 232), (GEMM32, ug, 168), (GEMMF16, ug, 200), (GEMMF16_R5, ug, 224), (GEMMF16_R5_QK, ug, 208),
                                 (GEMMFP6, ug, 244), (GEMMFP6_X, ug, 232), (GEMMFP6_QB, ug, 254), (GEMMF16_QB, ug, 254)):
        u = usage[name]
        assert u["ScratchSize"] == 0, (name, u)
        assert u["VGPRs"] + u.get("AGPRs", 0) <= ceiling, (name, u)
    assert ua[ATTN]["Occupancy"] >= 2 and ug[GEMM64]["Occupancy"] >= 2 and ug[GEMM32]["Occupancy"] >= 3 and ug[GEMMF16]["Occupancy"] >= 2
    assert ug[GEMMF16_R5]["Occupancy"] >= 2 and ug[GEMMF16_R5_QK]["Occupancy"] >= 2 and ug[GEMMFP6]["Occupancy"] >= 2


# the one instantiation hipcc 7.2 spills: 4 digits x 256-element block scales on the 8-wave tile (128 int32 + 64 f32 accumulators);
# reached only by digits = 4 on a 256-block-scaled matrix, never by bench.py or the decoder's default (2 digits)
# ... and k_gemm_f16h<1> (round 5: 64 x 128 wave tile of the BitNet32-F16 gate|up launch, 128 accumulators): 76 bytes, ONE scratch load per half-K step
# (128 MFMAs) in the loop, the rest in the prologue / epilogue -- bounded below so that it cannot grow unnoticed
GEMMF16H = "_ZN10bitnet_hip11k_gemm_f16hILi1EEEvNS_8GemmArgsEj"
KNOWN_SCRATCH = {"_ZN10bitnet_hip11k_gemm_mfmaILi4ELi2ELi1ELi1ELi2EEEvNS_8GemmArgsE", GEMMF16H}


def test_no_scratch_anywhere_else_in_the_prefill_sources(isa):
    for src, (lines, usage) in isa.items():
        for name, u in usage.items():
            assert u["ScratchSize"] == 0 or name in KNOWN_SCRATCH, (src, name, u)
    lines, usage = isa["kernels_gemm.hip"]
    assert usage[GEMMF16H]["ScratchSize"] <= 96 and usage[GEMMF16H]["Occupancy"] >= 2, usage[GEMMF16H]
    body = body_of(lines, GEMMF16H)
    for a, b in inner_loops(body):  # the K loop (both half steps unrolled): at most one scratch access per half step, 128 MFMAs each, no stores
        ins = [t for _, _, t in classify(body[a:b + 1])]
        if sum(t.startswith("v_mfma") for t in ins) >= 128:
            assert sum(t.startswith("scratch_") for t in ins) <= 2, [t for t in ins if t.startswith("scratch_")]
            assert not any(t.startswith(("global_store", "buffer_store")) for t in ins)
    for name in (ATTN, GEMM64, GEMM32, GEMMF16, GEMMF16_R5, GEMMF16_R5_QK, GEMMFP6):
        src = "kernels_prefill_attn.hip" if name == ATTN else "kernels_gemm.hip"
        assert not any(re.match(r"\s*scratch_", l) for l in body_of(isa[src][0], name)), name


def test_attention_tile_loop_has_only_the_kernels_own_memory_traffic(isa):
    lines, _ = isa["kernels_prefill_attn.hip"]
    body = body_of(lines, ATTN)
    loops = inner_loops(body)
    assert len(loops) == 1, loops
    lo, hi = loops[0]
    ins = [x for x in classify(body) if lo <= x[0] <= hi]
    dma = [x for x in ins if x[2].startswith("global_load_lds_dwordx4")]
    assert len(dma) == 8 and all(a for _, a, _ in dma), dma  # 4 K + 4 V^T pieces per tile, all from the kernel's asm
    stray = [x for x in ins if VMEM.match(x[2]) and not x[1]]
    assert not stray, f"compiler-emitted vector-memory instructions inside the LDS-DMA loop: {stray}"
    waits = [x for x in ins if "vmcnt" in x[2]]
    assert waits and all(a for _, a, _ in waits), f"a compiler-emitted vmcnt wait inside the LDS-DMA loop: {waits}"
    assert sorted(set(re.search(r"vmcnt\((\d+)\)", t).group(1) for _, _, t in waits)) == ["0", "4"], waits
    assert sum("s_barrier" in t for _, _, t in ins) == 2
    # the loop is entered with every compiler-tracked load retired (the builtin wait ahead of it)
    pre = [x for x in classify(body) if x[0] < lo]
    last_vm = max(i for i, _, t in pre if VMEM.match(t))
    assert any("vmcnt(0)" in t and not a for i, a, t in pre if i > last_vm), "no compiler-visible vmcnt(0) between the prologue's loads and the loop"


def test_a_stray_load_in_the_loop_is_caught(isa):
    """the detector itself: one global load pasted into the loop body must trip the check above"""
    lines, _ = isa["kernels_prefill_attn.hip"]
    body = list(body_of(lines, ATTN))
    lo, hi = inner_loops(body)[0]
    body.insert(lo + 5, "\tglobal_load_dword v1, v[2:3], off")
    ins = [x for x in classify(body) if lo <= x[0] <= hi + 1]
    assert [x for x in ins if VMEM.match(x[2]) and not x[1]]


@pytest.mark.parametrize("name,loads", [(GEMM64, 12), (GEMM32, 8)])
def test_matmul_k_loop_shape(isa, name, loads):
    lines, _ = isa["kernels_gemm.hip"]
    body = body_of(lines, name)
    loops = inner_loops(body)
    assert len(loops) == 1, loops
    lo, hi = loops[0]
    ins = [x for x in classify(body) if lo <= x[0] <= hi]
    assert sum(t.startswith("global_load_dwordx4") for _, _, t in ins) == loads  # 4 weight tiles + the thread's share of the activation tile
    assert not any(t.startswith(("global_store", "scratch_")) for _, _, t in ins)
    assert sum("s_barrier" in t for _, _, t in ins) == 1
    n_mfma = sum(t.startswith("v_mfma_i32_16x16x64_i8") for _, _, t in ins)
    assert n_mfma == (128 if name == GEMM64 else 64)


def test_build_warns_on_an_unvalidated_hipcc(monkeypatch, capsys):
    build = importlib.import_module("bitnet-rs_amd.build")
    assert build.hipcc_version() == build.VALIDATED_HIPCC, "new toolchain: re-run tests/test_isa_guard.py, then update VALIDATED_HIPCC"
    monkeypatch.setattr(build, "VALIDATED_HIPCC", "HIP version: 0.0")
    build.warn_if_unvalidated_hipcc()
    assert "hand-counted" in capsys.readouterr().err


GEMMFP6W = "_ZN10bitnet_hip11k_gemm_fp6wENS_8GemmArgsE"


def test_fp6w_k_loop_mixes_dma_and_tracked_loads_without_a_counted_wait(isa):
    """k_gemm_fp6w (round 5) stages its activation tile by LDS-DMA (invisible to hipcc's vmcnt bookkeeping) AND loads its weights by buffer loads hipcc
    does track: a counted `s_waitcnt vmcnt(n > 0)` that hipcc put in front of the weights' first use would, in hardware, also wait for DMA pieces or
    for part of the NEXT step's weights.  The kernel ends every step with the builtin vmcnt(0), so hipcc knows its loads have retired: the loop must
    hold that one wait and no other, 9 DMA pieces, 16 weight loads, 96 scaled MFMAs, one barrier, no scratch."""
    lines, usage = isa["kernels_gemm.hip"]
    u = usage[GEMMFP6W]
    assert u["ScratchSize"] == 0 and u["VGPRs"] + u.get("AGPRs", 0) <= 250 and u["Occupancy"] >= 2, u
    body = body_of(lines, GEMMFP6W)
    loops = [(a, b) for a, b in inner_loops(body) if sum(t.startswith("v_mfma_scale") for _, _, t in classify(body[a:b + 1])) >= 96]
    assert len(loops) == 1, loops
    lo, hi = loops[0]
    # hipcc rotates this loop: the block that requests the next tile sits BEFORE the header label and is reached by a branch from the loop's end
    for j in range(lo, hi + 1):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)\b", body[j])
        if m:
            at = next(i for i, l in enumerate(body) if l.startswith(m.group(1) + ":"))
            lo = min(lo, at)
    ins = classify(body[lo:hi + 1])
    assert sum(t.startswith("v_mfma_scale_f32_16x16x128_f8f6f4") for _, _, t in ins) == 96
    assert sum(t.startswith("global_load_lds_dwordx4") for _, _, t in ins) == 9
    assert sum(t.startswith("buffer_load_dwordx4") for _, _, t in ins) == 16
    assert sum("s_barrier" in t for _, _, t in ins) == 1
    assert not any(t.startswith(("scratch_", "global_store", "buffer_store")) for _, _, t in ins)
    waits = [t for _, _, t in ins if "vmcnt" in t]
    assert len(waits) == 1 and "vmcnt(0)" in waits[0], waits
