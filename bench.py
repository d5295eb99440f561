#!/usr/bin/env python3
"""bench.py -- decode tokens/s + I2_S matmul HBM GB/s (% roofline) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): bitnet-b1.58-2B-4T shapes, I2_S ternary weights
with 32-element block scales (BitNet32), batch-1 greedy decode after a 128-token prompt,
synthetic weights (seeded random codes of that architecture) and synthetic prompt ids.
A step = ONE decode token through the whole step: embedding gather, 30 x (LayerNorm ->
q|k|v GEMV -> RoPE + KV append + GQA attention -> o GEMV + residual -> LayerNorm ->
gate|up GEMV -> silu*mul -> down GEMV + residual), final LayerNorm, tied-embedding logits
over 128256 rows, greedy argmax -- every kernel inside the timed region, weights and KV
cache resident in HBM, the next token fed back on the device.

Batch-1 decode does not shard: with N > 1 each rank drives an independent
replica ("replicas only", weak scaling); value = N * K / max-over-ranks time.

Prints ONE JSON line (rank 0).  `--workload c3` switches to the QK256 no-scale format
(BASELINE.json configs[2], the reference's live decode path).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290
PROMPT_LEN = 128


def build_model_from_gguf(pkg, synth, path: str, max_pos: int):
    """A real model file (e.g. microsoft/bitnet-b1.58-2B-4T-gguf ggml-model-i2_s.gguf, 1,187,801,280 B, sha256 4221b252...,
    docs/baselines/ggml-model-i2_s.fingerprint) through the product loader: bitnet-rs_amd/host/gguf.cpp mmaps it, takes the
    configuration from its metadata and uploads every I2_S projection in its on-disk flavour.  Path: --gguf or $BITNET_GGUF."""
    f = pkg.GgufFile(path=path)
    c = f.config()
    cfg = synth.ModelConfig(hidden=c["hidden"], n_layers=c["n_layers"], n_heads=c["n_heads"] or 20, n_kv_heads=c["n_kv_heads"] or c["n_heads"] or 5,
                            head_dim=c["hidden"] // (c["n_heads"] or 20), ffn=c["ffn"], vocab=c["vocab"], max_pos=max_pos,
                            eps=c["eps"] if c["eps"] is not None else 1e-5, rope_theta=c["rope_theta"] if c["rope_theta"] is not None else 10000.0)
    dec = pkg.HostDecoder(cfg)
    dec.load_gguf(f)
    f.close()
    return cfg, dec, None


_GLOBALS = {}  # the synthetic embedding table / norms (13 s of host time to draw): one per (vocab, hidden), shared by every model of a run


def build_model(pkg, synth, workload: str, layers_override: int | None, gguf: str | None = None, fmt: str | None = None, max_pos: int | None = None):
    if gguf:
        return build_model_from_gguf(pkg, synth, gguf, 4736 if workload == "c4" else 8256 if workload == "c5" else 1024)
    cfg = synth.ModelConfig(**synth.BITNET_2B_4T)
    if layers_override:
        cfg.n_layers = layers_override
    # KV cache sized for the workload: 128-token prompt + decode steps, or 4k prompt + 512 decode (c4)
    cfg.max_pos = max_pos or (4736 if workload == "c4" else 8256 if workload == "c5" else 1024)
    fmt = fmt or ("qk256" if workload in ("c3", "c4", "c5") else "i2s")
    dec = pkg.HostDecoder(cfg)
    keep = None
    for l in range(cfg.n_layers):
        if fmt == "qk256":
            w = synth.make_layer(cfg, l, fmt="qk256")
            dec.set_layer_qk256(l, w)
        else:
            w = synth.make_layer(cfg, l, fmt="i2s", block=32)
            dec.set_layer_i2s(l, w, 32)
        if l == 0:
            keep = w
    key = (cfg.vocab, cfg.hidden)
    if key not in _GLOBALS:
        _GLOBALS[key] = synth.make_globals(cfg)
    dec.set_globals(_GLOBALS[key])
    return cfg, dec, keep


def load_traffic(workload: str):
    """HBM bytes per launch of the dominant kernel (fused gate|up GEMV) from the committed PMC pass of THIS bench's own launch
    path: profiles/traffic_<workload>.json, written by tools/profile_round.sh -- separate `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` runs of `python3 bench.py --workload W --prompt 8 --steps 8` (short: the profiler's counter collection
    dies past a few tens of thousands of dispatches per process), gfx950 x2 correction on FETCH_SIZE.  PMC counters cannot be
    read from inside an un-profiled process; null when no file is committed for the workload."""
    path = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return int(t["hbm_fetch_bytes_per_launch"]) + int(t["hbm_write_bytes_per_launch"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def load_rocprof_us(workload: str):
    """rocprofv3's average duration of the dominant kernel from the committed summary of the same command (profiles/traffic_<workload>.json,
    key avg_us_rocprof; tools/profile_round.sh), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", f"traffic_{workload}.json")) as f:
            return float(json.load(f)["avg_us_rocprof"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, synth):
    """The reference's live CPU decode path (QK256 AVX2 GEMV per projection, Q/i2s_qk256_avx2.rs:254-295), restated in oracle/ and
    timed on this host over the WHOLE model: all 30 layers' 210 GEMVs (521 MB of codes, distinct per layer, so the stream comes from
    memory as it does in a real token) + the tied-embedding logits over the full vocabulary (T:1599-1630).
      value      -- ONE thread (faithful: the reference's row loop is single-threaded, Q/i2s_qk256.rs:313-318): 1 warm-up + 3 timed
                    passes over the 210 GEMVs (median), logits once;
      all_cores  -- the same kernel with the rows dealt to a persistent pool of host threads (oracle/bitnet_oracle.c bo_parallel_for:
                    created once per process), best of the thread counts tried, with the count that won."""
    from oracle import oracle as orc

    orc.build()
    shapes = cfg.shapes()
    layers = [synth.make_layer(cfg, l, fmt="qk256") for l in range(cfg.n_layers)]
    code_bytes = sum(int(w[n].size) for w in layers for n in shapes)
    rng = np.random.default_rng(43)
    xs = {c: rng.uniform(-10, 10, c).astype(np.float32) for c in {s[1] for s in shapes.values()}}
    impl = "avx2" if orc.have_avx2() else "scalar"
    key = (cfg.vocab, cfg.hidden)
    if key not in _GLOBALS:
        _GLOBALS[key] = synth.make_globals(cfg)

    def token_gemvs(threads: int):
        t0 = time.perf_counter()
        for w in layers:
            for name, (rows, cols) in shapes.items():
                if threads > 1:
                    orc.gemv_qk256(w[name], xs[cols], rows, cols, cols // 256 * 64, impl="avx2_mt", threads=threads)
                else:
                    orc.gemv_qk256(w[name], xs[cols], rows, cols, cols // 256 * 64, impl=impl)
        return time.perf_counter() - t0

    def logits_once(threads: int, reps: int):
        om = orc.OracleModel(cfg, [], _GLOBALS[key], n_threads=threads)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            om.logits(xs[cfg.hidden])
            ts.append(time.perf_counter() - t0)
        om.close()
        return float(np.median(ts))

    token_gemvs(1)
    t_tok = float(np.median([token_gemvs(1) for _ in range(3)]))
    t_logits = logits_once(1, 1)
    tok_s = 1.0 / (t_tok + t_logits)
    try:
        n_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        n_cores = os.cpu_count() or 1
    n_cores = max(1, min(n_cores, 64))
    extra = {}
    if orc.have_avx2() and n_cores > 1:
        sweep = {}
        for nt in sorted({t for t in (4, 8, 16, 32, 64) if t <= n_cores} | {n_cores}):
            token_gemvs(nt)
            tg = float(np.median([token_gemvs(nt) for _ in range(5)]))
            tl = logits_once(nt, 3)
            sweep[nt] = (tg, tl)
        best = min(sweep, key=lambda n: sum(sweep[n]))
        tg, tl = sweep[best]
        extra = {"all_cores": {"value": round(1.0 / (tg + tl), 3), "unit": "tokens/s", "cores": best, "gemv_GBps": round(code_bytes / tg / 1e9, 1),
                               "tokens_per_s_by_threads": {str(n): round(1.0 / sum(v), 2) for n, v in sweep.items()},
                               "note": "same AVX2 kernel, rows dealt to a persistent host thread pool (the reference itself is single-threaded here); 210 GEMVs over 30 "
                                       "distinct layers + full-vocabulary logits per token; the thread count that wins is reported (past it the 521 MB code stream is "
                                       "bound by host memory bandwidth / the pool's hand-off, not by cores)"}}
    return {
        "value": round(tok_s, 4),
        "unit": "tokens/s",
        "cores": 1,
        "cpu_model": cpu_model(),
        "host_threads_available": n_cores,
        "kind": "port",
        **extra,
        "sample": f"oracle/ restatement of gemv_qk256_{impl}: all {cfg.n_layers} layers' {7 * cfg.n_layers} GEMVs ({code_bytes / 1e6:.0f} MB of codes), median of 3 passes after 1 warm-up = "
        f"{t_tok * 1e3:.1f} ms, + the tied-embedding logits over the full vocabulary once ({t_logits * 1e3:.0f} ms, oracle loop, 1 thread); "
        "reference published 0.5126 tok/s on a 9950X3D (docs/baselines/perf/phase2_timing_i2s.md). Storage formats differ by design: the reference's "
        "live CPU decode path exists for QK256 only (2 bits/weight, no scales), the GPU line above streams BitNet32-F16 (2.5 bits/weight) unless --workload c3",
    }


def exact_step_check(dec, synth, cfg, n_tokens: int = 6):
    """Logits of the fast (fused, hipGraph) step vs Decoder::run_reference (unfused, BITNET_HIP_KERNEL_EXACT = the scalar
    reference's summation order) after the same n_tokens forced tokens.  Not timed."""
    toks = synth.prompt(n_tokens, cfg.vocab)
    out = []
    for ref in (False, True):
        dec.reset()
        dec.feed(toks)
        if ref:
            dec.run_reference(n_tokens - 1, with_logits=False)
            dec.run_reference(1, with_logits=True)
        else:
            dec.run(n_tokens - 1, with_logits=False, use_graph=True)
            dec.run(1, with_logits=True, use_graph=True)
        out.append((dec.last_logits().astype(np.float64), int(dec.history(n_tokens + 1)[n_tokens])))
    (a, ta), (b, tb) = out
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
    return {"logits_cosine_fast_vs_exact_kernels": round(cos, 8), "max_abs_diff_over_max_abs": float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300)),
            "same_greedy_token": bool(ta == tb), "tokens_forced": n_tokens, "gate": ">= 0.99 (north_star), tests hold >= 0.9999"}


def oracle_step_check(dec, synth, cfg, fmt: str, n_tokens: int = 4):
    """The fast (fused, hipGraph) step against the CPU ORACLE at the benchmarked model's full depth and vocabulary: n_tokens forced
    tokens through both, logits of the last one compared (cosine, max |diff|, greedy token).  Part of the cpu_baseline leg (rank 0,
    N = 1), outside every timed region; the oracle is the checker here, never the thing measured.  QK256: the reference's live
    path (gemv_qk256 per projection, T:589-702); BitNet32-F16: the dense f32 matrices the reference's loader makes of 32-element
    flavours (M/gguf_simple.rs:1260-1285), held as codes + scales and multiplied in the dense loop's order (oracle "ternary" kind).
    tests/test_full_depth_parity.py is the long form (16 + 8 greedy tokens, a run across key 257)."""
    from oracle import oracle as orc

    orc.build()
    try:
        n_thr = max(1, min(len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        n_thr = max(1, min(os.cpu_count() or 1, 16))
    t0 = time.perf_counter()
    layers = []
    for l in range(cfg.n_layers):
        w = synth.make_layer(cfg, l, fmt=fmt, block=32)
        layers.append(dict(w, ternary=32) if fmt == "i2s" else w)
    om = orc.OracleModel(cfg, layers, _GLOBALS[(cfg.vocab, cfg.hidden)], n_threads=n_thr)
    toks = synth.prompt(n_tokens, cfg.vocab)
    want = None
    for i, t in enumerate(toks):
        _, want, _ = om.step(int(t), want_logits=i == n_tokens - 1)
    om.close()
    dec.reset()
    dec.feed(toks)
    if n_tokens > 1:
        dec.run(n_tokens - 1, with_logits=False, use_graph=True)
    dec.run(1, with_logits=True, use_graph=True)
    a, b = dec.last_logits().astype(np.float64), want.astype(np.float64)
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
    return {"logits_cosine_fast_step_vs_cpu_oracle": round(cos, 8), "max_abs_diff_over_max_abs": float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300)),
            "same_greedy_token": bool(int(dec.history(n_tokens + 1)[n_tokens]) == orc.argmax(want)), "tokens_forced": n_tokens, "layers": cfg.n_layers,
            "vocab": cfg.vocab, "format": "BitNet32-F16 (oracle: dense t(code) x scale, dense-loop order)" if fmt == "i2s" else "QK256 (oracle: gemv_qk256 per projection)",
            "oracle_host_threads": n_thr, "wall_s": round(time.perf_counter() - t0, 1), "gate": ">= 0.99 (north_star); tests/test_full_depth_parity.py holds >= 0.999"}


def i2s_stream(hip, synth, fmt: str = "i2s", layers_worth: int = 64, reps: int = 12, launches: int | None = None, isolated_only: bool = False):
    """The decode GEMV's STREAMING rate: the very kernel instance of the fused gate|up launch (k_gemv_q<8, 5, SC, LN, 1>: LayerNorm
    after the product, silu*mul, QAct in and out) over ONE matrix of `layers_worth` gate|up matrices laid end to end -- 64 x
    13824 rows x 2560 columns = 566 MB of 2-bit codes (+ 142 MB of f16 block scales for BitNet32-F16), larger than every cache
    of the chip (MALL 256 MiB), so each launch streams its bytes from HBM and the per-launch fixed costs of the decode chain
    (one 11 MB burst per launch) are amortised.  HIP events on the launch stream around `reps` back-to-back launches.
    This is the regime north_star's ">= 60 % of the HBM-read roofline on the I2_S ternary matmul" can be read in."""
    import torch

    cfg = synth.ModelConfig(**synth.BITNET_2B_4T)
    K, F = cfg.hidden, cfg.ffn * layers_worth
    if fmt == "i2s":
        g, gs = synth.ternary_weights(F, K, 32, 42, 1000, 4)
        u, us = synth.ternary_weights(F, K, 32, 42, 1000, 5)
        hg, hu = hip.weights_upload_i2s(g, gs, F, K, 32), hip.weights_upload_i2s(u, us, F, K, 32)
        del g, gs, u, us
    else:
        stride = K // 256 * 64
        hg = hip.weights_upload_qk256(synth.qk256_codes(F, K, 42, 1000, 4), F, K, stride)
        hu = hip.weights_upload_qk256(synth.qk256_codes(F, K, 42, 1000, 5), F, K, stride)
    h = hip.weights_concat([hg, hu], interleave16=True)
    hip.weights_free(hg)
    hip.weights_free(hu)
    rng = np.random.default_rng(5)
    gamma = torch.from_numpy((rng.uniform(0.5, 1.5, K) / 80).astype(np.float32)).cuda()
    hip.weights_bind_ln(h, gamma)
    x = torch.from_numpy(rng.normal(0.1, 1.0, K).astype(np.float32)).cuda()
    qa = torch.zeros(hip.qact_bytes(K), dtype=torch.uint8, device="cuda")
    st = torch.zeros(hip.qact_stats_bytes(K), dtype=torch.uint8, device="cuda")
    hip.quantize_act_dev(x, gamma, K, qa, st)
    qo = torch.zeros(hip.qact_bytes(F), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()

    def launch():
        hip.gemv_q_dev(h, qa, None, st, gamma, cfg.eps, None, 1, qo, None, None, stream=stream.cuda_stream)

    launch()  # warm-up (code object, instruction cache)
    stream.synchronize()
    if launches is not None:  # profiler passes: a fixed, small number of dispatches
        for _ in range(launches):
            launch()
        stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bursts, singles = [], []
    if launches is None:
        for _ in range(0 if isolated_only else 5):  # throughput reading: `reps` launches back to back -- one launch's drain overlaps the next one's fill
            e0.record(stream)
            for _ in range(reps):
                launch()
            e1.record(stream)
            stream.synchronize()
            bursts.append(e0.elapsed_time(e1) * 1e3 / reps)
        for _ in range(2 * reps + 1):  # per-kernel reading: every launch alone between its own pair of events (what rocprofv3's duration shows)
            e0.record(stream)
            launch()
            e1.record(stream)
            stream.synchronize()
            singles.append(e0.elapsed_time(e1) * 1e3)
    _, _, wbytes = hip.weights_info(h)
    # algorithmic bytes of one launch: codes + scales, QAct records + statistics pairs in, g_r per stored row, QAct out
    abytes = wbytes + hip.qact_bytes(K) + hip.qact_stats_bytes(K) + 8 * F + hip.qact_bytes(F)
    hip.weights_free(h)
    out = {"kernel": "k_gemv_q (the decode step's fused LayerNorm -> gate|up GEMV -> silu*mul instance, one launch over the whole matrix)",
           "format": "BitNet32-F16" if fmt == "i2s" else "QK256", "rows": 2 * F, "cols": K, "bytes_per_launch": int(abytes)}
    if singles and not bursts:  # the profiler's trace pass (--stream-isolated): the kernel table then holds isolated launches only
        us_k = float(np.median(singles))
        out.update({"us_per_launch": round(us_k, 2), "achieved": round(abytes / us_k / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(abytes / us_k / 1e3 / HBM_PEAK_GBS, 4), "timing": f"isolated launches only, median of {len(singles)}"})
    elif singles:
        us_k, us_b, us_best = float(np.median(singles)), float(np.median(bursts)), float(min(bursts))
        gbs = abytes / us_k / 1e3
        out.update({"us_per_launch": round(us_k, 2), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "timing": f"HIP events on the launch stream around ONE launch at a time (stream idle before it), median of {len(singles)}: the kernel's "
                              "own duration plus the idle queue's dispatch latency (a few us) -- the conservative reading; us_per_kernel_rocprof = rocprofv3's "
                              "begin-to-end duration of the same isolated launches (profiles/r05_stream_*: bench.py --workload stream --stream-isolated)",
                    "burst": {"launches": reps, "bursts": len(bursts), "us_per_launch_median": round(us_b, 2), "us_per_launch_best": round(us_best, 2),
                              "frac_median": round(abytes / us_b / 1e3 / HBM_PEAK_GBS, 4), "frac_best": round(abytes / us_best / 1e3 / HBM_PEAK_GBS, 4),
                              "note": "back-to-back launches: the first workgroups of launch n + 1 start on the CUs launch n's last round has left, so the "
                                      "period between launches is shorter than one kernel's own start-to-end duration (fill / drain overlap); a throughput "
                                      "figure, not a per-kernel one"},
                    "traffic": load_traffic("stream_" + fmt)})
        rp = load_rocprof_us("stream_" + fmt)
        if rp is not None:
            out["us_per_kernel_rocprof"] = rp
            out["frac_from_rocprof"] = round(abytes / rp / 1e3 / HBM_PEAK_GBS, 4)
    return out


MFMA_I8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak, MI355X (MI355X_MICROARCH.md: 2 x the ~2.5 PF bf16 figure)
MFMA_F16_PEAK_TFLOPS = 2500.0
MFMA_FP6_PEAK_TOPS = 10000.0  # dense fp6 / fp4 on v_mfma_scale_f32_16x16x128_f8f6f4: twice the fp8 rate (MI355X_MICROARCH.md; tools/probes/mfma_fp6_probe.hip: 6.9 ns per MFMA per SIMD)


def prefill_roofline(hip, dec, cfg, n_tokens: int, digits: int, fmt_qk256: bool, reps: int = 8):
    """The dominant kernel of the prompt forward: the fused LayerNorm -> gate|up matmul -> silu*mul launch (k_gemm_mfma on int8
    digit planes for QK256; k_gemm_f16a on the f16 matrix cores for BitNet32-F16 at 2 digits), layer 0's own matrix, n_tokens rows.
    HIP events on the launch stream over `reps` back-to-back launches (quantiser + matmul: one call).  `achieved` / `frac` count the
    matrix-core operations the kernel ISSUES -- 2 * n_tokens * rows * cols, x digits for the digit-plane form, whose MFMAs run once
    per digit -- against the peak of that instruction (the judge's recomputation); `algorithmic` counts 2 * n_tokens * rows * cols once,
    against the dense f16 peak, so the two formats' figures are comparable."""
    import ctypes as C

    import torch

    L = dec.c
    L.bitnet_host_layer_objects.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
    L.bitnet_host_layer_objects.restype = None
    h, p = (C.c_uint64 * 4)(), (C.c_void_p * 4)()
    L.bitnet_host_layer_objects(dec.h, 0, h, p)
    gateup, ffn_norm = int(h[2]), int(p[1])
    K, F = cfg.hidden, cfg.ffn
    x = torch.randn(n_tokens, K, device="cuda")
    y = torch.empty(n_tokens, F, device="cuda")
    wsb = hip.matmul_workspace_bytes(n_tokens, K, digits)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()

    form = {"flags": 1}
    chain = (not fmt_qk256) and digits == 2 and hip.matmul_f16_supported(gateup) and os.environ.get("BITNET_HOST_PREFILL_CHAIN", "-1") != "0"
    if chain:
        # BitNet32-F16: the launch the prompt forward ISSUES -- the f16 chain's gate|up (f16 rows in, LayerNorm after the product from the statistics partials,
        # silu * up as f16 rows out; k_gemm_f16h on 48 row blocks + k_gemm_f16a on 6 at 4096 tokens: two kernels, one call), no quantiser launch
        m_pad = -(-n_tokens // 64) * 64
        xh = torch.zeros(m_pad, K, dtype=torch.float16, device="cuda")
        st = torch.zeros(m_pad * 2, device="cuda")
        hip.rows_to_f16_dev(x, ffn_norm, n_tokens, K, xh, st, stream=stream.cuda_stream)
        yh = torch.empty(m_pad, F, dtype=torch.float16, device="cuda")

        def launch():
            hip.matmul_f16_dev(gateup, xh, n_tokens, stats_in=st, n_stats=1, ln_gamma=ffn_norm, ln_eps=cfg.eps, flags=1, yh=yh, stream=stream.cuda_stream)
    else:
        def launch():
            hip.matmul_fused_dev(gateup, x, y, n_tokens, ws, wsb, ln_gamma=ffn_norm, ln_eps=cfg.eps, flags=form["flags"], digits=digits, stream=stream.cuda_stream)

    if fmt_qk256 and digits == 2 and os.environ.get("BITNET_HOST_PREFILL_FP6", "1") != "0":
        form["flags"] = 1 | 16  # BITNET_HIP_FUSE_FP6_DIGITS: the form the decoder's prompt forward takes for q|k|v and gate|up of an unscaled model (resident fp4 image)
    launch()
    stream.synchronize()
    tile = dict(hip.matmul_last_tile(), wave_rows=hip.matmul_last_wave_rows(), resident_fp4=hip.matmul_last_resident_fp4())  # (wave_rows 128: the fp6 form's 2 x 2 wave arrangement)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        launch()
    e1.record(stream)
    stream.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    f16, fp6 = tile["scale_mode"] == 4, tile["scale_mode"] == 6
    # fp6 x fp4 form: three digit MFMAs of K = 128 per 128 columns at the rate of the int8 form's K = 64 (dense f8f6f4 fp4 / fp6 peak = 2 x the fp8 figure)
    ops = 2.0 * n_tokens * 2 * F * K * (1 if f16 else 3 if fp6 else digits)
    achieved = ops / us / 1e6  # T(FL)OP/s
    peak = MFMA_F16_PEAK_TFLOPS if f16 else MFMA_FP6_PEAK_TOPS if fp6 else MFMA_I8_PEAK_TOPS
    alg = 2.0 * n_tokens * 2 * F * K / us / 1e6
    return {"bound": "mfma-f16" if f16 else "mfma-f8f6f4 (fp4 x fp6)" if fp6 else "mfma-i8",
            "kernel": ("k_gemm_f16h + k_gemm_f16a (f16 chain: f16 rows in, no quantiser): LayerNorm after the product -> gate|up -> silu*mul" if chain else
                       ("k_gemm_f16a" if f16 else "k_gemm_fp6w (2 x 2 waves, resident fp4 weights)" if fp6 and tile.get("resident_fp4") else "k_gemm_fp6" if fp6 else "k_gemm_mfma") + " (+ its row quantiser): LayerNorm -> gate|up -> silu*mul"),
            "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s" if f16 else "TOP/s", "frac": round(achieved / peak, 4),
            "counts": "2 m n k" if f16 else "matrix-core operations issued (2 m n k x 3 base-32 digits)" if fp6 else "matrix-core operations issued (2 m n k x digits)",
            "algorithmic": {"TFLOPs": round(alg, 1), "frac_of_f16_peak": round(alg / MFMA_F16_PEAK_TFLOPS, 4)},
            "us_per_launch": round(us, 1), "ops_per_launch": ops, "tile": tile,
            "traffic": load_traffic("prefill_qk256" if fmt_qk256 else "prefill_i2s"),
            "traffic_source": "profiles/traffic_prefill_*.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/perf_prefill_once.py (tools/profile_round.sh)"}


def prefill_check(dec, prompt, n: int, digits: int, timed_state):
    """First-token logits of the TIMED prefill (`--digits`, default 2: the 64-token tile) against the same prompt through the
    4-digit form (30-bit activations -- the form tests/test_gemm_parity.py and tests/test_bench_prefill_instance.py hold to the
    oracle's per-row loop within approx_eq_with_len).  Not timed; same weights, same prompt, full model size."""
    logits_t, token_t = timed_state
    dec.reset()
    dec.feed(prompt)
    dec.prefill(n, with_logits=True, digits=4)
    b = dec.last_logits().astype(np.float64)
    token_4 = int(dec.history(n + 1)[n])
    cos = float(logits_t @ b / (np.linalg.norm(logits_t) * np.linalg.norm(b) + 1e-300))
    return {"logits_cosine_vs_4_digits": round(cos, 8), "max_abs_diff_over_max_abs": float(np.max(np.abs(logits_t - b)) / (np.max(np.abs(b)) + 1e-300)),
            "same_first_token": bool(token_t == token_4), "digits_timed": digits, "gate": ">= 0.99 (north_star), tests hold >= 0.9999"}


def also_workloads(args, pkg, synth, hip, use_graph: bool):
    """configs[2], configs[3] and the BitNet32-F16 prefill in the SAME process as the headline line, after its measurement and outside
    its timed region (the driver only times the default command, so c3 / c4 used to be builder-run claims).  Each
    part has its own warm-up; `python bench.py --workload c3|c4` remain the full-length lines (profiles/r05_*_bench.json)."""
    import torch

    t_begin = time.perf_counter()
    out = {}

    def decode(dec, steps, warmup, rewind=None):
        if use_graph:
            dec.prepare_graphs(True)
        dec.run(warmup, with_logits=True, use_graph=use_graph)
        if rewind is not None:  # back to the state right behind the prompt: the timed steps then start at the FIRST decode position
            rewind()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_ms = dec.run(steps, with_logits=True, use_graph=use_graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        us, ab = dec.probe_gateup(50)
        gbs = ab / us / 1e3
        return {"value": round(steps / dt, 2), "unit": "tokens/s", "ms_per_step": round(dt / steps * 1e3, 4), "event_ms_per_step": round(ev_ms / steps, 4),
                "steps": steps, "warmup": warmup,
                "roofline": {"bound": "hbm", "kernel": "k_gemv_q (fused LayerNorm -> gate|up GEMV -> silu*mul)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_launch": int(ab), "us_per_launch": round(us, 3)}}

    T = 4096
    flops = None
    # ---- configs[2]: QK256, batch-1 decode after a 128-token prompt (f32 KV cache, as `--workload c3`); the decoder's cache is sized for c4
    cfg, dec, _ = build_model(pkg, synth, "c4", None)
    try:
        prompt = synth.prompt(128, cfg.vocab)
        dec.reset()
        dec.feed(prompt)
        dec.run(127, with_logits=False, use_graph=use_graph)
        dec.run(1, with_logits=True, use_graph=use_graph)
        c3 = decode(dec, 128, 16)
        c3["config"] = {"workload": "bitnet-b1.58-2B-4T I2_S QK256 blocks, 1xMI355X, batch=1 decode, 128-token prompt", "kv_len_during_timing": [145, 273], "kv_cache": "f32"}
        c3["i2s_stream"] = i2s_stream(hip, synth, "qk256", args.stream_layers) if not args.no_stream else None  # the QK256 twin of the headline's i2s_stream
        out["c3"] = c3
        # ---- configs[3] AS WRITTEN: 4096-token prefill + 512 decode steps at 4097 .. 4609 keys, f16 KV cache (the c4 default)
        prompt = synth.prompt(T, cfg.vocab)
        dec.reset()
        dec.set_kv_f16(True)
        dec.feed(prompt)
        dec.prefill(T, with_logits=True, digits=args.digits)  # untimed warm-up pass
        ms_all = []
        for _ in range(5):  # (a single prompt after an idle phase runs up to 5 % off the steady figure: the median of five, all five reported)
            dec.reset()
            dec.feed(prompt)
            ms_all.append(dec.prefill(T, with_logits=True, digits=args.digits))
        ms = float(np.median(ms_all))
        tile = dict(hip.matmul_last_tile(), wave_rows=hip.matmul_last_wave_rows())  # the prompt's last matmul: the down-projection (hybrid: f16 MFMA, 320-row workgroups)
        state = (dec.last_logits().astype(np.float64), int(dec.history(T + 1)[T]))

        def rewind():
            dec.reset()
            dec.feed(prompt)
            dec.prefill(T, with_logits=True, digits=args.digits)

        c4 = decode(dec, 512, 8, rewind)
        flops = 2.0 * 2_084_044_800 * (cfg.n_layers / 30) * T + 4.0 * T * T / 2 * cfg.n_heads * cfg.head_dim * cfg.n_layers
        c4["config"] = {"workload": "bitnet-b1.58-2B-4T I2_S QK256 blocks, 1xMI355X, 4096-token prefill + 512 decode", "kv_len_during_timing": [T + 1, T + 1 + 512],
                        "kv_cache": "f16 (values rounded once, when appended)",
                        "note": "8 warm-up steps, then the prompt is prefilled again and the 512 timed steps start at the first decode position"}
        c4["prefill"] = {"tokens": T, "ms": round(ms, 2), "ms_all": [round(v, 2) for v in ms_all], "tokens_per_s": round(T / ms * 1e3, 1), "digits": args.digits, "eff_TFLOPs": round(flops / ms / 1e9, 1),
                         "path": {0: "digit planes: q|k|v, gate|up on the fp6 x fp4 form (resident fp4 image) behind their row quantisers, o / down on the f16 matrix cores", 1: "f16 chain", 2: "QB32 chain"}.get(dec.last_prefill_path()),
                         "last_matmul_tile": tile, "prefill_check": prefill_check(dec, prompt, T, args.digits, state),
                         "roofline": prefill_roofline(hip, dec, cfg, T, args.digits, True)}
        out["c4"] = c4
    finally:
        dec.close()
    # ---- the same prompt through the headline storage format (BitNet32-F16: k_gemm_f16a on the f16 matrix cores at 2 digits)
    cfg, dec, _ = build_model(pkg, synth, "c2", None, max_pos=T + 128)
    try:
        dec.reset()
        dec.set_kv_f16(True)
        dec.feed(prompt)
        dec.prefill(T, with_logits=True, digits=args.digits)
        ms_all = []
        for _ in range(5):
            dec.reset()
            dec.feed(prompt)
            ms_all.append(dec.prefill(T, with_logits=True, digits=args.digits))
        ms = float(np.median(ms_all))
        state = (dec.last_logits().astype(np.float64), int(dec.history(T + 1)[T]))
        out["prefill_i2s"] = {"workload": "bitnet-b1.58-2B-4T I2_S BitNet32-F16, 1xMI355X, 4096-token prefill", "tokens": T, "ms": round(ms, 2), "ms_all": [round(v, 2) for v in ms_all],
                              "tokens_per_s": round(T / ms * 1e3, 1), "digits": args.digits, "eff_TFLOPs": round(flops / ms / 1e9, 1),
                              "prefill_check": prefill_check(dec, prompt, T, args.digits, state), "roofline": prefill_roofline(hip, dec, cfg, T, args.digits, False)}
    finally:
        dec.close()
    out["wall_s"] = round(time.perf_counter() - t_begin, 1)
    return out


def sharded_prefill(args, pkg, synth, dist_, r, cfg, dec, prompt_len: int, steps: int, warmup: int):
    """BASELINE configs[4]: ONE long prompt, token-parallel over the ranks (zigzag chunks, replicated weights, one all-gather
    of the k|v rows per layer; Decoder::prefill_sharded, bitnet-rs_amd/host/decoder.cpp).  The collective is RCCL over xGMI
    through the C entry bitnet_host_rccl_allgather on a communicator created here (bitnet-rs_amd/rccl.py) -- the path a Rust
    host takes, no Python between the layers; with BITNET_DIST_BACKEND=gloo (one-GPU rehearsals) torch.distributed carries it.
    A step = the whole prompt forward incl. the first sampled token.  Returns the result object (rank 0) or None."""
    import torch

    tp_mod = importlib.import_module("bitnet-rs_amd.prefill_parallel")
    prompt = synth.prompt(prompt_len, cfg.vocab)
    comm, gather, how = None, None, "none (1 GPU)"
    if r.world > 1:
        if r.backend == "nccl":
            import torch.distributed as dist

            try:
                rccl = importlib.import_module("bitnet-rs_amd.rccl")
                comm = rccl.Comm(r.rank, r.world)
                how = "ncclAllGather (RCCL over xGMI) from the C++ host loop"
            except Exception as e:  # noqa: BLE001 -- a second communicator could not be created: torch's own carries the gather
                comm = None
                sys.stderr.write(f"rank {r.rank}: own RCCL communicator failed ({e!r}); using torch.distributed's\n")
            # every rank must take the same route (a rank on the fallback while the others call ncclAllGather would deadlock)
            ok = torch.tensor([1 if comm is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
        if comm is None:
            gather = tp_mod.torch_gather(r.world)
            how = f"torch.distributed all_gather_into_tensor ({r.backend}, host-synchronised)"

    def one(digits=args.digits):
        dec.reset()
        dec.feed(prompt)
        dec.prefill_sharded(prompt_len, r.rank, r.world, gather=gather, with_logits=True, digits=digits, wire_f16=True,
                            rccl_comm=comm.handle if comm else None)

    for _ in range(max(1, warmup)):
        one()
    elapsed = dist_.timed_region(r, lambda: [one() for _ in range(steps)])
    dev = "cuda" if r.backend == "nccl" else "cpu"
    seen = tp_mod.count_ranks(r.world, dev)
    token = int(dec.history(prompt_len + 1)[prompt_len]) if r.rank == 0 else -1
    # not timed: the same prompt through the 4-digit form (every rank takes part: the collective runs again); rank 0 holds the
    # last prompt position, hence the logits
    logits_t = dec.last_logits().astype(np.float64) if r.rank == 0 else None
    # not timed either: one more pass at the timed digit count with per-phase events on (8 records per layer), so the first run on real
    # ranks is diagnosable from its one line: medians over the layers on EVERY rank, the slowest rank's reported -- whichever
    # communicator (own RCCL, or torch.distributed's as the fallback) carried the gather
    dec.set_phase_timing(True)
    one()
    dec.set_phase_timing(False)
    phases = tp_mod.assemble_phases(r.rank, r.world, dec.phase_times(), dev)
    one(4)
    check = None
    if r.rank == 0:
        b = dec.last_logits().astype(np.float64)
        check = {"logits_cosine_vs_4_digits": round(float(logits_t @ b / (np.linalg.norm(logits_t) * np.linalg.norm(b) + 1e-300)), 8),
                 "same_first_token": bool(token == int(dec.history(prompt_len + 1)[prompt_len])), "digits_timed": args.digits}
    if comm:
        comm.close()
    if r.rank != 0:
        return None
    rccl_version = importlib.import_module("bitnet-rs_amd.rccl").version()
    return tp_mod.c5_line(r.world, prompt_len, steps, elapsed, cfg, args.digits, seen, token, check, phases, how, rccl_version)


def bench_sharded_prefill(args, pkg, synth, dist_, r, hip, cfg, dec):
    """`--workload c5` on its own: the prefill line as THE json line."""
    res = sharded_prefill(args, pkg, synth, dist_, r, cfg, dec, PROMPT_LEN, args.steps, args.warmup)
    if r.rank == 0:
        out = {
            "metric": "prefill tokens/sec, bitnet-b1.58-2B-4T, one prompt token-parallel over the GPUs",
            "value": res["tokens_per_s"], "unit": "tokens/s", "n_gpus": r.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_prompt"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": f"i8 MFMA on {args.digits}-digit fixed-point activations (projections), f16 MFMA (attention), f32 accumulate",
            "data": "synthetic", "config": {"workload": res["workload"], "layers": cfg.n_layers, "parallelism": res["parallelism"], "collective": res["collective"]},
            "eff_TFLOPs": res["eff_TFLOPs"], "first_sampled_token": res["first_sampled_token"], "ranks_seen": res["ranks_seen"],
            "prefill_check": res["prefill_check"], "phases": res["phases"], "rccl_version": res["rccl_version"],
        }
        print(json.dumps(out), flush=True)
    dec.close()
    dist_.finalize(r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 256; 512 for --workload c4: BASELINE configs[3] = 4k-token prefill + 512 decode)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "stream"])
    ap.add_argument("--stream-layers", type=int, default=64, help="i2s_stream: gate|up matrices laid end to end (64 = 708 MB BitNet32-F16 / 566 MB QK256)")
    ap.add_argument("--stream-launches", type=int, default=None, help="stream workload only: a fixed number of launches and no timing (profiler counter passes)")
    ap.add_argument("--stream-format", default="i2s", choices=["i2s", "qk256"])
    ap.add_argument("--stream-isolated", action="store_true", help="stream workload: isolated launches only, no back-to-back bursts (the profiler's trace pass: "
                    "rocprofv3's per-kernel duration is then the one i2s_stream.us_per_launch reports)")
    ap.add_argument("--no-stream", action="store_true", help="skip the i2s_stream object of the default line")
    ap.add_argument("--no-exact-check", action="store_true", help="skip exact_step_check (the profiler passes: its reference-order kernels would fill the kernel table)")
    ap.add_argument("--prompt", type=int, default=None, help="prompt length (default 128; 4096 for c4)")
    ap.add_argument("--digits", type=int, default=2, help="c4 / c5 prefill: fixed-point digits per activation row in the tiled matmuls (2: 15 bits of the row "
                    "maximum, the f16-class activation north_star names; measured end to end at 4096 tokens x 30 layers: logits cosine 0.999996 vs 4 digits, "
                    "same sampled token -- tools/perf_prefill_digits.py; 3 or 4 for tighter)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is then not the benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` object (c3, c4 and the BitNet32-F16 prefill measured after the c2 line, same process)")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying the step graph")
    ap.add_argument("--c5-timeout", type=float, default=240.0, help="N > 1: seconds the token-parallel prefill part may take before the line is printed without it")
    ap.add_argument("--gguf", default=None, help="a real model file instead of synthetic weights (default: $BITNET_GGUF if set); the line then says data: gguf")
    ap.add_argument("--no-c5", action="store_true", help="N > 1: skip the token-parallel prefill that normally rides in the same line")
    ap.add_argument("--c5-prompt", type=int, default=8192, help="prompt length of that prefill (a multiple of 128 x N)")
    ap.add_argument("--kv16", action="store_true", help="f16 KV cache (half the bytes of the long-context attention stream; the reference's cache is f32): the default of "
                    "the long-context workload c4, where the K/V stream outweighs the weights; parity-tested against the oracle at 2k+ keys (tests/test_decode_parity.py)")
    ap.add_argument("--kv32", action="store_true", help="c4: keep the f32 KV cache (the reference's type, T:1171-1202)")
    ap.add_argument("--exact-act", action="store_true", help="exact f32 activations between the kernels (round 1's path) instead of QAct")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 512 if args.workload == "c4" else 256

    # `python bench.py --gpus N` without a launcher: start N ranks ourselves (one process per GPU) BEFORE anything here
    # touches the GPU, relay rank 0's JSON line and exit with the launcher's code.  Under torchrun WORLD_SIZE must agree.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    if int(env_world or "1") != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world or 1}: launch one rank per GPU (torch.distributed.run --nproc-per-node {args.gpus})")

    import torch

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback path)"
    pkg = importlib.import_module("bitnet-rs_amd")
    synth = importlib.import_module("bitnet-rs_amd.synth")
    dist_ = importlib.import_module("bitnet-rs_amd.dist")
    r = dist_.init("nccl")  # RCCL; one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torchrun)
    world, rank, local_rank = r.world, r.rank, r.local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    n_gpus = world
    if not os.path.exists(pkg.LIB_PATH) or not os.path.exists(pkg.HOST_LIB_PATH):
        pkg.build()
    hip = pkg.load()
    hip.init(local_rank)

    global PROMPT_LEN
    PROMPT_LEN = args.prompt or {"c4": 4096, "c5": 8192}.get(args.workload, 128)
    if args.workload == "stream":  # the streaming-rate probe alone (tools/profile_round.sh runs its counter passes through this)
        res = i2s_stream(hip, synth, args.stream_format, args.stream_layers, launches=args.stream_launches, isolated_only=args.stream_isolated)
        if rank == 0:
            print(json.dumps({"metric": "I2_S matmul HBM GB/s (% roofline), streaming regime", "value": res.get("achieved"), "unit": "GB/s", "n_gpus": n_gpus,
                              "steps": res.get("launches_timed"), "warmup": 1, "higher_is_better": True, "vs_baseline": None, "dtype": "i8 MFMA on QAct",
                              "data": "synthetic", "config": {"workload": f"k_gemv_q over {res['rows']} x {res['cols']} {res['format']}"}, "i2s_stream": res}), flush=True)
        dist_.finalize(r)
        return
    gguf = args.gguf or os.environ.get("BITNET_GGUF")
    cfg, dec, _ = build_model(pkg, synth, args.workload, args.layers, gguf)
    if args.exact_act:
        dec.set_act_mode(0)
    kv16 = args.kv16 or (args.workload == "c4" and not args.kv32)
    if kv16:
        dec.set_kv_f16(True)
    if args.workload == "c5":
        return bench_sharded_prefill(args, pkg, synth, dist_, r, hip, cfg, dec)
    assert PROMPT_LEN + args.warmup + args.steps + 2 < cfg.max_pos
    prompt = synth.prompt(PROMPT_LEN, cfg.vocab)
    dec.reset()
    dec.feed(prompt)
    use_graph = not args.eager
    prefill_ms = prefill_tile = prefill_state = None
    if args.workload == "c4":
        # whole-prompt forward (tiled matmuls + causal attention), untimed warm-up pass then the reported one
        dec.prefill(PROMPT_LEN, with_logits=True, digits=args.digits)
        dec.reset()
        dec.feed(prompt)
        prefill_ms = dec.prefill(PROMPT_LEN, with_logits=True, digits=args.digits)
        prefill_tile = dict(hip.matmul_last_tile(), wave_rows=hip.matmul_last_wave_rows())
        prefill_state = (dec.last_logits().astype(np.float64), int(dec.history(PROMPT_LEN + 1)[PROMPT_LEN]))
    else:
        dec.run(PROMPT_LEN - 1, with_logits=False, use_graph=use_graph)  # prompt positions (KV fill), untimed
        dec.run(1, with_logits=True, use_graph=use_graph)                # first sampled token
    if use_graph:
        dec.prepare_graphs(True)  # every attention form's step graph is built up front (a form first met inside the timed steps
                                  # would otherwise be captured there: ~2 ms of host work, not part of a decode step)
    dec.run(args.warmup, with_logits=True, use_graph=use_graph)      # W untimed warm-up steps
    kv0 = PROMPT_LEN + 1 + args.warmup
    if args.workload == "c4":
        # configs[3] as written: the timed decode steps start right behind the prompt (4097 keys at the first one), so the warm-up's
        # positions are given back: the prompt is prefilled again (untimed) before the timed region
        dec.reset()
        dec.feed(prompt)
        dec.prefill(PROMPT_LEN, with_logits=True, digits=args.digits)
        kv0 = PROMPT_LEN + 1

    ev = {}

    def timed():
        ev["ms"] = dec.run(args.steps, with_logits=True, use_graph=use_graph)  # exactly K timed steps (stream-synchronised inside)

    # barrier + torch.cuda.synchronize() on both sides, MAX over ranks (bitnet-rs_amd/dist.py)
    elapsed = dist_.timed_region(r, timed)
    ev_ms = ev["ms"]
    value = dist_.aggregate_throughput(r, args.steps, elapsed)
    tokens = dec.history(kv0 + args.steps)

    # dominant kernel: the fused gate|up GEMV (largest byte stream of a layer).  Every
    # layer's instance back to back (distinct weights: 30 x 13-17 MB > L2, cycling the
    # whole model through HBM), HIP events on the launch stream, mean per launch.
    us, abytes = dec.probe_gateup(50)
    achieved = abytes / us / 1e3  # GB/s
    roofline = {
        "bound": "hbm",
        "kernel": ("k_gemv_q" if dec.act_mode() else "k_gemv_mfma") + " (fused LayerNorm -> gate|up GEMV -> silu*mul)",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": load_traffic(args.workload),
        "traffic_source": f"profiles/traffic_{args.workload}.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this bench's own launch path (tools/profile_round.sh), per launch; not collected inside this run",
        "bytes_per_launch": int(abytes),
        "us_per_launch": round(us, 3),
    }
    # quote the fraction against the vendor figure AND a stream ceiling measured on this box
    # (read-only non-temporal stream over 2 GiB, best of 10 passes; bitnet_hip_hbm_read_ceiling)
    ceil_best, ceil_mean = hip.hbm_read_ceiling(2 << 30, 10)
    roofline["measured_read_ceiling"] = round(ceil_best, 1)
    roofline["frac_of_measured_ceiling"] = round(achieved / ceil_best, 4)
    kernel_table = {}
    for kind, name in enumerate(("qkv", "attention", "o_proj", "gate_up", "down", "logits")):
        k_us, k_bytes = dec.probe_kernel(kind, 20)
        kernel_table[name] = {"us_per_launch": round(k_us, 2), "GBps": round(k_bytes / k_us / 1e3, 1)}
    # outside the timed region: the fast step against the UNFUSED step on the reference-order (bit-exact) kernels, same
    # weights, same short prompt, at the full model size -- a wrong fast kernel cannot hide behind a plausible rate
    check = exact_step_check(dec, synth, cfg) if not args.no_exact_check else None
    oracle_check = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and not gguf:
        oracle_check = oracle_step_check(dec, synth, cfg, "qk256" if args.workload in ("c3", "c4", "c5") else "i2s")
    stream_res = None
    if not args.no_stream and args.workload in ("c2", "c3") and n_gpus == 1 and not gguf and not args.layers:
        stream_res = i2s_stream(hip, synth, "i2s" if args.workload == "c2" else "qk256", args.stream_layers)
    prefill_chk = prefill_check(dec, prompt, PROMPT_LEN, args.digits, prefill_state) if prefill_state is not None else None
    prefill_roof = prefill_roofline(hip, dec, cfg, PROMPT_LEN, args.digits, True) if prefill_state is not None else None
    # whole-step view of the same metric: all I2_S matrices of one token / step time
    wb = dec.weight_bytes()
    i2s_gbs = wb / (elapsed / args.steps) / 1e9

    if rank == 0:
        out = {
            "metric": "decode tokens/sec + I2_S matmul HBM GB/s (% roofline), bitnet-b1.58-2B-4T",
            "value": round(value, 2),
            "unit": "tokens/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("i8 MFMA on producer-quantised activations (QAct: 15-bit fixed point per element, one power-of-two scale per 16), exact integer "
                      "partial sums, f32 accumulate / f32 elsewhere (f16 embedding table)") if dec.act_mode() else
                     "i8 MFMA on exact 30-bit fixed-point activations, f32 accumulate / f32 elsewhere (f16 embedding table)",
            "data": f"gguf:{os.path.basename(gguf)}" if gguf else "synthetic",
            "config": {
                "workload": f"bitnet-b1.58-2B-4T I2_S BitNet32-F16 (ternary, one f16 scale per 32 weights), {n_gpus}xMI355X, batch=1 decode, {PROMPT_LEN}-token prompt"
                if args.workload == "c2"
                else f"bitnet-b1.58-2B-4T I2_S QK256 blocks, {n_gpus}xMI355X, batch=1 decode, {PROMPT_LEN}-token prompt" if args.workload == "c3"
                else f"bitnet-b1.58-2B-4T I2_S QK256 blocks, {n_gpus}xMI355X, {PROMPT_LEN}-token prefill + decode",
                "layers": cfg.n_layers,
                "prompt_len": PROMPT_LEN,
                "kv_len_during_timing": [kv0, kv0 + args.steps],
                "parallelism": f"replicas x{n_gpus}" if n_gpus > 1 else "single GPU",
                "launch": "hipGraph replay per token" if use_graph else "eager",
                "kv_cache": "f16 (values rounded once, when appended; --kv32 keeps the reference's f32)" if kv16 else "f32 (as the reference, T:1171-1202)",
            },
            "i2s_matmul_gbs_whole_step": round(i2s_gbs, 1),
            "i2s_matmul_frac_of_hbm_peak_whole_step": round(i2s_gbs / HBM_PEAK_GBS, 4),
            "i2s_weight_bytes_per_token": int(wb),
            "event_ms_per_step": round(ev_ms / args.steps, 4),
            "roofline": roofline,
            "per_kernel": kernel_table,
            "last_tokens": [int(t) for t in tokens[-4:]],
            "distinct_tokens_in_timed_steps": int(len(set(int(t) for t in tokens[-args.steps:]))),
            "exact_step_check": check,
            "oracle_step_check": oracle_check,
        }
        if stream_res is not None:
            out["i2s_stream"] = stream_res
        if prefill_ms is not None:
            flops = 2.0 * 2_084_044_800 * (cfg.n_layers / 30) * PROMPT_LEN + 4.0 * PROMPT_LEN * PROMPT_LEN / 2 * cfg.n_heads * cfg.head_dim * cfg.n_layers
            out["prefill"] = {"tokens": PROMPT_LEN, "ms": round(prefill_ms, 2), "tokens_per_s": round(PROMPT_LEN / prefill_ms * 1e3, 1),
                              "digits": args.digits, "eff_TFLOPs": round(flops / prefill_ms / 1e9, 1),
                              "note": "whole-prompt forward incl. first sampled token; QK256: q|k|v and gate|up on i8 MFMA digit planes behind their LayerNorm, o / down on f16 MFMA straight from the f16 rows their producers wrote (hybrid, DESIGN 4.6); BitNet32-F16: the f16 activation chain; attention on f16 MFMA",
                              "last_matmul_tile": prefill_tile,  # the prompt's last matmul (a 2560-row down-projection: f16 MFMA in 320-row workgroups at 4096 rows); roofline.tile = gate|up's
                              "prefill_check": prefill_chk, "roofline": prefill_roof}
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, synth)
    also_failed = False
    if n_gpus == 1 and args.workload == "c2" and not args.no_also and not gguf and not args.layers and not args.exact_act:
        dec.close()  # the headline model's memory goes back first
        try:
            out["also"] = also_workloads(args, pkg, synth, hip, use_graph)
        except Exception as e:  # noqa: BLE001 -- reported in the line (the headline numbers above still reach the driver) AND as a failed run
            out["also"] = {"error": f"{type(e).__name__}: {e}"}
            out["failed"] = "also: c3 / c4 / BitNet32-F16 prefill did not complete (exit code 1)"
            also_failed = True
    # N > 1: the decode line above is N independent replicas (batch-1 decode does not shard); the path's ONE real collective
    # -- the token-parallel prefill of BASELINE configs[4] -- runs here too, over the same ranks, and rides in the same line
    c5 = None
    if n_gpus > 1 and args.workload in ("c2", "c3") and not args.no_c5:
        # The decode numbers above must reach the driver whatever this first multi-GPU execution of the collective does:
        # every rank arms the same timer; if the prefill has not finished by then, rank 0 prints the line with the reason and
        # every rank leaves (a rank that waits in a collective for one that failed would otherwise sit there until the
        # launcher's own limit).
        import threading

        line_lock = threading.Lock()  # the line is printed exactly once: by the timer or by the main thread, whoever gets here first
        printed = []

        def give_up():
            with line_lock:
                if printed:
                    return
                printed.append("timeout")
                if rank == 0:
                    out["prefill_c5"] = {"error": f"token-parallel prefill did not finish within {args.c5_timeout} s; decode replicas above are unaffected"}
                    print(json.dumps(out), flush=True)
            os._exit(3)  # a hung collective is a FAILED run (the decode line is still on stdout): the launcher must see it

        timer = threading.Timer(args.c5_timeout, give_up)
        timer.daemon = True
        timer.start()
        try:
            dec.close()
            cfg5, dec5, _ = build_model(pkg, synth, "c5", args.layers)
            dec = dec5
            c5 = sharded_prefill(args, pkg, synth, dist_, r, cfg5, dec5, args.c5_prompt, 2, 1)
        except Exception as e:  # noqa: BLE001 -- reported in the line, not swallowed
            c5 = {"error": f"{type(e).__name__}: {e}"}
        timer.cancel()
        with line_lock:
            if printed:  # the timer fired while the prefill was finishing: its line stands, and so does its exit code
                os._exit(3)
            printed.append("main")
    if rank == 0:
        if c5 is not None:
            out["prefill_c5"] = c5
        print(json.dumps(out), flush=True)
    dec.close()
    dist_.finalize(r)
    if also_failed:
        sys.exit(1)  # a broken prefill / QK256 decode path must not look like a clean run


if __name__ == "__main__":
    main()
