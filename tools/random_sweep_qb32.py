"""Developer tool (GPU box): random shapes through the QB32 entry points (bitnet_hip_rows_to_qb32_dev -> bitnet_hip_matmul_qb32_dev: plain, LayerNorm
after the product, residual, silu * up -> f16 rows) and the row-scaled fp6 form on / off the resident fp4 image, against the CPU oracle.
    python tools/random_sweep_qb32.py [n] [seed]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
from oracle import oracle  # noqa: E402  (checker)

hip = pkg.load(); hip.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    rows = 256 * int(rng.choice([1, 2, 3, 5, 10, 15]))
    cols = 256 * int(rng.choice([1, 2, 3, 4, 10, 27]))
    m = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 130, 300, 700]))
    silu = bool(rng.integers(0, 3) == 0)
    if silu and cols > 5120:
        cols = 2560  # (a paired matrix binds its LayerNorm through the GEMV's K split: at most 20 blocks of 256 columns per row tile pair)
    ln = silu or bool(rng.integers(0, 2))  # silu * up leaves as f16 rows: only behind a LayerNorm are its values sure to fit (else: bitnet_hip_f16_saturations)
    res = (not silu) and bool(rng.integers(0, 2))
    stride = cols // 256 * 64
    x = (rng.normal(rng.uniform(-0.5, 0.5), 1.0, (m, cols)) * np.exp(rng.uniform(-4, 4, (m, 1)))).astype(np.float32)
    gam = (rng.uniform(0.5, 1.5, cols) / 40).astype(np.float32)
    gd = torch.from_numpy(gam).cuda()
    mp = -(-m // 64) * 64
    try:
        if silu:
            qg, qu = (rng.integers(0, 256, rows * stride, dtype=np.uint8) for _ in range(2))
            hg, hu = hip.weights_upload_qk256(qg, rows, cols, stride), hip.weights_upload_qk256(qu, rows, cols, stride)
            h = hip.weights_concat([hg, hu], interleave16=True)
            hip.weights_free(hg); hip.weights_free(hu)
        else:
            qs = rng.integers(0, 256, rows * stride, dtype=np.uint8)
            h = hip.weights_upload_qk256(qs, rows, cols, stride)
        if ln:
            hip.weights_bind_ln(h, gd)
        xin = np.stack([oracle.layernorm(x[i], gam, 1e-5) for i in range(m)]) if ln else x
        if silu:
            a = np.stack([oracle.gemv_qk256(qg, xin[i], rows, cols, stride) for i in range(m)]).astype(np.float64)
            b = np.stack([oracle.gemv_qk256(qu, xin[i], rows, cols, stride) for i in range(m)]).astype(np.float64)
            want = a / (1.0 + np.exp(-a)) * b
        else:
            want = np.stack([oracle.gemv_qk256(qs, xin[i], rows, cols, stride) for i in range(m)]).astype(np.float64)
        resid = rng.normal(0, 1, (m, rows)).astype(np.float32)
        if res:
            want = want + resid
        qb = torch.zeros(hip.qb32_bytes(m, cols), dtype=torch.uint8, device="cuda")
        st = torch.zeros(2 * mp, device="cuda")
        hip.rows_to_qb32_dev(torch.from_numpy(x).cuda(), gd if ln else None, m, cols, qb, st)
        kw = dict(stats_in=st, n_stats=1, ln_gamma=gd, ln_eps=1e-5) if ln else {}
        if silu:
            yh = torch.full((mp, rows), float("nan"), dtype=torch.float16, device="cuda")
            hip.matmul_qb32_dev(h, qb, m, flags=1, yh=yh, **kw)
            got = yh.cpu().numpy()[:m].astype(np.float64)
            tol = 2e-3 * max(1e-6, float(np.max(np.abs(want))))  # f16 output rows
        else:
            y = torch.full((m, rows), float("nan"), device="cuda")
            hip.matmul_qb32_dev(h, qb, m, y=y, residual=torch.from_numpy(resid).cuda() if res else None, **kw)
            got = y.cpu().numpy().astype(np.float64)
            tol = 3e-4 * max(1.0, float(np.max(np.abs(want)))) + 2e-4 * np.sqrt(cols / 256.0)
        torch.cuda.synchronize()
        err = float(np.max(np.abs(got - want)))
        ok = np.isfinite(got).all() and err <= tol
        if ok and not silu and not ln:  # the row-scaled fp6 form on the image and expanding in the loop: identical bits
            wsb = hip.matmul_workspace_bytes(m, cols, 2)
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            ya, yb = torch.empty(m, rows, device="cuda"), torch.empty(m, rows, device="cuda")
            hip.matmul_fused_dev(h, torch.from_numpy(x).cuda(), ya, m, ws, wsb, digits=2, flags=16)
            hip.matmul_fused_dev(h, torch.from_numpy(x).cuda(), yb, m, ws, wsb, digits=2, flags=16 | 32)
            torch.cuda.synchronize()
            ok = bool(torch.equal(ya, yb))
            err = "resident != expanding" if not ok else err
    except pkg.BitNetHipError as e:
        ok, err, tol = False, repr(e), 0
    if not ok:
        bad += 1
        print("FAIL", rows, cols, m, "silu" if silu else "", "ln" if ln else "", "res" if res else "", err, tol, flush=True)
    hip.weights_free(h)
print(f"{n_cases - bad}/{n_cases} cases agree with the oracle", flush=True)
sys.exit(1 if bad else 0)
