"""Developer tool (GPU box): the one-token attention op (RoPE + append + GQA softmax; 64- and 128-position forms, f32 and f16
caches, f32 and QAct outputs) at random head counts and context lengths against a f64 numpy reference -- the differential companion of
tests/test_decode_parity.py::test_attention_decode_op_vs_f64.   python tools/random_sweep_attn.py [n] [seed]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bitnet-rs_amd")
from oracle import oracle  # noqa: E402
from tests.qact_ref import dequantize_qact  # noqa: E402

hip = pkg.load(); hip.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
D, bad = 128, 0
for case in range(n_cases):
    n_heads, n_kv = [(20, 5), (8, 2), (4, 2), (3, 3), (4, 1), (2, 2)][int(rng.integers(0, 6))]
    max_pos = int(rng.choice([64, 192, 512, 1024, 4160]))
    pos = int(rng.integers(0, max_pos - 1))
    wide, kv16, qout = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and (n_heads * D) % 256 == 0
    group = n_heads // n_kv
    mp64 = (max_pos + 63) // 64 * 64
    sin, cos = oracle.rope_tables(D, max_pos, 10000.0)
    kc = rng.normal(0, 1, (n_kv, D, mp64)).astype(np.float32)
    vc = rng.normal(0, 1, (n_kv, mp64, D)).astype(np.float32)
    if kv16:
        kc, vc = kc.astype(np.float16).astype(np.float32), vc.astype(np.float16).astype(np.float32)
    kc_in, vc_in = kc.copy(), vc.copy()
    kc_in[:, :, pos:] = 3.0   # stale slots at / past the new token: finite garbage
    vc_in[:, pos:] = -7.0
    if kv16:  # K [kv][chunk][D/2][64][2] halves, V [kv][pos][D] halves
        kt = kc_in.reshape(n_kv, D // 2, 2, mp64 // 64, 64).transpose(0, 3, 1, 4, 2)
        kcd, vcd = dev(np.ascontiguousarray(kt).astype(np.float16)), dev(vc_in.astype(np.float16))
    else:
        kcd, vcd = dev(kc_in.reshape(n_kv, D, mp64 // 64, 64).transpose(0, 2, 1, 3)), dev(vc_in)
    qkv = rng.normal(0, 1.5, (n_heads + 2 * n_kv) * D).astype(np.float32)
    sb = hip.c.bitnet_hip_attention_scratch_bytes(n_kv, max_pos)
    scratch = torch.zeros(sb // 4 + 16, device="cuda")
    out = torch.full((n_heads * D,), float("nan"), device="cuda")
    qa = torch.zeros(hip.qact_bytes(n_heads * D), dtype=torch.uint8, device="cuda") if qout else None
    pos_d = torch.tensor([pos], dtype=torch.int32, device="cuda")
    try:
        hip.attention_decode_q_dev(dev(qkv), dev(sin), dev(cos), kcd, vcd, n_heads, n_kv, D, max_pos, pos_d, scratch, out, qa, wide=wide, kv_f16=kv16)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(n_heads, D).astype(np.float64)
        rot = lambda x: np.concatenate([x[..., :64] * cos[pos] - x[..., 64:] * sin[pos], x[..., :64] * sin[pos] + x[..., 64:] * cos[pos]], axis=-1)
        q = rot(qkv[: n_heads * D].reshape(n_heads, D).astype(np.float64))
        kn = rot(qkv[n_heads * D:(n_heads + n_kv) * D].reshape(n_kv, D).astype(np.float64))
        vn = qkv[(n_heads + n_kv) * D:].reshape(n_kv, D).astype(np.float64)
        if kv16:
            kn, vn = kn.astype(np.float16).astype(np.float64), vn.astype(np.float16).astype(np.float64)
        want = np.zeros((n_heads, D))
        for h in range(n_heads):
            kvh = h // group
            K = np.concatenate([kc[kvh, :, :pos].T.astype(np.float64), kn[kvh][None]], axis=0)
            V = np.concatenate([vc[kvh, :pos].astype(np.float64), vn[kvh][None]], axis=0)
            s = K @ q[h] / np.sqrt(D)
            p = np.exp(s - s.max())
            want[h] = (p / p.sum()) @ V
        tol = (2e-5 if not kv16 else 2e-3) * max(1.0, np.abs(want).max())
        err = float(np.max(np.abs(got - want)))
        ok = np.isfinite(got).all() and err <= tol
        if ok and qout:
            back = dequantize_qact(qa.cpu().numpy(), n_heads * D).reshape(n_heads, D)
            gmax = np.abs(got).reshape(-1, 16).max(1).repeat(16).reshape(n_heads, D)
            ok = bool(np.all(np.abs(back - got) <= np.maximum(gmax, 2.0 ** -94) * 2.0 ** -14))
    except pkg.BitNetHipError as e:
        ok, err, tol = False, repr(e), 0
    if not ok:
        bad += 1
        print("FAIL", n_heads, n_kv, max_pos, pos, "wide" if wide else "", "kv16" if kv16 else "", "qout" if qout else "", err, tol, flush=True)
print(f"{n_cases - bad}/{n_cases} cases agree", flush=True)
sys.exit(1 if bad else 0)
