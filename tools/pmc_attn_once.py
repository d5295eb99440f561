"""Developer tool: four 4096-token prefill attention calls (2B-4T heads) for a rocprofv3 --pmc pass.
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_attn -- python tools/pmc_attn_once.py"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
NH, NK, D, T = 20, 5, 128, 4096
sin = torch.zeros(T, D // 2, device="cuda"); cos = torch.ones(T, D // 2, device="cuda")
kc, vc = torch.zeros(NK * T * D, device="cuda"), torch.zeros(NK * T * D, device="cuda")
qkv = torch.randn(T, (NH + 2 * NK) * D, device="cuda")
wsb = hip.attention_prefill_workspace_bytes(NH, NK, T); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
out = torch.empty(T, NH * D, device="cuda")
for _ in range(4):
    hip.attention_prefill_dev(qkv, sin, cos, kc, vc, NH, NK, D, T, T, ws, wsb, out)
torch.cuda.synchronize()
