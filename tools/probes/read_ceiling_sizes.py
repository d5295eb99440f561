import importlib, sys
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("bitnet-rs_amd"); hip = pkg.load(); hip.init(0)
for mb in (657, 1314, 2048):
    best, mean = hip.hbm_read_ceiling(mb << 20, 20)
    print(mb, "MB: best", round(best, 1), "GB/s mean", round(mean, 1), "->", round((mb << 20) / best / 1e3, 1), "us")
