set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gemm_parity.py tests/test_f16_chain.py tests/test_bench_prefill_instance.py tests/test_prefill_parity.py tests/test_headline_parity.py -x -q -m gpu > gpurun_out/r4_t5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t5.log
python bench.py --steps 64 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?" >> gpurun_out/r4_t5.log
BITNET_HOST_PREFILL_CHAIN=0 python bench.py --steps 64 --no-cpu-baseline --no-stream > gpurun_out/bench_nochain.json 2> gpurun_out/bench_nochain.err; echo "bench0 rc $?" >> gpurun_out/r4_t5.log
tail -12 gpurun_out/r4_t5.log
python - <<'PY'
import json
for f in ('gpurun_out/bench_default.json','gpurun_out/bench_nochain.json'):
    d=json.load(open(f))
    a=d['also']
    print(f,'c2',d['value'],'c3',a['c3']['value'],'c4',a['c4']['value'],'prefill qk',a['c4']['prefill']['ms'],a['c4']['prefill']['prefill_check']['logits_cosine_vs_4_digits'],'prefill i2s',a['prefill_i2s']['ms'], a['prefill_i2s']['prefill_check']['logits_cosine_vs_4_digits'])
PY
