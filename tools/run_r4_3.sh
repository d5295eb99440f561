set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_f16_chain.py -x -q -m gpu > gpurun_out/r4_t3.log 2>&1; echo "pytest chain rc $?" >> gpurun_out/r4_t3.log
python -m pytest tests/test_bench_prefill_instance.py tests/test_prefill_parity.py tests/test_headline_parity.py tests/test_decode_parity.py -x -q -m gpu >> gpurun_out/r4_t3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_t3.log
python bench.py --steps 64 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?" >> gpurun_out/r4_t3.log
tail -30 gpurun_out/r4_t3.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
a=d['also']
print('c2',d['value'],'c3',a['c3']['value'],'c4',a['c4']['value'],'prefill qk',a['c4']['prefill']['ms'],'prefill i2s',a['prefill_i2s']['ms'], a['prefill_i2s']['prefill_check'])
PY
