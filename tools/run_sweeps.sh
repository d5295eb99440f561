set -o pipefail
for t in "random_sweep.py 200" "random_sweep_qact.py 150" "random_sweep_attn.py 60" "random_sweep_prefill_attn.py 120" "random_sweep_provider.py 60" "random_sweep_decoder.py 12" "random_sweep_qb32.py 80"; do
  echo "== $t"; timeout -k 10 500 python tools/$t 2>&1 | tail -n 2 || echo "FAILED $t"
done
