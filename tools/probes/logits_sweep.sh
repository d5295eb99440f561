for W in 256 512 1024 2048; do for R in 2 3 4; do
BITNET_HOST_LOGITS_WGS=$W BITNET_HIP_LOGIT_ROWS=$R timeout -k 10 200 python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wgs', $W, 'rows', $R, d['per_kernel']['logits'], d['value'])"
done; done
