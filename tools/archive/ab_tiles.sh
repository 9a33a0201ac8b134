python tools/parity_s1.py 2>&1 | tail -5
for i in 1 2; do
for t in "--opt gemm_wide_tiles=0" ""; do
python bench.py --no-cpu-baseline --no-train --streams 3 --steps 40 $t 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$t', round(d['ms_per_step'],4), round(d['value']))"
done; done
