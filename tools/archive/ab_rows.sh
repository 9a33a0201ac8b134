python tools/parity_s1.py 2>&1 | tail -3
for o in "--streams 3" "--streams 3" "--opt chain_rows64=1 --streams 1"; do
python bench.py --no-cpu-baseline --no-train $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$o', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['end_to_end']['kernel_ms_per_step'].items()})"
done
