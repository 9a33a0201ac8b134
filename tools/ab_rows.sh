python -m pytest tests/test_gpu_model.py -x -q -k "inference_matches or long_text or full_size" 2>&1 | tail -3
for o in "--opt chain_rows64=0 --streams 1" "--opt chain_rows64=1 --streams 1" "--opt chain_rows64=0 --streams 3" "--opt chain_rows64=1 --streams 3" "--opt chain_rows64=1 --streams 2" "--opt chain_rows64=1 --streams 4"; do
python bench.py --no-cpu-baseline --no-train $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$o', round(d['ms_per_step'],4), round(d['value']), d['parity'] if 'parity' in d else '', {k:round(v,4) for k,v in d['end_to_end']['kernel_ms_per_step'].items()})"
done
