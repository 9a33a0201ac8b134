for o in "--opt late_dec_kv=1" "--opt late_dec_kv=0" "--opt late_dec_kv=1" "--opt late_dec_kv=0"; do
python bench.py --no-cpu-baseline --no-train --streams 1 $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$o', round(d['ms_per_step'],4), round(d['roofline_cross_attention']['avg_launch_us'],2), {k:round(v,4) for k,v in d['end_to_end']['kernel_ms_per_step'].items()})"
done
for o in "--opt late_dec_kv=1" "--opt late_dec_kv=0"; do
python bench.py --no-cpu-baseline --no-train $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('3 streams $o', round(d['ms_per_step'],4), round(d['value']))"
done
