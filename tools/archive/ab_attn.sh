for o in "" "--opt attn_presplit_self=0" "--opt attn_presplit_self=0 --opt attn_presplit=0" ""; do
python bench.py --no-cpu-baseline --no-train $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$o', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['end_to_end']['kernel_ms_per_step'].items()})"
done
