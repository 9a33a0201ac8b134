for t in "" "VNR_SPLIT_TILE=1" "VNR_SPLIT_TILE=0" "VNR_SPLIT_TILE=1 VNR_SPLIT_STAGES=4" "VNR_SPLIT_TILE=0 VNR_SPLIT_STAGES=4"; do
env $t python bench.py --no-cpu-baseline --no-train --streams 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$t', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['end_to_end']['kernel_ms_per_step'].items()})"
done
