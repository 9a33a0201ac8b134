for s in 3 4 5 3 4; do
python bench.py --no-cpu-baseline --no-train --streams $s --steps 40 --profile-steps 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('streams $s', round(d['value']), round(d['ms_per_step'],4))"
done
