#!/bin/bash
# round 6: per-kernel times of the encoder GEMMs with and without the k-split wave sets (VNR_GEMM_KS) and ring depths, one box
# usage: r06_ks_prof.sh <outdir> "<label>|<env assignments>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-ksprof}; shift; mkdir -p $out
for cfg in "$@"; do
  IFS='|' read -r label envs <<< "$cfg"
  for e in $envs; do export $e; done
  rocprofv3 --kernel-trace --stats -d $out/$label -o $label --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-exact-pass --in-flight 0 --no-attn-phase > $out/$label.log 2>&1
  for e in $envs; do unset ${e%%=*}; done
  echo "== $label ($envs)"
  python3 - $out/$label/${label}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm2_kernel' in r['Name'] and (', 4, ' in r['Name']):
        print("  %-52s calls %4s avg %7.2f us  min %7.2f" % (r['Name'][17:66], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done 2>&1 | tee $out/summary.txt
