#!/bin/bash
# round 6: the kernel-gradient GEMMs (gemm_tn3_kernel) of one training step by launch geometry (rocprofv3 kernel trace of tools/bench_train.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tn3}; mkdir -p $O
rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- python3 tools/bench_train.py 32 3 > $O/train.log 2>&1
python3 - $O/tr/t_kernel_trace.csv <<'PY' | tee $O/tn3_shapes.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if 'gemm_tn3' in r['Kernel_Name'] or 'col_sum4' in r['Kernel_Name'] or 'tn_finish' in r['Kernel_Name'] or 'gemm_tn' in r['Kernel_Name']:
        k = (r['Kernel_Name'][:40], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
        acc[k][0] += 1; acc[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("all kernels: %.1f ms over %d dispatches" % (tot / 1e6, len(rows)))
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-42s wgs %6d y %s z %s  calls %5d  total %9.1f us  avg %7.2f us" % (k[0], k[1], k[2], k[3], n, t / 1e3, t / 1e3 / n))
PY
tail -3 $O/train.log
