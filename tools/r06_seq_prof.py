"""Round 6: per-dispatch durations of the gemm2 launches along ONE forward pass, averaged over the last 20 passes of each rocprofv3
kernel trace given (tools/r06_ks_prof.sh output directories): python tools/r06_seq_prof.py <dir> <label> [<label> ...]"""
import csv, sys
def load(d, label):
    rows = list(csv.DictReader(open(f'{d}/{label}/{label}_kernel_trace.csv')))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'gather_rows' in r['Kernel_Name']]
    ss = [rows[idx[j]:idx[j + 1]] for j in range(len(idx) - 1)]
    n = len(ss[-1]); ss = [s for s in ss if len(s) == n][-20:]
    return [(ss[-1][i]['Kernel_Name'], ss[-1][i]['Grid_Size_X'], sum((int(s[i]['End_Timestamp']) - int(s[i]['Start_Timestamp'])) for s in ss) / len(ss) / 1e3) for i in range(n)]
d, labels = sys.argv[1], sys.argv[2:]
cols = [load(d, l) for l in labels]
print("%-52s %8s " % ("kernel", "threads") + " ".join("%9s" % l for l in labels))
for i, (k, g, _) in enumerate(cols[0]):
    if 'gemm2_kernel' in k:
        print("%-52s %8s " % (k[17:69], g) + " ".join("%9.2f" % (c[i][2] if i < len(c) and c[i][0][:40] == k[:40] else float('nan')) for c in cols))
