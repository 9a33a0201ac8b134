#!/bin/bash
# round 6: the decoder branch's stream inside one T1 training step -- every launch with the gap in front of it and what the other streams
# ran meanwhile (rocprofv3 kernel trace of tools/bench_train.py 32 3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/${1:-branch}; mkdir -p $O
rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- python3 tools/bench_train.py 32 3 > $O/train.log 2>&1
python3 - $O/tr/t_kernel_trace.csv <<'PY' | tee $O/branch_gaps.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a0, a1 = adam[-3], adam[-2]
step = rows[a0 + 1:a1 + 1]
t0 = int(rows[a0]['End_Timestamp'])
def short(n): return n.replace('vnr::', '').split('(')[0][:44]
byq = collections.defaultdict(list)
for r in step: byq[r['Stream_Id'] if 'Stream_Id' in r else r['Queue_Id']].append(r)
qs = sorted(byq, key=lambda q: -sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in byq[q]))
print("streams by kernel time:", [(q, len(byq[q]), round(sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in byq[q]) / 1e6, 2)) for q in qs])
br = qs[2] if len(qs) > 2 else qs[-1]
prev = None
for r in byq[br]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    line = "+%8.3f ms  %-44s %7.1f us   gap before %7.1f us" % ((s - t0) / 1e6, short(r['Kernel_Name']), (e - s) / 1e3, gap)
    if gap > 40:
        others = collections.Counter()
        for q in qs:
            if q == br: continue
            for x in byq[q]:
                xs, xe = int(x['Start_Timestamp']), int(x['End_Timestamp'])
                ov = min(xe, s) - max(xs, prev)
                if ov > 0: others[(q, short(x['Kernel_Name']))] += ov
        line += "   <- meanwhile: " + ", ".join("%s %s %.0f us" % (q, n, t / 1e3) for (q, n), t in others.most_common(4))
    print(line)
    prev = max(prev or 0, e)
PY
