#!/usr/bin/env python3
"""Instruction histogram of one kernel in a device assembly file (hipcc -S --cuda-device-only): whole kernel and its innermost loops.
usage: isa_hist.py file.s mangled-name-substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'\n(\S*%s\S*):' % re.escape(sys.argv[2]), s)
i = m.start(); j = s.index('s_endpgm', i)
lines = s[i:j].split('\n')
def hist(a, b):
    c = collections.Counter()
    for l in lines[a:b]:
        t = l.strip(); mm = re.match(r'([a-z_0-9]+)', t)
        if mm and not t.endswith(':') and not t.startswith('.') and not t.startswith(';'): c[mm.group(1)] += 1
    return sum(c.values()), ', '.join('%s %d' % kv for kv in c.most_common(40))
print(m.group(1)); print('whole: %d  %s' % hist(0, len(lines)))
lab = {}
for n, l in enumerate(lines):
    mm = re.match(r'(\.LBB\d+_\d+):', l.strip())
    if mm: lab[mm.group(1)] = n
loops = set()
for n, l in enumerate(lines):
    mm = re.match(r'\s*s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in lab and lab[mm.group(1)] < n: loops.add((lab[mm.group(1)], n))
# innermost loops only (no other loop strictly inside), largest bodies first
inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
for a, b in sorted(inner, key=lambda x: x[0] - x[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 8]:
    if b - a < 40: continue
    print('innermost loop lines %d-%d: %d  %s' % ((a, b) + hist(a, b)))
