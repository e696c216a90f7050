#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 kernel trace (csv): start (us, relative to the end of the previous
update's gradient-norm kernel: the window opens with that update's clip + Adam launch), duration, stream, kernel -- side-stream kernels and gaps of the compute stream become visible.
usage: step_timeline.py <kernel_trace.csv> [step-from-the-end, default 2]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seq = [(re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')[:40],
        int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id']) for r in rows]
seq.sort(key=lambda x: x[1])
idx = [i for i, (k, _, _, _) in enumerate(seq) if 'sumsq_partial' in k]
a, b = idx[-back - 1], idx[-back]
t0 = seq[a][2]
main = max(set(s for _, _, _, s in seq[a:b]), key=lambda s: sum(e - st for _, st, e, ss in seq[a:b] if ss == s))
last_end, busy = t0, 0
names = {}                 # other streams: sidea, sideb, ... in order of appearance
for k, s, e, st in seq[a + 1:b + 1]:
  gap = ''
  if st == main:
    if s - last_end > 3000:
      gap = '   <-- compute stream idle %.0f us' % ((s - last_end) / 1e3)
    last_end = max(last_end, e)
    busy += e - s
  print('%8.1f %7.1f  %s %s%s' % ((s - t0) / 1e3, (e - s) / 1e3, 'main' if st == main else 'side' + names.setdefault(st, chr(ord('a') + len(names))), k, gap))
print('step %.1f us, compute-stream kernels %.1f us' % ((seq[b][2] - t0) / 1e3, busy / 1e3))
