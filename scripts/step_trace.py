#!/usr/bin/env python3
"""Prints the kernel launches of the last training step found in a rocprofv3 kernel trace (csv):
duration, grid and a short kernel name, in launch order.  Usage: step_trace.py <dir-with-*_kernel_trace.csv>"""
import csv
import glob
import sys

import os
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'sumsq_partial' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
total = 0.0
for r in rows[a + 1:b + 1]:
  n = r['Kernel_Name']
  d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
  total += d
  short = n[n.index('gemm'):n.index('>') + 1] if 'gemm' in n else n.split('(')[0].split('::')[-1][-40:]
  if d >= float(sys.argv[2]) if len(sys.argv) > 2 else True:
    print('%8.1f us  wg=%-6d %s' % (d, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), short))
print('kernel time %.1f us, span %.1f us' % (total, (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3))
