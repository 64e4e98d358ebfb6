"""prints a stretch of a rocprofv3 kernel trace as a timeline: start, end (us, relative), stream / queue, kernel"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Queue_Id', '?'), r.get('Stream_Id', '?'))
      for r in rows if 'k_synth' not in r['Kernel_Name']]
ev.sort()
bgs = [i for i, e in enumerate(ev) if 'k_bg_stream' in e[2]]
a = bgs[len(bgs) // 2]
if len(sys.argv) > 3:                                        # start 30 kernels before the given kernel's first launch after the middle
    hit = [i for i, e in enumerate(ev) if sys.argv[3] in e[2] and i > a]
    if hit:
        a = max(hit[0] - 30, 0)
t0 = ev[a][0]
for s, e, n, q, st in ev[a:a + int(sys.argv[2]) if len(sys.argv) > 2 else a + 80]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  q{q:>3s} s{st:>3s}  {n[:40]}")
