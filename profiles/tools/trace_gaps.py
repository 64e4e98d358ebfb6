"""Idle time of the GPU in a rocprofv3 kernel trace of bench.py: intervals with no kernel running inside the timed steps,
summed by (kernel that ended, kernel that started)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in rows
      if 'k_synth' not in r['Kernel_Name']]
ev.sort()
bgs = [i for i, e in enumerate(ev) if 'k_bg_stream' in e[2]]
a, b = bgs[4], bgs[-2]
nst = len(bgs) - 2 - 4
seg = ev[a:b]
gaps = collections.defaultdict(float)
cur_end, cur_name = seg[0][1], seg[0][2]
for s, e, n in seg[1:]:
    if s > cur_end:
        gaps[(cur_name[:22], n[:22])] += s - cur_end
    if e > cur_end:
        cur_end, cur_name = e, n
tot = sum(gaps.values())
print(f"{nst} steps, wall {(ev[b][0] - ev[a][0]) / nst / 1e3:.0f} us/step, idle {tot / nst / 1e3:.0f} us/step")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:14]:
    print(f"{v / nst / 1e3:7.1f} us/step  after {k[0]:22s} before {k[1]}")
