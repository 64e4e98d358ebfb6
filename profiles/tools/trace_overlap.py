"""Reads a rocprofv3 kernel trace (CSV) of bench.py and prints, for the timed steps, wall time per step, the time at least
one kernel was running, the sum of kernel durations, per-kernel averages and the timeline of one step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r['Stream_Id']) for r in rows
      if 'k_synth' not in r['Kernel_Name']]
ev.sort()
bgs = [i for i, e in enumerate(ev) if 'k_bg_stream' in e[2]]
a, b = bgs[4], bgs[-2]
nst = len(bgs) - 2 - 4
t0, t1 = ev[a][0], ev[b][0]
seg = ev[a:b]
iv = sorted((e[0], e[1]) for e in seg)
busy = 0
cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(e[1] - e[0] for e in seg)
print(f"{nst} steps: wall {(t1 - t0) / nst / 1e3:.0f} us/step, some kernel running {busy / nst / 1e3:.0f} us/step, "
      f"sum of kernel durations {tot / nst / 1e3:.0f} us/step")
d = collections.defaultdict(list)
for e in seg:
    d[e[2]].append((e[1] - e[0]) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f"{k[:30]:30s} n {len(v):3d} avg {sum(v) / len(v):8.1f} us")
if len(sys.argv) > 2:
    for e in ev[bgs[6] - 3:bgs[6] + int(sys.argv[2])]:
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f} s{e[3]} {e[2][:28]}")
