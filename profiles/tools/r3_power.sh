#!/bin/bash
# clocks and power of the GPU while the bench runs (rocm-smi sampled every 0.2 s): r3_power.sh <prefix> [bench args]
cd $GRAFT_REPO_ROOT
P=${1:-pw}; shift
( for i in $(seq 1 200); do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > gpurun_out/${P}_smi.jsonl &
SMI=$!
timeout -k 10 300 python bench.py --steps ${PW_STEPS:-600} --warmup 5 --cpu off --host-batches 0 "$@" > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err
kill $SMI 2>/dev/null
python - gpurun_out/${P}_smi.jsonl gpurun_out/${P}_bench.json <<'PY'
import json, sys
rows = []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line: continue
    try: d = json.loads(line)
    except Exception: continue
    for card, v in d.items():
        if isinstance(v, dict): rows.append(v)
keys = sorted({k for r in rows for k in r})
print("fields:", [k for k in keys][:40])
def num(x):
    try: return float(str(x).strip("()MhzWwCc ").replace("Mhz", ""))
    except Exception: return None
for k in keys:
    vals = [num(r.get(k)) for r in rows]
    vals = [v for v in vals if v is not None]
    if vals: print(f"{k[:60]:60s} n {len(vals):3d} min {min(vals):9.1f} max {max(vals):9.1f} first {vals[0]:9.1f} last {vals[-1]:9.1f}")
b = json.load(open(sys.argv[2])); print("bench", b["value"], b["ms_per_step"])
PY
python - gpurun_out/${P}_smi.jsonl <<'PY'
import json, sys
for i, line in enumerate(open(sys.argv[1])):
    try: d = json.loads(line)
    except Exception: continue
    for card, v in d.items():
        if isinstance(v, dict) and i % 3 == 0:
            print(i, v.get('sclk clock speed:'), v.get('Temperature (Sensor junction) (C)'), v.get('Temperature (Sensor memory) (C)'), {k: x for k, x in v.items() if 'ower' in k})
PY
