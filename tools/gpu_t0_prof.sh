#!/bin/bash
# kernel-trace of the T0_3B few-shot tool: which kernels the T5 encoder / decoder passes spend their time in
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/t0p
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/t0p -o run --output-format csv -- python3 $R/tools/t0_fewshot_bench.py > /tmp/t0p.log 2>&1 || { tail -5 /tmp/t0p.log; exit 1; }
grep -v amdgpu.ids /tmp/t0p.log | tail -7
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/t0p/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.2f} ms {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%  {r['Name'][:100]}")
PY
