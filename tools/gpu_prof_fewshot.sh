#!/bin/bash
# rocprofv3 kernel stats of the few-shot phases (tools/fewshot_profile.py); usage: gpu_prof_fewshot.sh <tag>
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_fewshot_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/tools/fewshot_profile.py > $OUT/stats.log 2>&1
echo "rc=$?"; cat $OUT/stats.log | tail -5
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/stats/*kernel_stats.csv")[0])))
for r in rows[:24]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):5.1f}%")
PY
