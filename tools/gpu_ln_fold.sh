#!/bin/bash
# eavqa_gemm_ln A / B (run through gpurun from the repo root): rocprofv3 kernel statistics of the cfg2 bench with the LayerNorm folded
# (EAVQA_LN_FOLD=1, the default) and with the LayerNorm kernels (=0), then the un-profiled step times of both, interleaved.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
W=${1:-cfg2}
ARGS="--workload $W --steps 10 --warmup 3 --cpu-baseline-samples 0 --no-roofline --no-fewshot --no-extra-train --no-t0"
for f in 1 0; do
  export EAVQA_LN_FOLD=$f
  rm -rf /tmp/lnf$f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/lnf$f -o run --output-format csv -- python3 $R/bench.py $ARGS > /tmp/lnf$f.log 2>&1 || { tail -5 /tmp/lnf$f.log; exit 1; }
  echo "== EAVQA_LN_FOLD=$f ($W): per step of 13"
  python3 - $f <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/lnf{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if not r["Name"].startswith("void at::") and "at::native" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
gem = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_" in r["Name"])
ln = sum(float(r["TotalDurationNs"]) for r in rows if "ln_fwd" in r["Name"])
print(f"all kernels {tot/13e6:.3f} ms   gemm kernels {gem/13e6:.3f} ms   ln_fwd {ln/13e6:.3f} ms")
for r in rows[:12]:
    print(f"{float(r['TotalDurationNs'])/13e6:8.3f} ms {int(r['Calls'])/13:7.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
done
cd $R
for i in 1 2 3; do
  for f in 1 0; do
    EAVQA_LN_FOLD=$f timeout -k 10 300 python3 bench.py --workload $W --steps 40 --warmup 10 --cpu-baseline-samples 0 --no-roofline --no-fewshot --no-extra-train --no-t0 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=$f run $i', d['value'], d['ms_per_step'])" || exit 1
  done
done
