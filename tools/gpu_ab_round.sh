#!/bin/bash
# this tree against the checkout under _ab_old/ (git worktree of the previous round's last commit, built), the default bench of both on ONE box
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do
  (cd _ab_old && timeout -k 10 500 python3 bench.py --cpu-baseline-samples 0 2>/dev/null > ../gpurun_out/ab_round_old_$i.json) || exit 1
  timeout -k 10 500 python3 bench.py --cpu-baseline-samples 0 2>/dev/null > gpurun_out/ab_round_new_$i.json || exit 1
done
python3 - <<'PY'
import json
for n in ("old_1", "new_1", "old_2", "new_2"):
    d = json.loads(open(f"gpurun_out/ab_round_{n}.json").read().strip().splitlines()[-1])
    print(n, "cfg2", d["value"], [(e["config"]["workload"][:14], e["value"]) for e in d["extra"]])
PY
