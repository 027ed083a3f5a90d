#!/bin/bash
# L2 hit rate of the 256 x 256 kernels on the CLIP-tower FFN-up shape (rocprofv3 --pmc, own pass, kernel-trace only)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 2; do
  rm -rf /tmp/l2p$mode
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d /tmp/l2p$mode -o run --output-format csv -- python3 $R/tools/gemm_bench.py --big $mode --iters 5 --only "vitL fc1,vitL fc2,kslope 4096" > /tmp/l2p$mode.log 2>&1 || { tail -5 /tmp/l2p$mode.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/l2p$mode/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gemm_bf16" not in k: continue
    key = (k[:60], r.get("Grid_Size"))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); agg[key]["n_" + r["Counter_Name"]] += 1
for key, c in agg.items():
    n = c["n_TCC_HIT_sum"] or 1
    hit, miss, req, ea = c["TCC_HIT_sum"] / n, c["TCC_MISS_sum"] / n, c["TCC_REQ_sum"] / n, c["TCC_EA0_RDREQ_sum"] / n
    print(f"mode $mode {key}: per launch hit {hit:.3e} miss {miss:.3e} req {req:.3e} ea_rdreq {ea:.3e}  hit rate {hit / max(hit + miss, 1):.3f}")
PY
done
