#!/bin/bash
# HBM-side traffic of the decode attention kernel (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE, own pass): do the 160-byte head slices of the
# K / V cache cost whole extra lines?
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/attn_decode_probe.py 320 24 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/dct
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d /tmp/dct -o run --output-format csv -- python3 $R/tools/attn_decode_probe.py 320 8 > /tmp/dct.log 2>&1 || { tail -5 /tmp/dct.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/dct/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "attn_decode" not in k: continue
    agg[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); agg[k[:70]]["n_" + r["Counter_Name"]] += 1
for key, c in agg.items():
    n = c["n_FETCH_SIZE"] or 1
    print(f"{key}: launches {int(n)}  FETCH_SIZE x2 {c['FETCH_SIZE'] / n * 2 * 1024 / 1e6:.1f} MB  WRITE_SIZE {c['WRITE_SIZE'] / n * 1024 / 1e6:.2f} MB per launch")
PY
