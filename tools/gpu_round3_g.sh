#!/bin/bash
# tile order of the 256 x 256 kernel inside an XCD: m fastest (round 2) against column groups, interleaved in ONE process per shape set
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for rep in 1 2; do
for g in 1 3 5 7 9 13; do echo "=== rep $rep --big 2 --group-n $g"; python tools/gemm_bench.py --big 2 --group-n $g --only "vitL qkv,vitL proj,vitL fc1,vitL fc2,vitL64,prefill,square" 2>/dev/null; done
done
} > gpurun_out/r3g_order.log 2>&1
python - <<'PY'
import re, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
g = None
for line in open("gpurun_out/r3g_order.log"):
    m = re.match(r"=== rep \d+ --big 2 --group-n (\d+)", line)
    if m: g = int(m.group(1)); continue
    m = re.match(r"(\S.*?)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+([\d.]+) us", line)
    if m: res[m.group(1)][g].append(float(m.group(5)))
gs = sorted({g for v in res.values() for g in v})
print("shape".ljust(16), *[f"g={g - 1 if g > 1 else 'm'}".rjust(9) for g in gs])
for name, v in res.items():
    print(name.ljust(16), *[f"{min(v[g]):9.1f}" for g in gs])
PY
