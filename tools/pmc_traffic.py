#!/usr/bin/env python3
"""HBM traffic of the GEMM kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md
prescribes: separate passes; FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half
their bytes, so it is doubled.  Writes profiles/<tag>_gemm_traffic.json (bytes per launch, averaged over GEMM launches)."""
import collections
import csv
import glob
import json
import re
import sys

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                m = re.search(r"(gemm_\w+_kernel|eavqa_attn_mfma::\w+|\w+_kernel)", r["Kernel_Name"])
                acc[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    return acc


F, W = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
rows, tot_f, tot_w, n = [], 0.0, 0.0, 0
for k in sorted(F):
    if "gemm" not in k:
        continue
    f, w = F[k], W.get(k, [])
    rows.append(dict(kernel=k[-60:], launches=len(f), fetch_kib_avg=sum(f) / len(f), write_kib_avg=(sum(w) / len(w)) if w else None))
    tot_f += sum(f); tot_w += sum(w); n += len(f)
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f_ in ("gemm.hip", "gemm_k64.hip", "gemm_fp8.hip", "common.h"):
    with open(os.path.join(ROOT, "explicit-alignment-for-vqa-tasks_amd", "csrc", f_), "rb") as fh:
        h.update(fh.read())
# per operand format: a step of the fp8 workload launches fp8 GEMMs (frozen LM) AND bf16 GEMMs (mapper, CLIP tower)
families = {}
for fam in ("fp8", "bf16", "f32"):
    ff = sum(sum(F[k]) for k in F if "gemm_" + fam in k)
    fw = sum(sum(W.get(k, [])) for k in F if "gemm_" + fam in k)
    fn = sum(len(F[k]) for k in F if "gemm_" + fam in k)
    if fn:
        families[fam] = dict(gemm_launches=fn, bytes_per_launch=(2 * ff + fw) * 1024 / fn, read_bytes_per_launch=2 * ff * 1024 / fn,
                             write_bytes_per_launch=fw * 1024 / fn)
res = dict(gemm_source_digest=h.hexdigest()[:12], families=families, counter_units="KiB", fetch_correction="x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)",
           gemm_launches=n, bytes_per_launch=(2 * tot_f + tot_w) * 1024 / max(n, 1),
           read_bytes_per_launch=2 * tot_f * 1024 / max(n, 1), write_bytes_per_launch=tot_w * 1024 / max(n, 1), kernels=rows)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
for r in rows:
    print(r)
