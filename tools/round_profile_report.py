#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the committed summaries under profiles/:
  <out>_kernel_stats.csv / <out>.md      per-kernel time of the bench command (rocprofv3 --kernel-trace --stats)
  <out>_mfma.md                          per GEMM / attention kernel: MFMA-busy % (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024
                                         SIMDs), the box's MfmaUtil expression), LDS bank-conflict and LDS-issue-stall shares
  gemm traffic json                      via tools/pmc_traffic.py (FETCH_SIZE x 2 + WRITE_SIZE, KiB)
  <out>_hbm.md                           the HBM-bound kernels (LayerNorm fwd / bwd, CE fwd / bwd, AdamW, fp8 row quantiser, transposes): algorithmic
                                         bytes per launch (bench.py --hbm-bytes-out: every operand read once, every result written once) / the
                                         kernel's average duration in the rocprofv3 kernel trace / 8 TB/s
usage: round_profile_report.py <prof_dir> <out_prefix> <traffic_json> "<command>" [<hbm_bytes_json>] """
import collections
import csv
import glob
import os
import re
import subprocess
import sys

prof, out, traffic_json, command = sys.argv[1:5]
hbm_json = sys.argv[5] if len(sys.argv) > 5 else None
HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:80]


# ---- kernel stats
rows = list(csv.DictReader(open(glob.glob(prof + "/stats/*kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e3:.2f}",
                    f"{float(r['MinNs'])/1e3:.2f}", f"{float(r['MaxNs'])/1e3:.2f}", f"{float(r['Percentage']):.2f}"])
with open(out + ".md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\nCommand (on the MI355X box): `{command}`\n(raw per-kernel table: "
            f"`{os.path.basename(out)}_kernel_stats.csv`; the CLIP tower runs on a side stream beside the LM forward, so durations overlap; the "
            "`at::native::*` rows are the one-time random initialisation of the weights).\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:28]:
        f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |\n")

# ---- HBM-bound kernels: algorithmic bytes (bench.py's byte model of the same step) over rocprof's kernel durations
if hbm_json and os.path.exists(hbm_json):
    import json
    model = json.load(open(hbm_json))
    steps = None
    with open(out + "_hbm.md", "w") as f:
        f.write(f"# HBM-bound kernels of one step: algorithmic bytes / rocprofv3 average kernel duration / 8 TB/s\n\nCommand: `{command}` under "
                "`rocprofv3 --kernel-trace --stats` (durations); bytes per launch from `bench.py --hbm-bytes-out` on the same workload "
                f"({model['workload']} {model['dtype']}): every operand read once, every result written once.  An op that runs as several kernels "
                "(CE forward: row pass + reduction) is priced against the sum of their time.\n\n"
                "| op | kernels matched | launches / step | bytes / launch | avg kernel us | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|\n")
        for op, m in model["ops"].items():
            hit = [r for r in rows if m["kernel_pattern"] in r["Name"]]
            if not hit:
                continue
            total_ns = sum(float(r["TotalDurationNs"]) for r in hit)
            counts = {int(r["Calls"]) for r in hit}
            # several kernels with EQUAL call counts = one op launch runs all of them (CE forward: row pass + reduction); different counts =
            # template variants, one per launch (LayerNorm of two widths)
            launches = counts.pop() if (len(hit) > 1 and len(counts) == 1) else sum(int(r["Calls"]) for r in hit)
            per_launch_us = total_ns / 1e3 / launches
            bytes_per = m["bytes_per_step"] / m["launches_per_step"]
            gbs = bytes_per / per_launch_us / 1e3
            names = ", ".join(sorted({short(r["Name"])[:40] for r in hit}))
            f.write(f"| `eavqa_{op}` | `{names}` | {m['launches_per_step']} | {bytes_per / 1e6:.2f} MB | {per_launch_us:.2f} | {gbs:.0f} | {gbs / 8000:.3f} |\n")

# ---- MFMA busy
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(prof + "/mfma/*counter_collection.csv")[0])):
    n = short(r["Kernel_Name"])
    if "gemm" not in n and "attn" not in n:
        continue
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "_mfma.md", "w") as f:
    f.write(f"# MFMA-busy and LDS counters per kernel (separate rocprofv3 --pmc pass)\n\nCommand: `{command}` under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
            "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES` (counters serialise kernels and lower the "
            "clock: ratios, not times).\nMFMA busy % = sum SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) - the `MfmaUtil` expression of "
            "the box's counter list; LDS columns are shares of SQ_BUSY_CYCLES.\n\n| kernel | dispatches | MFMA busy % | LDS bank conflict % | LDS issue stall % |\n|---|---|---|---|---|\n")
    tot_busy = tot_active = 0.0
    for n, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        busy, act = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), sum(d.get("GRBM_GUI_ACTIVE", [1]))
        sq = max(sum(d.get("SQ_BUSY_CYCLES", [1])), 1.0)
        if "gemm" in n:
            tot_busy += busy; tot_active += act
        f.write(f"| `{n}` | {len(d.get('GRBM_GUI_ACTIVE', []))} | {100*busy/(act/8*1024):.1f} | {100*sum(d.get('SQ_LDS_BANK_CONFLICT',[0]))/sq:.2f} | "
                f"{100*sum(d.get('SQ_WAIT_INST_LDS',[0]))/sq:.1f} |\n")
    f.write(f"\nAll GEMM dispatches together: MFMA busy {100*tot_busy/(tot_active/8*1024):.1f} % of the time a GEMM kernel is on the chip.\n")

subprocess.run([sys.executable, os.path.join(HERE, "pmc_traffic.py"), prof + "/fetch", prof + "/write", traffic_json], check=True, stdout=subprocess.DEVNULL)
print(open(out + "_mfma.md").read()[-1800:])
