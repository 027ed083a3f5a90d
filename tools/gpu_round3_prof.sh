#!/bin/bash
# round-3 profiles: cfg2 bf16 (headline) and cfg5 fp8 (kernel stats, FETCH / WRITE, MFMA + LDS counters, HBM table) and the few-shot phases.
# The raw rocprofv3 output (hundreds of MiB of per-dispatch CSV) is summarised ON the box; only the summaries travel back.
cd ${GRAFT_REPO_ROOT:-.}
S=gpurun_out/r3_profiles; mkdir -p $S
for W in "cfg2 bf16" "cfg5 fp8"; do
  set -- $W
  tag=r3_$1
  extra=""; [ "$1" != "cfg2" ] && extra="--workload $1 --dtype $2"
  bash tools/profile_round.sh $tag $extra > gpurun_out/prof_$tag.log 2>&1; echo "$tag profiled rc=$?"
  python3 tools/round_profile_report.py gpurun_out/prof_$tag $S/round3_bench_$1_$2 $S/round3_gemm_traffic_$1_$2.json \
      "python3 bench.py --steps 10 --warmup 3 --cpu-baseline-samples 0 --no-roofline --no-fewshot --no-extra-train $extra" gpurun_out/prof_$tag/hbm_bytes.json > gpurun_out/report_$tag.log 2>&1
  echo "$tag report rc=$?"; tail -3 gpurun_out/report_$tag.log
  cp gpurun_out/prof_$tag/plain.json $S/plain_$1_$2.json 2>/dev/null
  rm -rf gpurun_out/prof_$tag
done
bash tools/gpu_prof_fewshot.sh r3 > $S/round3_fewshot_stats.txt 2>&1; echo "fewshot rc=$?"
cp gpurun_out/prof_fewshot_r3/stats/*kernel_stats.csv $S/round3_fewshot_kernel_stats_raw.csv 2>/dev/null
rm -rf gpurun_out/prof_fewshot_r3
du -sh gpurun_out; ls -la $S
