#!/bin/bash
# round-3 call D: direct epilogue - parity of every GEMM test, then A / B timings
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py -q -x -k "gemm or fp8" > gpurun_out/r3d_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r3d_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
{
for sel in "--k64 7" "--k64 13"; do echo "=== 128x80 $sel"; python tools/gemm_bench.py $sel --only "packed proj,packed fc2,packed da"; python tools/gemm_bench.py $sel --epi res32 --only "packed proj,packed fc2"; done
for sel in "--k64 5" "--k64 14"; do echo "=== 256x128 $sel"; python tools/gemm_bench.py $sel --only "packed qkv"; done
for sel in "--k64 6" "--k64 15"; do echo "=== 256x160 $sel"; python tools/gemm_bench.py $sel --only "packed fc1"; python tools/gemm_bench.py $sel --epi fc1 --only "packed fc1"; python tools/gemm_bench.py $sel --epi bwd --only "packed fc1"; done
for sel in "--big 2 --epi-lds" "--big 2"; do echo "=== big $sel"; python tools/gemm_bench.py $sel --only "vitL qkv,vitL proj,vitL fc1,vitL fc2,square 4k"; python tools/gemm_bench.py $sel --epi quick --only "vitL fc1"; python tools/gemm_bench.py $sel --epi res16 --only "vitL proj,vitL fc2"; python tools/gemm_bench.py $sel --epi res32 --only "vitL proj,vitL fc2"; done
} > gpurun_out/r3d_bench.log 2>&1
cat gpurun_out/r3d_bench.log
