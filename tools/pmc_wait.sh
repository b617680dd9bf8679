#!/bin/bash
# Where the waves of a scene's kernels wait: issue / wait shares and instruction- and scalar-cache hit rates (two PMC passes).
# usage: tools/pmc_wait.sh <scene> <spp> <tag>  -> gpurun_out/<tag>_wait.txt
set -e
scene=$1; spp=$2; tag=$3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B="python3 bench.py --scene $scene --spp $spp --res 1920 1080 --steps 5 --warmup 1 --no-cpu-baseline --side-steps 0 --prewarm-ms 0"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d gpurun_out/${tag}_wa -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_VMEM -d gpurun_out/${tag}_wb -- $B > /dev/null 2>&1 || echo "pass b failed"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM -d gpurun_out/${tag}_wc -- $B > /dev/null 2>&1 || echo "pass c failed"
for p in wa wb wc; do python3 tools/pmc_summary.py gpurun_out/${tag}_$p; done > gpurun_out/${tag}_wait.txt
rm -rf gpurun_out/${tag}_wa gpurun_out/${tag}_wb gpurun_out/${tag}_wc
cat gpurun_out/${tag}_wait.txt
