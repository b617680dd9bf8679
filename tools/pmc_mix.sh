#!/bin/bash
# Instruction-mix / occupancy counters of one bench scene, three separate PMC passes (kernel-trace only, as the pool requires).
# usage: tools/pmc_mix.sh <scene> <tag> [spp]   -> gpurun_out/pmc_<tag>_{a,b,c}/, summary in gpurun_out/pmc_<tag>.txt
set -e
scene=$1; tag=$2; spp=${3:-16}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
common="--kernel-trace --output-format csv"
rocprofv3 $common --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d gpurun_out/pmc_${tag}_a -- python3 bench.py --scene $scene --spp $spp --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 $common --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d gpurun_out/pmc_${tag}_b -- python3 bench.py --scene $scene --spp $spp --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 $common --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU -d gpurun_out/pmc_${tag}_c -- python3 bench.py --scene $scene --spp $spp --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "pass c failed (counter names?)"
for p in a b c; do python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_$p; done > gpurun_out/pmc_${tag}.txt
find gpurun_out/pmc_${tag}_* -name "*.csv" -size +2M -delete
cat gpurun_out/pmc_${tag}.txt
