#!/bin/bash
# All profiles of ONE bench scene at 1920x1080 (GPU box): kernel stats, HBM traffic (two PMC passes), VALU occupancy (two PMC
# passes), merged into one roofline block per scene by tools/scene_roofline.py.  The program comes directly after `--`.
# usage: tools/profile_scene.sh <tag> <scene> <spp>   -> gpurun_out/<tag>_<scene>x<spp>_*  (copy what should be judged into profiles/)
set -e
tag=$1; scene=$2; spp=$3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=gpurun_out/${tag}_${scene}x${spp}
B="python3 bench.py --scene $scene --spp $spp --res 1920 1080 --steps 20 --warmup 3 --no-cpu-baseline --side-steps 0"
P="$B --prewarm-ms 0 --no-boundary --no-alone"   # the counter passes: a known number of frames (3 + 20 + 1, all FP64), no clock pre-warm, Colour.Zero written every frame
# the two traffic passes run with FT_OPTS=zero_fill_skip=0: k_resolve then writes every pixel every frame and its bytes are known exactly (the FETCH_SIZE calibration rests on them)
# kernel durations: with the frame pipeline off (FT_OPTS below), every kernel alone on the device - queued frames overlap on two main streams and stretch one another's launches
SERIAL=mains=1,classify_ahead=0,resolve_aside=0
FT_OPTS=$SERIAL rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_stats -- $B > ${T}_bench_under_stats.json 2>/dev/null
cp $(ls ${T}_stats/*/*kernel_stats.csv | head -1) ${T}_kernel_stats.csv
FT_OPTS=zero_fill_skip=0 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d ${T}_fetch -- $P > /dev/null 2>&1
FT_OPTS=zero_fill_skip=0 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d ${T}_write -- $P > /dev/null 2>&1
ACTIVE_PIXELS=$(python3 -c "import json,sys; d=json.loads(open('${T}_bench_under_stats.json').read().strip().splitlines()[-1]); r=d['rays_per_frame']; print((r['primary_listed_all_ranks'] - r['primary_never_generated_all_ranks']) // $spp)")
python3 tools/pmc_traffic.py ${T}_fetch ${T}_write --pixels 2073600 --spp $spp --frames 24 --f64-frames 24 --active-pixels $ACTIVE_PIXELS > ${T}_pmc_traffic.json
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d ${T}_va -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d ${T}_vb -- $P > /dev/null 2>&1
python3 tools/pmc_valu.py ${T}_va ${T}_vb > ${T}_pmc_valu.json
FT_OPTS=$SERIAL $B > ${T}_bench.json 2>/dev/null          # the same command without the profiler, pipeline off: the kernel times the roofline block quotes
$B > ${T}_bench_queued.json 2>/dev/null                  # ... and as it runs by default: the frame period
python3 tools/scene_roofline.py ${T} > ${T}_roofline.json
find ${T}_* -name "*.csv" -size +1M -delete
rm -rf ${T}_stats ${T}_fetch ${T}_write ${T}_va ${T}_vb
cat ${T}_roofline.json
