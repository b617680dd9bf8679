#!/bin/bash
# All profiles of the default bench command for one round (GPU box): kernel stats, HBM traffic (two PMC passes), VALU occupancy
# (two PMC passes).  usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
set -e
tag=$1
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --side-steps 0"
P="$B --prewarm-ms 0 --no-boundary --no-alone"   # the counter passes: a known number of frames (3 + 20 + 1, all FP64), no clock pre-warm, Colour.Zero written every frame
# the two traffic passes run with FT_OPTS=zero_fill_skip=0: k_resolve then writes every pixel every frame and its bytes are known exactly (the FETCH_SIZE calibration rests on them)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $B > gpurun_out/${tag}_bench_under_stats.json 2>/dev/null
cp $(ls gpurun_out/${tag}_stats/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv
python3 tools/trace_percentiles.py gpurun_out/${tag}_stats > gpurun_out/${tag}_kernel_percentiles.txt     # frames are pipelined: overlapped launches stretch one another (median vs mean)
# the same command with the frame pipeline off: every kernel alone on the device (what bench.py reports as roofline.alone)
FT_OPTS=mains=1,classify_ahead=0,resolve_aside=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_serial -- $B > /dev/null 2>&1
cp $(ls gpurun_out/${tag}_stats_serial/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_serial_kernel_stats.csv
python3 tools/trace_percentiles.py gpurun_out/${tag}_stats_serial >> gpurun_out/${tag}_kernel_percentiles.txt
FT_OPTS=zero_fill_skip=0 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch -- $P > /dev/null 2>&1
FT_OPTS=zero_fill_skip=0 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/${tag}_write -- $P > /dev/null 2>&1
ACTIVE_PIXELS=$(python3 -c "import json,sys; d=json.loads(open('gpurun_out/${tag}_bench_under_stats.json').read().strip().splitlines()[-1]); r=d['rays_per_frame']; print((r['primary_listed_all_ranks'] - r['primary_never_generated_all_ranks']) // 16)")
# frames the command renders: 3 warm-up + 20 timed + 1 for the fetch + 2 x 11 boundary frames (f64 frames only count for k_resolve's FP64 writes: 3+20+1+11 = 35; the 11 RGBA8 frames write 4 B)
python3 tools/pmc_traffic.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write --pixels 2073600 --spp 16 --frames 24 --f64-frames 24 --active-pixels $ACTIVE_PIXELS > gpurun_out/${tag}_pmc_traffic.json
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d gpurun_out/${tag}_va -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/${tag}_vb -- $P > /dev/null 2>&1
python3 tools/pmc_valu.py gpurun_out/${tag}_va gpurun_out/${tag}_vb > gpurun_out/${tag}_pmc_valu.json
find gpurun_out/${tag}_* -name "*.csv" -size +1M -delete
cat gpurun_out/${tag}_kernel_stats.csv | cut -c1-150; cat gpurun_out/${tag}_pmc_valu.json
