#!/bin/bash
# The timing legs of tools/profile_scene.sh only (kernel stats with the pipeline off, the bench line with it off and on): the counter passes do not depend on the pipeline.
# usage: tools/profile_scene_times.sh <tag> <scene> <spp>
set -e
tag=$1; scene=$2; spp=$3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=gpurun_out/${tag}_${scene}x${spp}
B="python3 bench.py --scene $scene --spp $spp --res 1920 1080 --steps 20 --warmup 3 --no-cpu-baseline --side-steps 0"
SERIAL=mains=1,classify_ahead=0,resolve_aside=0
FT_OPTS=$SERIAL rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_stats -- $B > ${T}_bench_under_stats.json 2>/dev/null
cp $(ls ${T}_stats/*/*kernel_stats.csv | head -1) ${T}_kernel_stats.csv
FT_OPTS=$SERIAL $B > ${T}_bench.json 2>/dev/null
$B > ${T}_bench_queued.json 2>/dev/null
rm -rf ${T}_stats
