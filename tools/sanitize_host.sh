#!/bin/bash
# AddressSanitizer + UBSan over the HOST side of the product (scene graph, flattening, BSP / BVH builders, C-ABI argument handling,
# scene / PLY / image parsers): builds sanitized copies of both libraries under build/asan (g++; the device objects are linked as they
# are) and runs the CPU test suite and a flatten of every scene in scenes/ against them.  CPU only - GPU sanitizers are not available.
#   tools/sanitize_host.sh        -> prints the sanitizer findings (none expected) and the pytest summary
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
make -s -C functracer_amd/csrc ft_kernels.o ft_bvh.o
mkdir -p build/asan
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -std=c++17 -fPIC -ffp-contract=off"
g++ $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c functracer_amd/csrc/ft_capi.cpp -o build/asan/ft_capi.o
g++ $SAN -c functracer_amd/csrc/ft_scene.cpp -o build/asan/ft_scene.o
g++ $SAN -shared -o build/asan/libfunctracer_hip.so build/asan/ft_capi.o build/asan/ft_scene.o functracer_amd/csrc/ft_kernels.o functracer_amd/csrc/ft_bvh.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
(cd functracer_amd/host && g++ $SAN -shared -o "$ROOT/build/asan/libfunctracer_host.so" SceneParser.cpp ImageLoader.cpp host_api.cpp -lz)
export FT_HIP_LIB=$ROOT/build/asan/libfunctracer_hip.so FT_HOST_LIB=$ROOT/build/asan/libfunctracer_host.so
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
python3 -m pytest tests -q -s -m "not gpu" -p no:cacheprovider > build/asan/run.log 2>&1 || true
python3 - >> build/asan/run.log 2>&1 <<'PY'
import os
import functracer_amd as ft
c = ft.Context(host_only=True)
for n in sorted(os.listdir("scenes")):
    if n.endswith(".scene"):
        ft.parse_scene_file("scenes/" + n).lower(c)
        print("flattened", n, c.scene_info()["program_words"], "program words")
PY
echo "sanitizer findings: $(grep -c 'runtime error\|AddressSanitizer' build/asan/run.log || true)"
grep 'runtime error\|AddressSanitizer' build/asan/run.log | sort | uniq -c || true
grep -E "passed|failed" build/asan/run.log | tail -1
