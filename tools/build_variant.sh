#!/bin/bash
# diagnostic: build an alternative libfastf_amd.so with extra -D flags into build/<name>/
# usage: tools/build_variant.sh <name> [-DFOO=1 ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$ROOT/build/$name
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/fastf_amd/csrc -Wall -Wno-pass-failed "$@" \
    -c $ROOT/fastf_amd/csrc/umi_engine.hip -o $out/umi_engine.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libfastf_amd.so $out/umi_engine.o $ROOT/build/obj/mt_jump.o $ROOT/build/obj/host_prims.o $ROOT/build/obj/host_io.o \
    $ROOT/build/obj/bam2db_main.o $ROOT/build/obj/tag_cmds.o $ROOT/build/obj/inflate_fast.o $ROOT/build/obj/crc32_fast.o $ROOT/build/obj/deflate_fast.o -lz -lpthread -ldl -Wl,-rpath,/opt/rocm/lib
echo built $out/libfastf_amd.so
