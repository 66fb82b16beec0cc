#!/bin/bash
# builds tools/micro/fb_<tag> variants of fused_bench.hip: args "tag NCH RT NT NX ACT LN BWD [extra flags]"
set -e
cd "$(dirname "$0")/../.."
build() { tag=$1; shift; nch=$1 rt=$2 nt=$3 nx=$4 act=$5 ln=$6 bwd=$7; shift 7
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I pinns-rl-pde_amd/csrc \
    -DB_NCH=$nch -DB_RT=$rt -DB_NT=$nt -DB_NX=$nx -DB_ACT=$act -DB_LN=$ln -DB_BWD=$bwd "$@" tools/micro/fused_bench.hip -o tools/micro/fb_$tag; }
build c4f 8 8 1 3 1 0 0 -DPINN_FSTAMPS &
build c4b 8 8 1 3 1 0 1 -DPINN_FSTAMPS &
build c3f 8 8 1 2 0 1 0 -DPINN_FSTAMPS -DB_SKIP &
build c3f2 8 8 1 2 0 1 0 -DPINN_FSTAMPS &
build c3b 8 8 1 2 0 1 1 -DPINN_FSTAMPS -DB_SKIP &
build c3b2 8 8 1 2 0 1 1 -DPINN_FSTAMPS &
wait
