#!/bin/bash
# bash scripts/build_variant.sh NAME "-DFOO=1 ..." : an A/B build of the library at variants/NAME.so (use with RT_LIB)
set -e
NAME=$1; DEFS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/variants/build_$NAME; mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result $DEFS"
cd $ROOT/raytracer_amd
/opt/rocm/bin/hipcc $F ${KFLAGS--mllvm -amdgpu-sched-strategy=max-ilp} -c csrc/rt_core.hip -o $B/rt_core.o &
/opt/rocm/bin/hipcc $F -c csrc/rt_lbvh.hip -o $B/rt_lbvh.o &
/opt/rocm/bin/hipcc $F -x hip -c csrc/rt_bvh.cpp -o $B/rt_bvh.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/$NAME.so $B/rt_core.o $B/rt_lbvh.o $B/rt_bvh.o build/rt_host.o build/rt_host_c.o
echo built $NAME
