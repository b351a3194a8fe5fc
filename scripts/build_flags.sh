#!/bin/bash
# Whole-library A/B build with extra compiler flags:  scripts/build_flags.sh <name> "<flags>"  -> nsof/libnsof_<name>.so
# (every translation unit recompiled into build_flags_<name>/; use with NSOF_LIB=... / scripts/ab_bench.sh)
set -e
cd "$(dirname "$0")/../neuromorphic-spatiotemporal-optical-flow_amd"
name=$1; flags=$2
base="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
mkdir -p build_flags_$name
pids=""
for src in csrc/*.hip; do
  b=$(basename $src .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $base $flags -I../include -Icsrc -c $src -o build_flags_$name/$b.o 2> build_flags_$name/$b.log &
  pids="$pids $!"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $base $flags -I../include -Icsrc -DNSOF_PYR_FMA -c csrc/farneback_kernels.hip -o build_flags_$name/farneback_kernels_fma.o 2> build_flags_$name/fma.log &
pids="$pids $!"
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o nsof/libnsof_$name.so build_flags_$name/*.o
rm -rf build_flags_$name
echo nsof/libnsof_$name.so
