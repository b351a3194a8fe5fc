#!/bin/bash
# A/B builds of libnsof.so: recompile ONE translation unit with extra -D flags and link it with the regular objects.
#   scripts/build_variant.sh <name> <source.hip> "<extra flags>"   ->  neuromorphic-spatiotemporal-optical-flow_amd/nsof/libnsof_<name>.so
# Use with NSOF_LIB=<that file> (nsof/_lib.py) for timing runs; never shipped as the product library.
set -e
cd "$(dirname "$0")/../neuromorphic-spatiotemporal-optical-flow_amd"
name=$1; src=$2; flags=$3
make -s
mkdir -p build/var_$name
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-value \
    -Wno-unused-result -DNSOF_AB -I../include -Icsrc $flags -c csrc/$base.hip -o build/var_$name/$base.o
objs=""
for o in build/*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs build/var_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o nsof/libnsof_$name.so $objs
echo nsof/libnsof_$name.so
