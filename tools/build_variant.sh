#!/bin/bash
# Build a tuning variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>  ->  timbre-trap_amd/lib/libttrap_<name>.so
# (select at run time with TTRAP_LIB=libttrap_<name>.so)
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
src=$root/timbre-trap_amd/csrc
obj=/tmp/ttrap_variant_$name
mkdir -p $obj
pids=()
for f in $src/*.hip; do
  extra=""; [ "$(basename $f)" = cqt.hip ] && extra="-fno-slp-vectorize"      # as timbre_trap/_hip.py EXTRA_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -c $f -o $obj/$(basename $f .hip).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/timbre-trap_amd/lib/libttrap_$name.so $obj/*.o
ls -la $root/timbre-trap_amd/lib/libttrap_$name.so
