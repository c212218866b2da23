#!/bin/bash
# builds experiment variants of one kernel file: tools/exp/build_variants.sh k_rows_stream "NAME1:-DFLAG1 NAME2:-DFLAG2 ..."
set -e
cd "$(dirname "$0")/../../lbaudiodetective_amd/csrc"
file=$1; shift
mkdir -p ../lib/exp
objs=$(ls ../lib/obj/*.o | grep -v "/$file.o")
extra=$(make -pn 2>/dev/null | grep "^FLAGS_$file" | sed 's/.*= //')
for spec in $1; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $extra $flags -x hip -c $file.hip -o ../lib/exp/$file.$name.o &
done
wait
for spec in $1; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/lib_$name.so $objs ../lib/exp/$file.$name.o
done
ls -la ../lib/exp/*.so
