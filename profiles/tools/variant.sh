#!/bin/bash
# A variant build of the library for A/B runs on one GPU box: some translation units recompiled with extra flags, linked with
# the objects of the regular build.   usage: profiles/tools/variant.sh NAME "UNIT [UNIT ...]" [-DFLAG ...]   ->  pheniqs_amd/variants/libNAME.so
# (a flag that changes the streaming kernel's LDS map — PHQ_STREAM_WAVES, PHQ_STREAM_GLOBAL_RING — needs "pamld_stream.hip decoder_set.cpp":
# the host puts block addresses into the lookup table).  Run with PHQ_LIBRARY=pheniqs_amd/variants/libNAME.so python bench.py ...
set -e
name=$1; units=$2; shift 2
cd "$(dirname "$0")/../../pheniqs_amd/csrc"
mkdir -p ../variants build
objects=$(ls build/*.o | grep -v "variant_")
for unit in $units; do
  stem=${unit%.*}
  arch=""; [ "${unit##*.}" = hip ] && arch="--offload-arch=gfx950"
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-result $arch "$@" -c $unit -o build/variant_${name}_${stem}.o
  objects=$(echo "$objects" | grep -v "build/${stem}.o")
  objects="$objects build/variant_${name}_${stem}.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/lib${name}.so $objects -ldl -lz -lpthread
echo built pheniqs_amd/variants/lib${name}.so
