#!/bin/bash
# A variant build of the library for A/B runs on one GPU box: one translation unit recompiled with extra flags, linked with
# the objects of the regular build.   usage: profiles/tools/variant.sh NAME UNIT.hip [-DFLAG ...]   ->  pheniqs_amd/variants/libNAME.so
# run with PHQ_LIBRARY=pheniqs_amd/variants/libNAME.so python bench.py ...
set -e
name=$1; unit=$2; shift 2
cd "$(dirname "$0")/../../pheniqs_amd/csrc"
mkdir -p ../variants build
stem=${unit%.hip}
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-result --offload-arch=gfx950 "$@" -c $unit -o build/variant_${name}_${stem}.o
objects=$(ls build/*.o | grep -v "variant_" | grep -v "build/${stem}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/lib${name}.so $objects build/variant_${name}_${stem}.o -ldl -lz -lpthread
echo built pheniqs_amd/variants/lib${name}.so
