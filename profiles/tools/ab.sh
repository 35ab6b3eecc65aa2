#!/bin/bash
# A/B of library variants on one box: profiles/tools/ab.sh "bench args" name1 name2 ...  ("base" = the regular build)
args=$1; shift
for name in "$@"; do
  lib=""; [ "$name" != base ] && lib=$PWD/pheniqs_amd/variants/lib$name.so
  for rep in 1 2; do
    PHQ_LIBRARY=$lib python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['all_kernels_ms_per_step'].items()})"
  done
done
