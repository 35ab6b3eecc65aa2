#!/bin/bash
# GPU box: parity suite + one bench line; summaries under gpurun_out/.   usage: profiles/tools/check.sh tag [bench args]
tag=$1; shift
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_$tag.log | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("value %.0f %s, %.3f ms/step, kernels %s, frac %.3f" % (d["value"], d["unit"], d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"]["all_kernels_ms_per_step"].items()}, d["roofline"]["frac"]))
PY
