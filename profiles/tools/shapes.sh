#!/bin/bash
# One bench line per shape, kernel chain included: profiles/tools/shapes.sh "C3 S384:100000000 ..." [extra bench args]   (GPU box, repo root)
# A shape is CONFIG[:READS[:PRIORS]]; PHQ_LIBRARY picks a variant build.
shapes=$1; shift
for shape in $shapes; do
  IFS=: read config reads priors <<< "$shape"
  args="--config $config"
  [ -n "$reads" ] && args="$args --reads $reads"
  [ -n "$priors" ] && args="$args --priors $priors"
  python bench.py $args --steps 5 --warmup 2 --no-extra-legs --no-cpu-baseline --legs /tmp/legs_$$.json "$@" > /tmp/line_$$.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("/tmp/legs_$$.json"))
print("%-22s %8.3f ms  step_frac %.3f  parity %s  | %s" % ("$shape", d["ms_per_step"], d["roofline"]["step_frac"], (d.get("parity") or {}).get("posterior_worst"),
      " -> ".join("%s %.3f" % (c["kernel"].replace("pamld_", "").replace("_kernel", ""), c["ms_per_step"]) for c in d["roofline"]["chain"])))
PY
done
