# quick look at a few shapes: bash profiles/tools/scratch/quick.sh [shape ...]   (C2 C3 C5 C3p C2p C4 C4V)
shapes="$@"; [ -z "$shapes" ] && shapes="C2 C3 C5"
for c in $shapes; do
  extra=""; cfg=$c
  case $c in C3p) cfg=C3; extra="--priors estimated";; C2p) cfg=C2; extra="--priors estimated";; esac
  python bench.py --config $cfg $extra --reads ${READS:-50000000} --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],3), d['parity']['posterior_worst'], {k:round(v,3) for k,v in d['roofline']['all_kernels_ms_per_step'].items()})"
done
