for c in C2 C3 C5; do python bench.py --config $c --reads 50000000 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:4], round(d['ms_per_step'],3), d['parity']['posterior_worst'], {k:round(v,3) for k,v in d['roofline']['all_kernels_ms_per_step'].items()})"; done
