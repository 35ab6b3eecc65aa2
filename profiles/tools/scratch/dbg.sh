for dbg in 0 1 2 3 4 8 12; do
PHQ_DEBUG=$dbg python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('debug $dbg', {k: round(v,3) for k,v in d['roofline']['all_kernels_ms_per_step'].items()})"
done
