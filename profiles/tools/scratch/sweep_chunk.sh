for chunk in 1048576 2097152 4194304 8388608; do
  echo "chunk $chunk"
  PHQ_HOST_CHUNK=$chunk python bench.py --steps 2 --warmup 1 --no-parity --no-cpu-baseline --fastq 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['staged']
print({k:round(v['Mreads/s']) for k,v in d.items() if isinstance(v,dict)})"
done
