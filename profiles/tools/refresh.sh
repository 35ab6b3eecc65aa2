#!/bin/bash
# The profile set of a round, taken on ONE build: bash profiles/tools/refresh.sh r04 [shape ...]   (GPU box, repo root).
# Per workload: a kernel trace (--kernel-trace --stats) and two counter passes of their own (--pmc FETCH_SIZE, --pmc
# WRITE_SIZE: MI355X_MICROARCH.md's HBM recipe), each of the same bench.py command; then profiles/summarize.py folds
# them into profiles/rNN_traffic.json, stamped with the build (phq_build_id) and the commit.
round=${1:-r04}; shift
export TMPDIR=/tmp
declare -A ARGS=(
  [C2_pamld]=""
  [C2_pamld_literal]="--mode literal"
  [C2_mdd]="--algorithm mdd"
  [C2_pamld_estimated_priors]="--priors estimated"
  [C3_pamld]="--config C3"
  [C3_pamld_estimated_priors]="--config C3 --reads 50000000 --priors estimated"
  [C4_pamld]="--config C4 --reads 50000000"
  [C4V_pamld]="--config C4V --reads 50000000"
  [C5_pamld]="--config C5 --reads 50000000"
  [H94_pamld]="--config H94"
  [H94A_pamld]="--config H94A"
  [S384_pamld]="--config S384"
  [U384_pamld]="--config U384"
  [L12_pamld]="--config L12"
)
shapes=("$@"); [ ${#shapes[@]} -eq 0 ] && shapes=(C2_pamld C2_pamld_literal C2_mdd C2_pamld_estimated_priors C3_pamld C3_pamld_estimated_priors C4_pamld C4V_pamld C5_pamld H94_pamld H94A_pamld S384_pamld U384_pamld L12_pamld)
out=gpurun_out/prof_$round
for shape in "${shapes[@]}"; do
  a="${ARGS[$shape]}"
  d=$out/$shape; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 bench.py $a --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extra-legs --legs $d/legs.json > $d/trace.log 2>&1
  grep '^{"metric"' $d/trace.log | tail -1 > $d/bench.json
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --legs $d/fetch_legs.json > $d/fetch.log 2>&1
  grep '^{"metric"' $d/fetch.log | tail -1 > $d/fetch.json
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -- python3 bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --legs $d/write_legs.json > $d/write.log 2>&1
  grep '^{"metric"' $d/write.log | tail -1 > $d/write.json
  stats=$(ls -t $d/trace/*/*_kernel_stats.csv | head -1)
  (head -1 $stats; grep "phq::" $stats) > gpurun_out/${round}_${shape}_kernel_stats.csv
  cp $d/legs.json gpurun_out/${round}_bench_${shape}.json     # the full record (chain included); bench.json is the contract line
  echo "$shape: $(grep -o '"ms_per_step": [0-9.]*' $d/bench.json)"
done
python3 profiles/summarize.py $out gpurun_out/${round}_traffic.json "${shapes[@]}"
# the kernels either side of the path (rows f1, f2): 8 M reads of FASTQ text -> SAM records, in rounds of 1 M reads
d=$out/fastq_to_sam; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 profiles/tools/fastq_leg.py 8000000 1048576 zero null > $d/trace.log 2>&1
stats=$(ls -t $d/trace/*/*_kernel_stats.csv | head -1)
(head -1 $stats; grep -E "phq::|rocclr" $stats) > gpurun_out/${round}_fastq_to_sam_kernel_stats.csv
grep -E "^(zero|null) " $d/trace.log > gpurun_out/${round}_fastq_to_sam_rates.txt
