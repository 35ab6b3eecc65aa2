# kernel traces and bench lines of the other shapes: bash profiles/tools/refresh_shapes.sh r01   (GPU box, repo root)
round=${1:-r01}
export TMPDIR=/tmp
run() {   # name, bench arguments
  name=$1; shift
  rm -rf gpurun_out/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/prof_$name.log 2>&1
  stats=$(ls -t gpurun_out/prof_$name/*/*_kernel_stats.csv | head -1)
  (head -1 $stats; grep "phq::" $stats) > gpurun_out/${round}_${name}_kernel_stats.csv
  grep '^{"metric"' gpurun_out/prof_$name.log | tail -1 > gpurun_out/${round}_bench_${name}.json
  cut -c1-120 gpurun_out/${round}_${name}_kernel_stats.csv | head -6
  grep -o '"value": [0-9.]*' gpurun_out/${round}_bench_${name}.json
}
run mdd --algorithm mdd
run c3 --config C3 --reads 50000000
run c4 --config C4 --reads 50000000
run c5 --config C5 --reads 50000000
run literal --mode literal
run c4v --config C4V --reads 50000000
run priors --priors estimated
run c3priors --config C3 --reads 50000000 --priors estimated
