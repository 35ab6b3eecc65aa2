# refresh the default mode profile set of a round: bash profiles/tools/refresh_fast.sh r01   (on the GPU box, from the repo root)
round=${1:-r01}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/prof_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/prof_write.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write 100000000 gpurun_out/${round}_summary.json
cp $(ls -t gpurun_out/prof_trace/*/*_kernel_stats.csv | head -1) gpurun_out/${round}_fast_kernel_stats.csv
cp $(ls -t gpurun_out/prof_fetch/*/*_counter_collection.csv | head -1) gpurun_out/${round}_fast_pmc_fetch_size.csv
cp $(ls -t gpurun_out/prof_write/*/*_counter_collection.csv | head -1) gpurun_out/${round}_fast_pmc_write_size.csv
cp gpurun_out/${round}_summary.json profiles/${round}_summary.json      # bench.py quotes traffic from it
python3 bench.py --steps 10 --warmup 3 > gpurun_out/${round}_bench_fast.json 2> gpurun_out/bench_fast.err
tail -1 gpurun_out/${round}_bench_fast.json | cut -c1-1500
