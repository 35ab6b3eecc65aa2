cd /root/repo; export TMPDIR=/tmp
# usage: bash profiles/tools/pmc_sq.sh [bench args]   (GPU box, repo root); averages per launch on stdout
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_CYCLES" "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > gpurun_out/pmc_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name']
            kernel = 'stream' if 'pamld_stream_kernel' in name else 'near' if 'pamld_near_kernel' in name else 'lut' if 'pamld_lut_kernel' in name else 'exact' if 'exact' in name else None
            if kernel:
                acc[kernel + ' ' + row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items(): print(d, k, sum(v)/len(v), len(v))
PY
