# kernel times of the record kernels by reads per wave and output room (scratch)
cd /tmp && export TMPDIR=/tmp
for config in "64 0" "64 14000" "32 0" "32 7000" "16 3500"; do
  set -- $config
  export PHQ_FASTQ_READS_PER_WAVE=$1
  if [ $2 != 0 ]; then export PHQ_RECORD_OUTPUT_ROOM=$2; else unset PHQ_RECORD_OUTPUT_ROOM; fi
  rm -rf /tmp/prof_sw
  rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -o sw -- python3 $GRAFT_REPO_ROOT/profiles/tools/fastq_leg.py 4000000 1048576 zero > /dev/null 2>&1
  echo "rpw $1 room $2"
  python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/prof_sw/*.db')[0])
for r in db.cursor().execute("select * from top_kernels limit 12"):
    if 'record_' in r[0] or 'fastq_decode' in r[0]:
        print('   ', r[0][30:60], round(r[3], 1))
PY
done
