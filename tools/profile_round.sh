#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      e.g. r02_v1
# kernel trace of the default bench command + FETCH_SIZE / WRITE_SIZE in separate PMC passes (MI355X_MICROARCH.md),
# summaries written under gpurun_out/<tag>_* (copy the ones to keep into profiles/).
set -u
tag=${1:-r02}
extra=${2:-}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
python bench.py --cpu-seconds 12 $extra > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o bench -- python $OLDPWD/bench.py --cpu-seconds 0 --profile-steps 0 $extra > $out/${tag}_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/${tag}_pmc_$c -o bench -- python $OLDPWD/bench.py --cpu-seconds 0 --profile-steps 0 --steps 60 $extra > $out/${tag}_pmc_$c.log 2>&1
done
cd $OLDPWD
db=$(find $out/${tag}_trace -name "*.db" | head -1)
python tools/rocpd_summary.py $db 100 > $out/${tag}_kernel_trace_bench.txt 2>> $out/${tag}_trace.log
f=$(find $out/${tag}_pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find $out/${tag}_pmc_WRITE_SIZE -name "*.db" | head -1)
python tools/pmc_summary.py $f $w 50 ${3:-bench} > $out/${tag}_pmc_hbm_traffic.json 2>> $out/${tag}_trace.log
head -12 $out/${tag}_kernel_trace_bench.txt
cat $out/${tag}_bench.json | cut -c1-1500
# the databases are large: keep only the summaries
rm -rf $out/${tag}_trace $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE
