#!/bin/bash
# Collect a round's profiles on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [workload] [extra bench.py flags]      e.g. r03_v2 bench | r03_v2 gen_100x100 | r03_v2 grid_6x6
#   (extra flags, e.g. "--cfx layout=ring,ringLanesPerWave=30000", select a non-default form of the step; say so in the tag)
# kernel trace of the bench command on that workload + FETCH_SIZE / WRITE_SIZE in separate PMC passes (MI355X_MICROARCH.md),
# summaries written under gpurun_out/<tag>_<workload>_* (copy the ones to keep into profiles/).  The full default bench line
# (all legs) is a separate call: `python bench.py > gpurun_out/<tag>_bench.json`.
set -u
tag=${1:-r04}
wl=${2:-bench}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
case $wl in
  bench) extra="" ;;
  gen_100x100) extra="--scenario gen_100x100 --extra-flows 33000" ;;
  *) extra="--scenario $wl --extra-flows 0" ;;
esac
common="--cpu-seconds 0 --rl-seconds 0 --scale-steps 0 $extra ${3:-}"
pre=${tag}_${wl}
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/${pre}_trace -o bench -- python $OLDPWD/bench.py $common --profile-steps 0 > $out/${pre}_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/${pre}_pmc_$c -o bench -- python $OLDPWD/bench.py $common --profile-steps 0 --steps 60 > $out/${pre}_pmc_$c.log 2>&1
done
cd $OLDPWD
db=$(find $out/${pre}_trace -name "*.db" | head -1)
python tools/rocpd_summary.py $db 100 > $out/${pre}_kernel_trace.txt 2>> $out/${pre}_trace.log
f=$(find $out/${pre}_pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find $out/${pre}_pmc_WRITE_SIZE -name "*.db" | head -1)
python tools/pmc_summary.py $f $w 50 $wl > $out/${pre}_pmc_hbm_traffic.json 2>> $out/${pre}_trace.log
head -14 $out/${pre}_kernel_trace.txt
python - <<E
import json
d = json.load(open("$out/${pre}_pmc_hbm_traffic.json"))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_fetch_doubled"])[:8]:
    print("%-60s raw %8.2f MB   fetch-doubled %8.2f MB" % (k[:60], v["hbm_bytes_raw"] / 1e6, v["hbm_bytes_fetch_doubled"] / 1e6))
E
# the databases are large: keep only the summaries
rm -rf $out/${pre}_trace $out/${pre}_pmc_FETCH_SIZE $out/${pre}_pmc_WRITE_SIZE
