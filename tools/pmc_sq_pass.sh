#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out
root=$PWD
common="--cpu-seconds 0 --rl-seconds 0 --scale-steps 0 --scenario gen_100x100 --extra-flows 33000 --profile-steps 0 --steps 60"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $out/sq1 -o bench -- python $root/bench.py $common > $out/sq1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $out/sq2 -o bench -- python $root/bench.py $common > $out/sq2.log 2>&1
cd $root
for d in sq1 sq2; do
  db=$(find $out/$d -name "*.db" | head -1)
  python tools/pmc_sq_summary.py $db 50 > $out/r05_gen_100x100_pmc_$d.txt 2>> $out/$d.log
  cat $out/r05_gen_100x100_pmc_$d.txt | grep -v "amd_rocclr\|k_init\|kr_reset\|device_spin"
  rm -rf $out/$d
done
