#!/bin/bash
# usage: tools/profile_round.sh TAG   (on the GPU box; writes gpurun_out/TAG/*)
# The measurement set of one round: the bench line, its rocprofv3 kernel statistics, the counter
# passes over the dominant kernel (two workloads), the batch / resolution sweep, the data-parallel
# shard and the full training step with its kernel statistics.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
step() { echo "== $1" >&2; }
step kernel-stats
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $O/bench_under_rocprofv3.json 2> $O/prof.err || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
step pmc-12x192x640
bash $R/tools/pmc_tile.sh $O/pmc > $O/pmc.txt 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $O/pmc --kernel photo_tile --json $O/tile_pmc.json > /dev/null
step pmc-8x384x1280
export DW_B=8 DW_H=384 DW_W=1280
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout -s KILL 90 rocprofv3 --pmc $c --output-format csv -d $O/pmc_hi/$c -- python3 $R/tools/run_tile_only.py 3 > $O/pmc_hi_$c.log 2>&1 || exit 1
done
unset DW_B DW_H DW_W
python3 $R/tools/pmc_summary.py $O/pmc_hi --kernel photo_tile --json $O/tile_pmc_8x384x1280.json > /dev/null
# the counters go into the file bench.py reads `roofline.traffic` and the pipe limit from (stamped with the
# sources' hash; the instruction-cost figure was put there before this run by tools/valu_pipe_model.py --write),
# and only then is the bench line of the round taken, so that it carries them
step traffic
python3 $R/tools/pmc_traffic.py $O/tile_pmc.json 12 192 640 --source profiles/${T}_tile_pmc.json > /dev/null || exit 1
python3 $R/tools/pmc_traffic.py $O/tile_pmc_8x384x1280.json 8 384 1280 --source profiles/${T}_tile_pmc_8x384x1280.json > /dev/null || exit 1
cp $R/profiles/traffic_latest.json $O/traffic_latest.json
step bench
timeout -k 10 200 python3 $R/bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err || exit 1
step sweep
timeout -k 10 120 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --batch 48 > $O/bench_b48.json 2>> $O/bench.err || exit 1
timeout -k 10 120 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --batch 96 > $O/bench_b96.json 2>> $O/bench.err || exit 1
timeout -k 10 120 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --batch 8 --height 384 --width 1280 > $O/bench_384x1280_b8.json 2>> $O/bench.err || exit 1
step dp
timeout -k 10 120 python3 $R/bench.py --mode dp --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_dp_n1.json 2>> $O/bench.err || exit 1
step train
timeout -k 10 300 python3 $R/bench.py --mode train --steps 30 --warmup 5 > $O/bench_train.json 2>> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --mode train --steps 30 --warmup 5 > $O/bench_train_under_rocprofv3.json 2> $O/prof_train.err || exit 1
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
rm -rf $O/prof $O/prof_train $O/pmc/*/ $O/pmc_hi
echo done
