#!/bin/bash
# round measurement (run on the GPU box: gpurun -- bash profiles/measure.sh): bench lines, rocprofv3 kernel stats and the four PMC passes of the default (fp16-pair) kernel
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/measure; rm -rf $O; mkdir -p $O   # (also clear the local gpurun_out/measure first: gpurun merges, it does not mirror)
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-precision --no-train"
timeout -k 10 400 python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err && echo bench ok &&
timeout -k 10 300 python3 $R/bench.py --precision f32 --no-cpu-baseline > $O/bench_n1_f32.json 2>> $O/bench_n1.err && echo f32 ok &&
timeout -k 10 300 python3 $R/bench.py --workload lego_400x400_64c --steps 10 > $O/bench_c1.json 2>> $O/bench_n1.err && echo c1 ok &&
timeout -k 10 300 python3 $R/bench.py --workload fern_1008x756_ndc_64c+128f --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench_n1.err && echo c4 ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-precision --no-train > $O/stats.log 2>&1 && echo stats ok &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1 && echo f ok &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1 && echo w ok &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1 && echo s ok &&
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- $B > $O/pmc_l2.log 2>&1 && echo l ok &&
timeout -k 10 400 python3 $R/profiles/shard_projection.py > $O/shard_projection.log 2>&1 && echo shards ok &&
timeout -k 10 200 python3 $R/bench_train.py --iters 40 > $O/bench_train.json 2>> $O/bench_n1.err && echo train ok &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 $R/bench_train.py --iters 20 --warmup 3 > $O/train_stats.log 2>&1 && echo train stats ok
