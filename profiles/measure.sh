#!/bin/bash
# round measurement (run on the GPU box: gpurun -- bash profiles/measure.sh [part]): bench lines, rocprofv3 kernel stats and the
# PMC passes of the default (fp16-pair) kernel and of the training step's kernels, the shard projection, the microbenchmark
# of the last lever. part = bench | pmc | train | all (default all; the parts fit one gpurun call each)
set -o pipefail
PART=${1:-all}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/measure; mkdir -p $O   # (clear the local gpurun_out/measure first: gpurun merges, it does not mirror)
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-precision --no-train"
T="python3 $R/bench_train.py --iters 12 --warmup 3 --only"
PMC_SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PMC_L2="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
pmc() { timeout -k 10 240 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O/$2 -- $3 > $O/$2.log 2>&1; }
if [ $PART = bench ] || [ $PART = all ]; then
timeout -k 10 400 python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err && echo bench ok &&
timeout -k 10 300 python3 $R/bench.py --precision f32 --no-cpu-baseline > $O/bench_n1_f32.json 2>> $O/bench_n1.err && echo f32 ok &&
timeout -k 10 300 python3 $R/bench.py --workload lego_400x400_64c --steps 10 > $O/bench_c1.json 2>> $O/bench_n1.err && echo c1 ok &&
timeout -k 10 300 python3 $R/bench.py --workload fern_1008x756_ndc_64c+128f --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench_n1.err && echo c4 ok &&
timeout -k 10 300 python3 $R/bench.py --gpus 1 --force-collective --steps 3 --no-cpu-baseline --no-other-precision > $O/bench_n1_rccl_group.json 2>> $O/bench_n1.err && echo rccl-group ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-precision --no-train > $O/stats.log 2>&1 && echo stats ok &&
timeout -k 10 400 python3 $R/profiles/shard_projection.py > $O/shard_projection.log 2>&1 && echo shards ok || exit 1
fi
if [ $PART = pmc ] || [ $PART = all ]; then
pmc "FETCH_SIZE" pmc_fetch "$B" && echo f ok && pmc "WRITE_SIZE" pmc_write "$B" && echo w ok &&
pmc "$PMC_SQ" pmc_sq "$B" && echo s ok && pmc "$PMC_L2" pmc_l2 "$B" && echo l ok || exit 1
fi
if [ $PART = train ] || [ $PART = all ]; then
timeout -k 10 200 python3 $R/bench_train.py --iters 60 > $O/bench_train.json 2>> $O/bench_n1.err && echo train ok &&
NERF_TRAIN_BLOCKED=0 timeout -k 10 200 python3 $R/bench_train.py --iters 60 --only > $O/bench_train_rowmajor.json 2>> $O/bench_n1.err && echo train row-major ok &&
NERF_TRAIN_GLUE=legacy timeout -k 10 200 python3 $R/bench_train.py --iters 60 --only > $O/bench_train_legacy_glue.json 2>> $O/bench_n1.err && echo train legacy glue ok &&
timeout -k 10 200 python3 $R/bench_train.py --iters 60 --no-viewdirs > $O/bench_train_noviewdirs.json 2>> $O/bench_n1.err && echo train no-viewdirs ok &&
NERF_TRAIN_BWD=f32 timeout -k 10 200 python3 $R/bench_train.py --iters 60 --only > $O/bench_train_bwd_f32.json 2>> $O/bench_n1.err && echo train bwd-f32 ok &&
NERF_PRECISION=f32 timeout -k 10 200 python3 $R/bench_train.py --iters 40 --only > $O/bench_train_all_f32.json 2>> $O/bench_n1.err && echo train f32 ok &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 $R/bench_train.py --iters 20 --warmup 3 --only > $O/train_stats.log 2>&1 && echo train stats ok &&
pmc "FETCH_SIZE" train_pmc_fetch "$T" && pmc "WRITE_SIZE" train_pmc_write "$T" && pmc "$PMC_SQ" train_pmc_sq "$T" && pmc "$PMC_L2" train_pmc_l2 "$T" && echo train pmc ok &&
true || exit 1
fi
