#!/bin/bash
# L1 / texture-addresser counters of the weight-gradient kernel, one counter per pass (a pass that rocprofv3 cannot schedule aborts alone)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dw_pmc3; rm -rf $O; mkdir -p $O
K=${1:-grad_batch_pair_dma_kernel}
cd /tmp; export TMPDIR=/tmp
T="python3 $R/bench_train.py --iters 8 --warmup 2"
n=0
for c in "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum" "TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/q$n -- $T > $O/q$n.log 2>&1 && echo "q$n ok" || echo "q$n failed"
done
python3 - "$O" "$K" <<'PY'
import glob, sys
import pandas as pd
root, k = sys.argv[1], sys.argv[2]
for p in sorted(glob.glob(f"{root}/q*/*/*_counter_collection.csv")):
    df = pd.read_csv(p)
    df["dur_ns"] = df.End_Timestamp - df.Start_Timestamp
    d = df[df.Kernel_Name.str.contains(k)]
    print(p.split("/")[-3], "launches", d.Dispatch_Id.nunique(), "mean us %.1f" % (d.drop_duplicates("Dispatch_Id").dur_ns.mean() / 1e3))
    for c, v in d.groupby("Counter_Name").Counter_Value.mean().items():
        print("   %-40s %.4g" % (c, v))
PY
