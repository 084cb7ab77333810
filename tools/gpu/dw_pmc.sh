#!/bin/bash
# issue-level counters of the training kernels (one rocprofv3 --pmc pass per counter set, kernel trace only), printed per kernel as
# means per launch: bash tools/gpu/dw_pmc.sh [kernel substring]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dw_pmc; rm -rf $O; mkdir -p $O
K=${1:-grad_batch_pair_dma_kernel}
cd /tmp; export TMPDIR=/tmp
T="python3 $R/bench_train.py --iters 12 --warmup 3"
pass() { timeout -k 10 240 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O/$2 -- $T > $O/$2.log 2>&1; }
pass "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" p1 && echo p1 ok &&
pass "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SALU" p2 && echo p2 ok &&
pass "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" p3 && echo p3 ok &&
pass "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" p4 && echo p4 ok &&
pass "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum" p5 && echo p5 ok
python3 - "$O" "$K" <<'PY'
import glob, sys
import pandas as pd
root, k = sys.argv[1], sys.argv[2]
for p in ("p1", "p2", "p3", "p4", "p5"):
    files = glob.glob(f"{root}/{p}/*/*_counter_collection.csv")
    if not files:
        print(p, "no output"); continue
    df = pd.read_csv(files[0])
    df["dur_ns"] = df.End_Timestamp - df.Start_Timestamp
    d = df[df.Kernel_Name.str.contains(k)]
    print(p, k, "launches", d.Dispatch_Id.nunique(), "mean us %.1f" % (d.drop_duplicates("Dispatch_Id").dur_ns.mean() / 1e3))
    for c, v in d.groupby("Counter_Name").Counter_Value.mean().items():
        print("   %-28s %.4g" % (c, v))
PY
