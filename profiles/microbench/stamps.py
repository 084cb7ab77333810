"""Per-step cycle timeline of one wave of the fp16-pair MLP kernel (diagnostic build -DNERF_STAMPS).

    NERF_EXTRA_FLAGS=-DNERF_STAMPS NERF_LIB_OUT=$PWD/nerf-projects_amd/libnerf_stamps.so python nerf-projects_amd/build.py --force
    NERF_MI355X_LIB=.../libnerf_stamps.so NERF_STAMPS_FILE=/tmp/stamps.bin python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python profiles/microbench/stamps.py /tmp/stamps.bin

A record = (tag, s_memtime at the start of a step); tag = chunk << 8 | step << 4 | vector instructions paced per MFMA.
s_memtime ticks at 100 MHz on this part, so durations are coarse (1 tick = ~19 shader cycles); sums over many steps are exact.
"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2)
raw = raw[1:]                      # first record carries the dummy tag of the initial state
n = int(np.argmax(raw[:, 0] == 0x7fffff00)) if (raw[:, 0] == 0x7fffff00).any() else len(raw)
tags, t = raw[:n + 1, 0].astype(np.int64), raw[:n + 1, 1].astype(np.int64)
dt = np.diff(t)
chunk, step, valu = tags[:-1] >> 8, (tags[:-1] >> 4) & 15, tags[:-1] & 15
print(f"{len(dt)} steps, total {t[-1] - t[0]} ticks; mean {dt.mean():.2f} ticks per step")
print("by step index:", {int(s): round(float(dt[step == s].mean()), 2) for s in range(8)})
print("by paced vector instructions:", {int(v): round(float(dt[valu == v].mean()), 2) for v in np.unique(valu)})
per_chunk = np.array([dt[chunk == c].sum() for c in np.unique(chunk)])
print("ticks per chunk:", per_chunk.tolist())
