#!/usr/bin/env python3
"""Benchmark of the NeRF ray-chunk renderer on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One *step* = one pass of the hot path over one frame of synthetic rays: the 800x800 Blender-lego
camera at 64 coarse + 128 fine samples (BASELINE.json configs[2]; configs[3] is the same frame over
N GPUs). The frame's 640,000 rays are resident in HBM before the timed region, cut into N contiguous
shards (one process per GPU), rendered in the reference's 32,768-ray chunks
(nerf/yaml/lego_blender200k_fullres:6) and gathered to rank 0 with one RCCL gather per frame - the
total work is fixed, so scaling is "strong". Weights are seeded synthetic tensors of the reference
architecture (no checkpoint or dataset exists offline).

1 ray-sample = 1 MLP point evaluation; a 64+128 ray costs 64 + 192 = 256 of them
(1,186,816 FLOP each). `value` is the whole-job rate over all N GPUs.
The roofline object prices the fused encode+MLP kernel with HIP events recorded on its launch stream
inside the timed region. Default arithmetic ("f16x2"): every fp32 operand is carried exactly as two fp16
halves and every product costs three v_mfma_f32_32x32x16_f16, so the algorithmic FLOP rate is priced
against one third of the dense fp16 MFMA peak (2516.6 / 3 = 838.9 TFLOP/s, MI355X_MICROARCH.md);
`--precision f32` runs the v_mfma_f32_32x32x2_f32 kernel, priced against 157.3 TFLOP/s. After the timed region the
other mode renders two frames of the same rays and its figures are reported in `other_precision`. The cpu_baseline object times the CPU oracle (numpy port, oracle/nerf_oracle.py) on
a bounded sample of the same rays; it is reported next to the GPU number, never used by it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 1186816          # 2 * 593,408 MACs
PEAK_FP32_MFMA_TFLOPS = 157.3    # dense v_mfma_f32_32x32x2_f32 peak, MI355X_MICROARCH.md
PEAK_FP16_MFMA_TFLOPS = 2516.6   # dense v_mfma_f32_32x32x16_f16 peak: 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz

WORKLOADS = {
    # name: (H, W, N_samples, N_importance, ndc, white_bkgd)
    "lego_800x800_64c+128f": (800, 800, 64, 128, False, True),
    "lego_400x400_64c": (400, 400, 64, 0, False, True),
    "fern_1008x756_ndc_64c+128f": (756, 1008, 64, 128, True, False),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default="lego_800x800_64c+128f", choices=sorted(WORKLOADS))
    p.add_argument("--chunk", type=int, default=32768)
    p.add_argument("--precision", default=None, choices=["f16x2", "f32", "f16x2_s16"],
                   help="arithmetic of the fused MLP kernel (default: the library's, f16x2)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-other-precision", action="store_true",
                   help="skip the short run in the other arithmetic mode that is reported next to the result")
    p.add_argument("--force-collective", action="store_true",
                   help="under torch.distributed.run with one rank: still create the RCCL group and gather")
    return p.parse_args()


def cpu_baseline(sample_rays, sd_c, sd_f, Sc, Si, white, target_s):
    """Time the CPU oracle on a bounded sample of the workload's rays (rank 0, N=1 only)."""
    from oracle import nerf_oracle as O
    net_c = O.NeRF(8, 256, 63, 27, 4, (4,), True, sd_c)
    net_f = O.NeRF(8, 256, 63, 27, 4, (4,), True, sd_f)
    q = O.make_query_fn(O.get_embedder(10)[0], O.get_embedder(4)[0])
    kw = dict(N_samples=Sc, N_importance=Si, network_fine=net_f if Si else None, white_bkgd=white)
    probe = min(256, len(sample_rays))
    t0 = time.perf_counter()
    O.render_rays(sample_rays[:probe], net_c, q, **kw)
    dt = time.perf_counter() - t0
    n = int(min(len(sample_rays), max(probe, target_s / max(dt, 1e-6) * probe)))
    n = max(64, (n // 64) * 64)
    t0 = time.perf_counter()
    ret = O.batchify_rays(sample_rays[:n], 4096, network_fn=net_c, network_query_fn=q, **kw)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [os.cpu_count()])
    except Exception:
        cores = os.cpu_count()
    evals = n * (Sc + (Sc + Si if Si else 0))
    return ret, n, {"value": evals / dt, "unit": "ray-samples/s", "cores": int(cores), "kind": "port",
                    "sample": f"{n} rays of the same frame ({evals} MLP evals) in {dt:.1f} s, numpy/OpenBLAS "
                              f"oracle on the host CPU ({os.cpu_count()} logical cores visible); "
                              f"{n / dt:.0f} rays/s"}


def pmc_traffic(precision):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this
    same command (profiles/rNN_pmc_summary.json for f16x2, rNN_pmc_summary_f32.json for --precision f32,
    newest round; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950, WRITE_SIZE as read).
    bench.py itself cannot read PMC counters."""
    import glob
    suffix = "" if precision == "f16x2" else "_" + precision
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_summary{suffix}.json")))
    if not files:
        return None, None
    s = json.load(open(files[-1]))
    return s.get("hbm_bytes_per_launch_fetch_x2"), os.path.relpath(files[-1], ROOT)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    use_dist = world > 1 or (args.force_collective and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import nerf_projects_amd as N
    from nerf_projects_amd import synthetic

    H, W, Sc, Si, ndc, white = WORKLOADS[args.workload]
    if ndc:
        K, c2w, near, far = synthetic.fern_camera(H, W)
    else:
        K, c2w, near, far = synthetic.lego_camera(H, W)
    sd_c, sd_f = synthetic.synthetic_pair(0)
    mk = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    net_c = N.NeRF(**mk).load_state_dict(sd_c)
    net_f = N.NeRF(**mk).load_state_dict(sd_f)
    query = N.make_network_query_fn(N.get_embedder(10, 0)[0], N.get_embedder(4, 0)[0])
    kw = dict(network_fn=net_c, network_query_fn=query, N_samples=Sc, N_importance=Si,
              network_fine=net_f if Si else None, white_bkgd=white, perturb=0., raw_noise_std=0.)

    # inputs resident in HBM before the timed region: the packed ray record of this rank's shard
    packed, sh = N.pack_rays(H, W, K, c2w=c2w, ndc=ndc, near=near, far=far, use_viewdirs=True, device="cuda")
    n_total = packed.shape[0]
    lo, hi = N.shard_bounds(n_total, world, rank)
    shard = packed[lo:hi].contiguous()
    del packed
    evals_per_ray = Sc + (Sc + Si if Si else 0)
    ctx = N.get_context()
    if args.precision:
        ctx.set_precision(args.precision)
    precision = ctx.get_precision()

    def step():
        ret = N.batchify_rays(shard, args.chunk, **kw)
        local_out = {k: ret[k] for k in ("rgb_map", "disp_map", "acc_map")}
        if use_dist:
            return N.gather_frame(local_out, n_total, force_collective=True)
        return local_out

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        # RCCL opens its point-to-point connections on first use: one tiny gather before anything is timed, so that a
        # run with --warmup 0 does not time communicator setup (this is not a render step)
        N.gather_frame({"acc_map": torch.zeros(1, device="cuda")}, world, force_collective=True)
    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = step()
    fence()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    mlp_ms, mlp_launches, mlp_points = ctx.profile_read(reset=True)
    t = torch.tensor([dt, mlp_ms, float(mlp_points), float(mlp_launches)], device="cuda", dtype=torch.float64)
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        mlp_ms_sum, pts_sum, launches_sum = float(t[1]), float(t[2]), float(t[3])
    else:
        mlp_ms_sum, pts_sum, launches_sum = mlp_ms, float(mlp_points), float(mlp_launches)

    # the other arithmetic mode on the same rays, outside the timed region above: 1 warm-up + 2 timed frames,
    # reported in `other_precision` so that both kernels' numbers come from one run
    other = None
    if not args.no_other_precision:
        alt = "f32" if precision.startswith("f16x2") else "f16x2"
        ctx.set_precision(alt)
        step()
        fence()
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        fence()
        dt_alt = time.perf_counter() - t1
        ctx.profile_enable(False)
        a_ms, a_launches, a_points = ctx.profile_read(reset=True)
        ctx.set_precision(precision)
        ta = torch.tensor([dt_alt], device="cuda", dtype=torch.float64)
        if use_dist:
            dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        dt_alt = float(ta[0])
        a_tf = a_points / max(a_launches, 1) * FLOP_PER_EVAL / max(a_ms / max(a_launches, 1) * 1e-3, 1e-12) / 1e12
        a_peak = PEAK_FP32_MFMA_TFLOPS if alt == "f32" else PEAK_FP16_MFMA_TFLOPS / 3
        other = {"precision": alt, "steps": 2, "ms_per_step": dt_alt / 2 * 1e3,
                 "value": n_total * evals_per_ray * 2 / dt_alt, "unit": "ray-samples/s",
                 "roofline": {"achieved": a_tf, "peak": a_peak, "frac": a_tf / a_peak, "unit": "TFLOP/s",
                              "avg_launch_ms": a_ms / max(a_launches, 1), "note": "rank 0's kernel time"}}

    if rank == 0:
        total_evals = n_total * evals_per_ray * args.steps
        value = total_evals / dt
        # dominant kernel: algorithmic FLOP per launch / average launch duration (HIP events, per GPU)
        flop_per_launch = pts_sum / max(launches_sum, 1) * FLOP_PER_EVAL
        avg_launch_s = mlp_ms_sum / max(launches_sum, 1) * 1e-3
        achieved = flop_per_launch / max(avg_launch_s, 1e-12) / 1e12
        traffic, traffic_src = pmc_traffic(precision) if args.workload == "lego_800x800_64c+128f" else (None, None)
        if precision.startswith("f16x2"):
            products = 3          # W_lo*x_hi + W_hi*x_lo + W_hi*x_hi per term
            peak = PEAK_FP16_MFMA_TFLOPS / products
            shape = "16x16x32" if precision == "f16x2_s16" else "32x32x16"
            arith = {"dtype": f"f32 operands as exact fp16 pairs, fp32 accumulate (v_mfma_f32_{shape}_f16 x3)",
                     "kernel": "nerf_mlp_h3_kernel<rays>" if precision == "f16x2_s16" else "nerf_mlp_h2_kernel<rays>", "mfma_pipe": "f16", "mfma_pipe_peak": PEAK_FP16_MFMA_TFLOPS,
                     "mfma_products_per_term": products, "mfma_executed": achieved * products,
                     "vs_f32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS}
        else:
            peak = PEAK_FP32_MFMA_TFLOPS
            arith = {"dtype": "f32 (v_mfma_f32_32x32x2_f32)", "kernel": "nerf_mlp_kernel<rays>", "mfma_pipe": "f32",
                     "mfma_pipe_peak": PEAK_FP32_MFMA_TFLOPS, "mfma_products_per_term": 1, "mfma_executed": achieved,
                     "vs_f32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS}
        out = {
            "metric": "ray_samples_per_sec", "value": value, "unit": "ray-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": arith["dtype"],
            "data": "synthetic",
            "config": {"workload": args.workload, "rays_per_frame": n_total, "N_samples": Sc, "N_importance": Si,
                       "evals_per_ray": evals_per_ray, "chunk": args.chunk, "netdepth": 8, "netwidth": 256,
                       "precision": precision,
                       "parallelism": f"ray-shard x{world} + gather" if world > 1 else "single GPU"},
            "rays_per_sec": n_total * args.steps / dt,
            "ray_samples_per_sec_per_gpu": value / world,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                         "traffic_source": traffic_src,
                         "kernel": arith["kernel"], "mfma_pipe": arith["mfma_pipe"],
                         "mfma_pipe_peak": arith["mfma_pipe_peak"],
                         "mfma_products_per_term": arith["mfma_products_per_term"],
                         "mfma_executed": arith["mfma_executed"], "vs_f32_mfma_peak": arith["vs_f32_mfma_peak"],
                         "launches": int(launches_sum),
                         "avg_launch_ms": avg_launch_s * 1e3, "flop_per_launch": flop_per_launch,
                         "kernel_time_share": mlp_ms_sum * 1e-3 / world / dt},
        }
        if other is not None:
            out["other_precision"] = other
        if world == 1 and not args.no_cpu_baseline:
            # same rays, spread over the whole frame so empty, grazing and opaque rays are all present
            idx = np.linspace(0, n_total - 1, 4096).astype(np.int64)
            sample = shard[torch.from_numpy(idx).cuda()].cpu().numpy()
            ref, n_used, cb = cpu_baseline(sample, sd_c, sd_f, Sc, Si, white, args.cpu_seconds)
            out["cpu_baseline"] = cb
            got = frame["rgb_map"][torch.from_numpy(idx[:n_used]).cuda()].cpu().numpy()
            err = np.abs(got - ref["rgb_map"]).max(-1)
            mse = float(np.mean((got - ref["rgb_map"]) ** 2))
            out["parity"] = {"rays": int(n_used), "rgb_linf": float(err.max()), "rgb_p99": float(np.quantile(err, .99)),
                             "rgb_median": float(np.median(err)), "rays_above_1e-4": int((err > 1e-4).sum()),
                             "psnr_vs_cpu_oracle_db": (float(-10 * np.log10(mse)) if mse > 0 else float("inf"))}
            if Si:
                e0 = np.abs(frame_rgb0(N, shard, idx[:n_used], kw) - ref["rgb0"]).max()
                out["parity"]["rgb0_linf"] = float(e0)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def frame_rgb0(N, shard, idx, kw):
    """Coarse-pass colours of the sampled rays (the well-conditioned half of the parity report)."""
    rays = shard[torch.from_numpy(idx).cuda()].contiguous()
    return N.batchify_rays(rays, 32768, **kw)["rgb0"].cpu().numpy()


if __name__ == "__main__":
    main()
