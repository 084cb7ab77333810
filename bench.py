#!/usr/bin/env python3
"""Benchmark of the NeRF ray-chunk renderer on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work for any N. Called plainly with N > 1 (or with --force-collective), bench.py starts the second form
itself as a CHILD process - before this process has touched the GPU - relays the child's output and exits with its
return code (self_launch_command below).

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
`--precision f32` runs the v_mfma_f32_32x32x2_f32 kernel, priced against 157.3 TFLOP/s. After the timed region (N=1):
the other mode renders five timed frames of the same rays (`other_precision`); a short training run reports it/s of
the reference's loop body (`train`); the `parity` object compares the HIP render of the 4096 fixture rays of this frame
with the REFERENCE's own fp32 render of them (tests/golden/bench_frame.npz, oracle/parity.py); the `cpu_baseline` object
times the CPU oracle (oracle/nerf_oracle.py) on a bounded sample of the same rays with the thread count that maximises
it. These legs are reported next to the GPU number, never used by it.
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
    p.add_argument("--precision", default=None, choices=["f16x2", "f32"],
                   help="arithmetic of the fused MLP kernel (default: the library's, f16x2)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    p.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (cpu_baseline and parity)")
    p.add_argument("--no-train", action="store_true", help="skip the short training-throughput leg")
    p.add_argument("--no-other-precision", action="store_true",
                   help="skip the short run in the other arithmetic mode that is reported next to the result")
    p.add_argument("--force-collective", action="store_true",
                   help="under torch.distributed.run with one rank: still create the RCCL group and gather")
    return p.parse_args()


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when a communicator is created; the contract is ONE JSON line there. File
    descriptor 1 points at stderr while the process group and its first collective come up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def self_launch_command(argv, n_gpus, port=None):
    """The torch.distributed.run command line a plain `python bench.py --gpus N ...` turns into (one rank per GPU of this
    node over RCCL; 127.0.0.1 because a container's hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """Run this benchmark under torch.distributed.run as a child process and return its exit code. Nothing in this
    process has initialised the GPU (module-level `import torch` does not), so the hop is an ordinary subprocess - never an
    exec - and the child's ranks are the only processes on the devices."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this host driver
    env["NERF_BENCH_SELF_LAUNCHED"] = "1"
    cmd = self_launch_command(sys.argv[1:], args.gpus)
    print("bench.py: not under torch.distributed.run; starting " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:                                 # relay as it comes: the contract line is rank 0's stdout
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def cpu_baseline(sample_rays, sd_c, sd_f, Sc, Si, white, target_s):
    """Time the CPU oracle on a bounded sample of the workload's rays (rank 0, N=1 only).

    The oracle's GEMMs run through PyTorch's CPU sgemm (oracle.set_gemm_backend("torch"): the BLAS the reference itself
    uses; numpy's OpenBLAS is ~2.5x slower at equal threads and is timed once for the record). The thread count is the
    one that maximises the rate on a 512-ray probe, the batch the reference's own netchunk (65536 points)."""
    from oracle import nerf_oracle as O
    net_c = O.NeRF(8, 256, 63, 27, 4, (4,), True, sd_c)
    net_f = O.NeRF(8, 256, 63, 27, 4, (4,), True, sd_f)
    q = O.make_query_fn(O.get_embedder(10)[0], O.get_embedder(4)[0])
    kw = dict(N_samples=Sc, N_importance=Si, network_fine=net_f if Si else None, white_bkgd=white)
    evals_per_ray = Sc + (Sc + Si if Si else 0)
    probe = sample_rays[:: max(1, len(sample_rays) // 512)][:512]

    def rate(rays, chunk=1024):
        t0 = time.perf_counter()
        O.batchify_rays(rays, chunk, network_fn=net_c, network_query_fn=q, **kw)
        return len(rays) * evals_per_ray / (time.perf_counter() - t0)

    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    t_numpy = rate(probe[:128])                         # numpy/OpenBLAS backend, its own thread pool
    O.set_gemm_backend("torch")
    old_threads = torch.get_num_threads()
    tried = {}
    try:
        rate(probe[:64])
        for t in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
            torch.set_num_threads(t)
            tried[t] = rate(probe)
        best = max(tried, key=tried.get)
        torch.set_num_threads(best)
        # (half of the budget for this single-pool sample, half for the whole-host leg below)
        n = int(min(len(sample_rays), max(256, 0.5 * target_s * tried[best] / evals_per_ray)))
        n = max(64, (n // 64) * 64)
        t0 = time.perf_counter()
        O.batchify_rays(sample_rays[:n], 1024, network_fn=net_c, network_query_fn=q, **kw)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(old_threads)
        O.set_gemm_backend("numpy")
    evals = n * evals_per_ray
    whole = cpu_baseline_all_cores(sample_rays, Sc, Si, white, best, ncpu, evals / dt, 0.5 * target_s)
    return {"value": evals / dt, "unit": "ray-samples/s", "cores": int(best), "kind": "port",
            "per_core": evals / dt / best, **whole,
            "threads_tried": {str(k): round(v) for k, v in tried.items()},
            "numpy_openblas_backend": round(t_numpy),
            "reference_pytorch_cpu_survey_container": "0.96e5-1.24e5 ray-samples/s on 8 threads (BASELINE.md section 2; other host)",
            "sample": f"{n} rays of the same frame ({evals} MLP evals) in {dt:.1f} s: CPU oracle (numpy restatement, "
                      f"GEMMs through PyTorch's CPU sgemm, 1024-ray chunks = 65536-point MLP batches) on {best} threads "
                      f"of the host ({ncpu} logical cores usable), {n / dt:.0f} rays/s, {evals / dt / best:.0f} "
                      f"ray-samples/s/core; numpy/OpenBLAS backend on the probe: {t_numpy:.0f} ray-samples/s; the "
                      f"reference's own PyTorch-CPU path measured in the survey container: 0.96e5-1.24e5 on 8 threads"}


def cpu_baseline_all_cores(sample_rays, Sc, Si, white, threads, ncpu, single_pool_rate, target_s):
    """The node's host cores all at once (north_star: "timed on the node's own host cores"): one BLAS thread pool stops
    scaling at `threads` threads (threads_tried), so P = ncpu // threads worker PROCESSES (oracle/cpu_worker.py: fresh
    children, GPU hidden from them, `threads` threads each) render disjoint slices of the same sample side by side, started
    together; the rate is all their ray-samples over the time until the last one has finished."""
    import subprocess
    import tempfile
    procs = max(1, ncpu // max(threads, 1))
    evals_per_ray = Sc + (Sc + Si if Si else 0)
    if procs < 2:
        return {"value_all_cores": single_pool_rate, "processes": 1, "threads_per_process": int(threads),
                "all_cores_note": "one process already uses every usable core"}
    # side by side the processes share memory bandwidth: assume 60 % of the single-pool rate each when sizing the slices
    n_each = int(max(64, min(len(sample_rays) // procs, target_s * 0.6 * single_pool_rate / evals_per_ray)))
    n_each = max(64, n_each // 64 * 64)
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               PYTHONDONTWRITEBYTECODE="1")
    workers, files = [], []
    try:
        for p in range(procs):
            f = tempfile.NamedTemporaryFile(suffix=".npy", delete=False)
            f.close()
            np.save(f.name, np.ascontiguousarray(sample_rays[p * n_each:(p + 1) * n_each]))
            files.append(f.name)
            workers.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), str(threads),
                                             f.name, str(Sc), str(Si), str(int(bool(white)))], env=env, stdin=subprocess.PIPE,
                                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        for w in workers:
            if w.stdout.readline().strip() != "ready":
                raise RuntimeError("a CPU worker did not come up")
        t0 = time.perf_counter()
        for w in workers:
            w.stdin.write("go\n")
            w.stdin.flush()
        results = [json.loads(w.stdout.readline()) for w in workers]
        wall = time.perf_counter() - t0
        for w in workers:
            w.wait(timeout=30)
    except Exception as e:                      # reported, never fatal for the GPU line
        for w in workers:
            if w.poll() is None:
                w.kill()
        return {"value_all_cores": None, "processes": procs, "threads_per_process": int(threads),
                "all_cores_note": f"failed: {type(e).__name__}: {e}"}
    finally:
        for name in files:
            try:
                os.unlink(name)
            except OSError:
                pass
    total = sum(r["rays"] for r in results) * evals_per_ray
    return {"value_all_cores": total / wall, "processes": procs, "threads_per_process": int(threads),
            "cores_all": procs * int(threads),
            "all_cores_note": f"{procs} processes x {threads} threads, {n_each} rays each, started together: {total} MLP evals "
                              f"in {wall:.1f} s (slowest worker {max(r['seconds'] for r in results):.1f} s, fastest "
                              f"{min(r['seconds'] for r in results):.1f} s)"}


def reference_parity(N, net_c, net_f, query):
    """HIP render of the 4096 fixture rays of this frame against the REFERENCE's fp32 render of the same rays
    (tests/golden/bench_frame.npz, generated by tests/golden/make_golden.py from the reference itself), with the
    reference's own fp32-vs-fp64 flips beside it; criterion and definitions: oracle/parity.py."""
    from oracle import parity
    path = os.path.join(ROOT, "tests", "golden", "bench_frame.npz")
    g = np.load(path)
    rays = torch.from_numpy(g["rays"]).cuda()
    kw = dict(N_samples=64, N_importance=128, network_fine=net_f, white_bkgd=True)
    ex = {}
    ret = N.render_rays(rays, net_c, query, _extras=ex, **kw)
    z_fine = np.sort(np.concatenate([ex["z_coarse"].cpu().numpy(), g["z_samples"]], -1), -1)      # nerf.ipynb:467
    inj = N.render_rays(rays, net_c, query, _z_vals_fine=z_fine, **kw)
    got = {k: v.cpu().numpy() for k, v in ret.items()}
    out = {"against": "reference fp32 render of 4096 rays of this frame (tests/golden/bench_frame.npz)",
           "rgb0_linf": float(np.abs(got["rgb0"] - g["rgb0"]).max()),
           "acc0_linf": float(np.abs(got["acc0"] - g["acc0"]).max())}
    try:
        st = parity.check_resampled(got, g, injected={k: v.cpu().numpy() for k, v in inj.items()}, fp64=g,
                                    foreground=g["acc0"] > 1e-3)
        out["criterion"] = "pass"
    except AssertionError as e:                      # reported, not hidden: the numbers below are then partial
        st = {}
        out["criterion"] = "FAIL: " + str(e)[:300]
    st.pop("flip_rays", None)
    out.update(st)
    mse = float(np.mean((got["rgb_map"] - g["rgb_map"]) ** 2))
    out["psnr_vs_reference_db"] = float(-10 * np.log10(mse)) if mse > 0 else float("inf")
    return out


def train_leg(N, synthetic, iters=30, warmup=5, n_rand=1024, Sc=64, Si=128):
    """Training iterations per second of the reference's loop body (nerf.ipynb:1258-1282) at N_rand = 1024, 64+128:
    the same measurement as bench_train.py, short, after the timed region."""
    sd_c, sd_f = synthetic.synthetic_pair(0)
    mk = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    net_c, net_f = N.NeRF(**mk).load_state_dict(sd_c), N.NeRF(**mk).load_state_dict(sd_f)
    opt = N.Adam([net_c, net_f], lr=5e-4, betas=(0.9, 0.999))
    K, c2w, near, far = synthetic.lego_camera(800, 800)
    packed = N.generate_rays(800, 800, K, c2w, ndc=False, near=near, far=far, use_viewdirs=True)
    kw = dict(network_fn=net_c, network_fine=net_f, N_samples=Sc, N_importance=Si, white_bkgd=True, perturb=1.0,
              raw_noise_std=1.0, ndc=False, use_viewdirs=True, near=near, far=far)
    torch.manual_seed(0)

    # batches as the reference's use_batching mode draws them (nerf.ipynb:1209-1230): one shuffle, consecutive windows
    state = {"perm": torch.randperm(packed.shape[0], device="cuda"), "i_batch": 0}

    def one(i):
        if state["i_batch"] + n_rand > packed.shape[0]:
            state["perm"], state["i_batch"] = torch.randperm(packed.shape[0], device="cuda"), 0
        idx = state["perm"][state["i_batch"]:state["i_batch"] + n_rand]
        state["i_batch"] += n_rand
        r = packed[idx]
        target = torch.rand((n_rand, 3), device="cuda")
        out = N.train_on_batch(800, 800, K, (r[:, 0:3], r[:, 3:6]), target, opt, **kw)
        opt.param_groups[0]['lr'] = 5e-4 * (0.1 ** (i / 500000))
        return out

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    ctx = N.get_context()
    t0 = time.perf_counter()
    for i in range(iters):
        out = one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the step's kernels, each against the roofline of the pipe it runs on (HIP events on the step's stream, a second short
    # run after the timed one: the events are not inside the reported it/s)
    ctx.profile_enable(True)
    ctx.profile_read_train(reset=True)
    n_prof = 10
    for i in range(n_prof):
        one(i)
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    spans = ctx.profile_read_train(reset=True)
    pair = ctx.get_precision() == "f16x2"
    env = lambda k: os.environ.get(k, "").lower().startswith("f3")
    pair_forward, pair_bwd, pair_dw = (pair and not env("NERF_TRAIN_FORWARD"), pair and not env("NERF_TRAIN_FORWARD") and
                                       not env("NERF_TRAIN_BWD"), pair and not env("NERF_TRAIN_DW"))
    # algorithmic FLOP per point: forward 2 x 593 408; backward-data 2 x (128 x 256 + 256 x 256 + 256 + 7 x 256 x 256 + 3 x 128);
    # hidden-width weight gradients 2 x (8 x 256 x 256 + 128 x 256); they read dY and X once per job: 10 jobs x 2 KB per point
    flop = {"forward": FLOP_PER_EVAL, "backward_data": 2 * (128 * 256 + 256 * 256 + 256 + 7 * 256 * 256 + 3 * 128),
            "weight_gradients_hidden": 2 * (8 * 256 * 256 + 128 * 256)}
    pipe_pair, pipe_f32 = PEAK_FP16_MFMA_TFLOPS / 3, PEAK_FP32_MFMA_TFLOPS
    kernels = {}
    for name, on_pair in (("forward", pair_forward), ("backward_data", pair_bwd), ("weight_gradients_hidden", pair_dw)):
        ms, launches, pts = spans[name]
        if launches == 0:
            continue
        tf = pts * flop[name] / (ms * 1e-3) / 1e12
        peak = pipe_pair if on_pair else pipe_f32
        k = {"ms_per_iter": ms / n_prof, "tflops": tf, "pipe": "f16 (3 products per term)" if on_pair else "f32",
             "pipe_peak_tflops": peak, "frac_of_pipe_peak": tf / peak}
        if name == "weight_gradients_hidden":
            gbs = pts * 10 * 2048 / (ms * 1e-3) / 1e9      # (incl. the slice reductions; every operand byte is read once)
            k.update(hbm_gb_per_s=gbs, frac_of_hbm_peak=gbs / 8000.0, bound="hbm" if on_pair else "mfma")
        kernels[name] = k
    ms_o, _, _ = spans["weight_gradients_other"]
    kernels["weight_gradients_other"] = {"ms_per_iter": ms_o / n_prof, "pipe": "f32"}
    # networks without view directions (use_viewdirs=False: output_linear on the trunk, nerf.ipynb:885-896): the same loop body,
    # short (since round 4 on the same three fp16-pair kernels, output_linear as one more chunk in both directions)
    nv = dict(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
    nv_c, nv_f = (N.NeRF(**nv).load_state_dict(synthetic.synthetic_state_dict(s_, input_ch_views=0, use_viewdirs=False, output_ch=5))
                  for s_ in (8, 48))
    nv_opt = N.Adam([nv_c, nv_f], lr=5e-4, betas=(0.9, 0.999))
    nv_kw = dict(kw, network_fn=nv_c, network_fine=nv_f, use_viewdirs=False)

    def one_nv(i):
        idx = state["perm"][(i * n_rand) % (packed.shape[0] - n_rand):][:n_rand]
        r = packed[idx]
        return N.train_on_batch(800, 800, K, (r[:, 0:3], r[:, 3:6]), torch.rand((n_rand, 3), device="cuda"), nv_opt, **nv_kw)

    for i in range(warmup):
        one_nv(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_nv = 20
    for i in range(n_nv):
        one_nv(i)
    torch.cuda.synchronize()
    noviewdirs = {"value": n_nv / (time.perf_counter() - t1), "unit": "it/s", "iters": n_nv,
                  "model": "use_viewdirs=False, output_ch=5 (create_nerf's other branch)"}
    evals = n_rand * (Sc + Sc + Si)
    tf = evals * FLOP_PER_EVAL * 3 * iters / dt / 1e12          # forward + dX + dW
    return {"metric": "train_iterations_per_sec", "value": iters / dt, "unit": "it/s", "iters": iters,
            "ms_per_iter": dt / iters * 1e3, "n_rand": n_rand, "N_samples": Sc, "N_importance": Si,
            "tflops_effective": tf, "kernels": kernels, "train_noviewdirs": noviewdirs,
            "arithmetic": ("forward, backward-data and the weight gradients of every Linear but rgb_linear: fp16-pair kernels on the "
                           "row-equalised network (3 x v_mfma_f32_32x32x16_f16 per term, fp32 accumulate, fp32-level error), kept "
                           "activations blocked by 32 points; "
                           if (pair_forward and pair_bwd and pair_dw) else
                           f"forward: {'fp16-pair' if pair_forward else 'f32'}; backward-data: {'fp16-pair' if pair_bwd else 'f32'}; "
                           f"hidden-width weight gradients: {'fp16-pair' if pair_dw else 'f32'}; ") +
                          "rgb_linear's weight gradient, compositing, Adam: f32, fp32 master weights",
            "batches": "use_batching windows of one shuffle (nerf.ipynb:1209-1230)",
            "final_loss": float(out["loss"]),
            "reference_stored_run": "5.6-7.4 it/s (ship 96+192, unknown CUDA GPU; BASELINE.md section 1)"}


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
    # dmabuf IPC between the ranks of one node: this pool's host driver supports nothing else, and HSA reads the variable when
    # the first GPU call initialises it - so it is set here, before torch.cuda.set_device, in BOTH launch forms (the image
    # exports it already; a launcher that scrubs the environment must not be able to undo that)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_collective):
        # plain `python bench.py --gpus N`: become the launcher (before any GPU call) instead of asking for one
        if os.environ.get("NERF_BENCH_SELF_LAUNCHED"):
            raise SystemExit("bench.py: the launcher did not set WORLD_SIZE")
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    use_dist = world > 1 or (args.force_collective and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
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

    # inputs resident in HBM before the timed region: the packed ray record of this rank's shard, generated on this
    # rank's GPU for its own pixel range only (nerf_generate_rays; no rank ever holds the whole frame)
    n_total = H * W
    lo, hi = N.shard_bounds(n_total, world, rank)
    shard = N.generate_rays(H, W, K, c2w, ndc=ndc, near=near, far=far, use_viewdirs=True, first_pixel=lo,
                            n_pixels=hi - lo)
    evals_per_ray = Sc + (Sc + Si if Si else 0)
    ctx = N.get_context()
    if args.precision:
        ctx.set_precision(args.precision)
    precision = ctx.get_precision()

    gather_events = []      # (start, end) around every frame's gather on this rank's stream (N > 1 only)

    def step():
        ret = N.batchify_rays(shard, args.chunk, **kw)
        local_out = {k: ret[k] for k in ("rgb_map", "disp_map", "acc_map")}
        if use_dist:
            # the collective runs on RCCL's own stream, which torch orders after `e0` and makes the current stream wait for
            # before `e1`: the pair brackets packing + wire time + (on the other ranks) the wait for the slowest shard
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = N.gather_frame(local_out, n_total, force_collective=True)
            e1.record()
            gather_events.append((e0, e1))
            return out
        return local_out

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        # RCCL opens its point-to-point connections on first use: one tiny gather before anything is timed, so that a
        # run with --warmup 0 does not time communicator setup (this is not a render step)
        with stdout_to_stderr():
            N.gather_frame({"acc_map": torch.zeros(1, device="cuda")}, world, force_collective=True)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    gather_events.clear()
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
    gather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / max(len(gather_events), 1)
    t = torch.tensor([dt, mlp_ms, float(mlp_points), float(mlp_launches)], device="cuda", dtype=torch.float64)
    per_rank = None
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # one line must be enough to read a scaling record: every rank's own clock, kernel time and gather time
        mine = torch.tensor([dt / args.steps * 1e3, mlp_ms / args.steps, gather_ms, float(hi - lo)], device="cuda",
                            dtype=torch.float64)
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = {"ms_per_step": [round(float(r[0]), 3) for r in rows],
                    "mlp_kernel_ms_per_step": [round(float(r[1]), 3) for r in rows],
                    "gather_ms": [round(float(r[2]), 3) for r in rows],
                    "rays": [int(r[3]) for r in rows]}
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        mlp_ms_sum, pts_sum, launches_sum = float(t[1]), float(t[2]), float(t[3])
    else:
        mlp_ms_sum, pts_sum, launches_sum = mlp_ms, float(mlp_points), float(mlp_launches)

    # the other arithmetic mode on the same rays, outside the timed region above: 1 warm-up + 5 timed frames,
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
        n_alt = 5
        for _ in range(n_alt):
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
        other = {"precision": alt, "steps": n_alt, "warmup": 1, "ms_per_step": dt_alt / n_alt * 1e3,
                 "value": n_total * evals_per_ray * n_alt / dt_alt, "unit": "ray-samples/s",
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
            arith = {"dtype": "f32 operands as fp16 (hi, lo) pairs, fp32 accumulate (v_mfma_f32_32x32x16_f16 x3)",
                     "kernel": "nerf_mlp_h2_kernel<rays>", "mfma_pipe": "f16", "mfma_pipe_peak": PEAK_FP16_MFMA_TFLOPS,
                     "mfma_products_per_term": products, "mfma_executed": achieved * products,
                     "vs_f32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS}
        else:
            peak = PEAK_FP32_MFMA_TFLOPS
            arith = {"dtype": "f32 (v_mfma_f32_32x32x2_f32)", "kernel": "nerf_mlp_kernel<rays>", "mfma_pipe": "f32",
                     "mfma_pipe_peak": PEAK_FP32_MFMA_TFLOPS, "mfma_products_per_term": 1, "mfma_executed": achieved,
                     "vs_f32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS}
        out = {
            "metric": "ray_samples_per_sec", "value": value, "unit": "ray-samples/s", "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if use_dist else 1,      # what the process group itself reports
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
        if per_rank is not None:
            # gather_ms: HIP events around gather_frame on each rank (rank 0's is the collective as the root sees it; the
            # others' include waiting for the root); ms_per_step: each rank's own wall clock over the timed steps
            out["per_rank"] = per_rank
            out["gather_ms"] = per_rank["gather_ms"][0]
            out["hsa_enable_ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
        if other is not None:
            out["other_precision"] = other
        out["precision_status"] = int(ctx.precision_status())      # layers whose a-priori scale bound was loose (0 = none)
        if world == 1 and not args.no_train and not use_dist:
            out["train"] = train_leg(N, synthetic)
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "lego_800x800_64c+128f":
                out["parity"] = reference_parity(N, net_c, net_f, query)
            # same rays, spread over the whole frame so empty, grazing and opaque rays are all present
            idx = np.linspace(0, n_total - 1, 4096).astype(np.int64)
            sample = shard[torch.from_numpy(idx).cuda()].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample, sd_c, sd_f, Sc, Si, white, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
