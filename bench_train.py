#!/usr/bin/env python3
"""Secondary benchmark: training iterations per second.

The reference's stored run trains at 5.6-7.4 it/s with N_rand = 1024 rays per iteration
(ship, 96+192 samples, unknown CUDA GPU). This times the same loop body
(nerf.ipynb:1258-1282: render with the training kwargs, two MSE losses, backward, Adam, lr decay)
on one MI355X with synthetic rays/targets and seeded weights.

    python bench_train.py [--iters 50] [--n-rand 1024] [--samples 64 --importance 128]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--iters", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--n-rand", type=int, default=1024)
    p.add_argument("--samples", type=int, default=64)
    p.add_argument("--importance", type=int, default=128)
    p.add_argument("--no-viewdirs", action="store_true", help="networks without view directions (use_viewdirs=False, 5-channel head)")
    p.add_argument("--only", action="store_true", help="no second, short run of create_nerf's other branch (train_noviewdirs)")
    a = p.parse_args()
    torch.cuda.set_device(0)
    res = run(a, a.no_viewdirs, a.iters, a.warmup, True)
    if not a.no_viewdirs and not a.only:      # create_nerf's other branch (nerf.ipynb:885-896), 20 iterations of it
        nv = run(a, True, 20, 3, False)
        res["train_noviewdirs"] = {"value": nv["value"], "unit": "it/s", "ms_per_iter": nv["ms_per_iter"], "iters": 20,
                                   "model": "use_viewdirs=False, output_ch=5"}
    print(json.dumps(res))


def run(a, no_viewdirs, iters, warmup, spans_wanted):
    import nerf_projects_amd as N
    from nerf_projects_amd import synthetic
    sd_c, sd_f = synthetic.synthetic_pair(0)
    mk = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    if no_viewdirs:
        mk.update(input_ch_views=0, use_viewdirs=False)
        sd_c, sd_f = (synthetic.synthetic_state_dict(s, input_ch_views=0, use_viewdirs=False, output_ch=5) for s in (8, 48))
    net_c, net_f = N.NeRF(**mk).load_state_dict(sd_c), N.NeRF(**mk).load_state_dict(sd_f)
    opt = N.Adam([net_c, net_f], lr=5e-4, betas=(0.9, 0.999))
    K, c2w, near, far = synthetic.lego_camera(800, 800)
    packed = N.generate_rays(800, 800, K, c2w, ndc=False, near=near, far=far, use_viewdirs=True)
    kw = dict(network_fn=net_c, network_fine=net_f, N_samples=a.samples, N_importance=a.importance, white_bkgd=True,
              perturb=1.0, raw_noise_std=1.0, ndc=False, use_viewdirs=not no_viewdirs, near=near, far=far)
    torch.manual_seed(0)
    lrate, lrate_decay = 5e-4, 500

    # The batches: the reference's use_batching mode (nerf.ipynb:1209-1230) - all rays shuffled once, consecutive windows of
    # N_rand, a new shuffle after an epoch - so that the step, not the sampler, is what is timed (a fresh permutation of the
    # 640 000 pixels per iteration, as the per-image mode's np.random.choice(replace=False) does, is a sort of its own:
    # 0.18 ms of device time per iteration here, 12 ms on the reference's host)
    state = {"perm": torch.randperm(packed.shape[0], device="cuda"), "i_batch": 0}

    def one(i):
        if state["i_batch"] + a.n_rand > packed.shape[0]:
            state["perm"], state["i_batch"] = torch.randperm(packed.shape[0], device="cuda"), 0
        idx = state["perm"][state["i_batch"]:state["i_batch"] + a.n_rand]
        state["i_batch"] += a.n_rand
        r = packed[idx]
        target = torch.rand((a.n_rand, 3), device="cuda")
        out = N.train_on_batch(800, 800, K, (r[:, 0:3], r[:, 3:6]), target, opt, **kw)
        opt.param_groups[0]['lr'] = lrate * (0.1 ** (i / (lrate_decay * 1000)))     # nerf.ipynb:1278-1282
        return out

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        out = one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evals = a.n_rand * (a.samples + (a.samples + a.importance if a.importance else 0))
    flops = evals * 1186816 * 3          # forward + dX + dW
    # the step's big kernels by kind (HIP events on the step's stream, a short second run: not inside the reported it/s)
    spans = {}
    if spans_wanted:
        ctx = N.get_context()
        ctx.profile_enable(True)
        ctx.profile_read_train(reset=True)
        for i in range(10):
            one(i)
        torch.cuda.synchronize()
        ctx.profile_enable(False)
        spans = {k: {"ms_per_iter": v[0] / 10, "launches_per_iter": v[1] / 10} for k, v in ctx.profile_read_train(reset=True).items()}
    return {"metric": "train_iterations_per_sec", "value": iters / dt, "unit": "it/s", "kernels": spans,
            "ms_per_iter": dt / iters * 1e3, "n_rand": a.n_rand, "N_samples": a.samples,
            "N_importance": a.importance, "mlp_evals_per_iter": evals,
            "approx_tflops": flops * iters / dt / 1e12, "final_loss": float(out["loss"]),
            "forward_arithmetic": ("f32" if os.environ.get("NERF_TRAIN_FORWARD", "").lower().startswith("f3") or
                                   N.get_context().get_precision() != "f16x2" else "f16x2 (fp16-pair kernel)"),
            "reference_stored_run_it_per_s": "5.6-7.4 (ship 96+192, unknown CUDA GPU; BASELINE.md)"}


if __name__ == "__main__":
    main()
