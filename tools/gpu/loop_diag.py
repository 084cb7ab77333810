"""Per-iteration distance of the HIP training loop from the reference's fp32 run, next to the reference's own fp32-vs-fp64 distance
(tests/golden/train_loop.npz), both arithmetics, both starts. tools/gpu: run on the box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_projects_amd as N
from nerf_projects_amd import synthetic
g, frame = np.load(ROOT + "/tests/golden/train_loop.npz"), np.load(ROOT + "/tests/golden/bench_frame.npz")
ctx = N.get_context()
mk = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
for precision in ("f16x2", "f32"):
    ctx.set_precision(precision)
    for start in ("init", "pair"):
        n_iters, n_rand = int(g[f"{start}.n_iters"]), int(g["n_rand"])
        sd_c, sd_f = (synthetic.default_init_state_dict(11), synthetic.default_init_state_dict(12)) if start == "init" else synthetic.synthetic_pair(0)
        net_c, net_f = N.NeRF(**mk).load_state_dict(sd_c), N.NeRF(**mk).load_state_dict(sd_f)
        opt = N.Adam([net_c, net_f], lr=float(g["lrate"]))
        kw = dict(network_fn=net_c, network_fine=net_f, N_samples=64, N_importance=128, white_bkgd=True, perturb=1.0, raw_noise_std=1.0,
                  pytest=True, ndc=False, use_viewdirs=True, near=2., far=6.)
        rays, target, perm = torch.from_numpy(frame["rays"]).cuda(), torch.from_numpy(g["target"]).cuda(), g["perm"]
        ctx.precision_detail(reset=True)
        rows = []
        for it in range(n_iters):
            lo = (it * n_rand) % len(perm)
            sel = torch.from_numpy(perm[lo:lo + n_rand]).cuda()
            out = N.train_on_batch(800, 800, None, None, target[sel], opt, _packed_rays=rays[sel], **kw)
            opt.param_groups[0]['lr'] = float(g["lrate"]) * (0.1 ** (it / (int(g["lrate_decay"]) * 1000)))
            rows.append((abs(float(out["img_loss"]) - g[f"{start}.img_loss"][it]), abs(g[f"{start}.img_loss"][it] - g[f"{start}.img_loss.f64"][it]),
                         abs(float(out["img_loss0"]) - g[f"{start}.img_loss0"][it]), abs(g[f"{start}.img_loss0"][it] - g[f"{start}.img_loss0.f64"][it])))
        print(precision, start, "events", ctx.precision_detail(reset=True))
        for it, r in enumerate(rows):
            print(f"   it {it:2d}  img_loss: ours-ref32 {r[0]:.2e}  ref32-ref64 {r[1]:.2e}   img_loss0: {r[2]:.2e}  {r[3]:.2e}")
