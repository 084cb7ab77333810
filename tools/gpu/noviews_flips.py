"""Do the fine depths of the no-viewdirs training fixture differ between the two arithmetics (a resampling flip)? tools/gpu."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_projects_amd as N
from nerf_projects_amd import synthetic
g = np.load(ROOT + "/tests/golden/train_step_noviewdirs.npz")
arch = dict(input_ch_views=0, use_viewdirs=False, output_ch=5)
sd_c, sd_f = synthetic.synthetic_state_dict(8, **arch), synthetic.synthetic_state_dict(48, **arch)
mk = dict(D=8, W=256, input_ch=63, skips=[4], **arch)
net_c, net_f = N.NeRF(**mk).load_state_dict(sd_c), N.NeRF(**mk).load_state_dict(sd_f)
q = N.make_network_query_fn(N.get_embedder(10, 0)[0], None)
rays = torch.from_numpy(g["rays"]).cuda()
ctx = N.get_context()
z = {}
for prec in ("f16x2", "f32"):
    ctx.set_precision(prec)
    ex = {}
    N.render_rays(rays, net_c, q, N_samples=64, N_importance=128, network_fine=net_f, white_bkgd=True, perturb=1.0, raw_noise_std=1.0,
                  pytest=True, _extras=ex)
    z[prec] = (ex["z_fine"].cpu().numpy(), ex["weights_coarse"].cpu().numpy())
d = np.abs(z["f16x2"][0] - z["f32"][0])
print("largest |z_fine(f16x2) - z_fine(f32)| per ray:", np.sort(d.max(1))[-6:])
print("rays with a sample moved by more than 1e-3:", int((d.max(1) > 1e-3).sum()), "of", len(d), "; coarse weights differ by at most", np.abs(z["f16x2"][1] - z["f32"][1]).max())
