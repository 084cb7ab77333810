"""Adversarial precision probes for the fp16-pair kernel (prints rms/max error vs fp64 for f32 and f16x2)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import nerf_projects_amd as N
from nerf_projects_amd import synthetic
from test_hip_parity import _forward_fp64, make_net

def run(name, sd, x, arch=dict(D=8, skips=[4], use_viewdirs=True, output_ch=4)):
    net = make_net(N, sd, **arch)
    want = _forward_fp64(sd, x, arch["D"], arch["skips"], arch["use_viewdirs"])
    scale = np.abs(want).max(0)
    ctx = N.get_context(); out = {}
    ctx.precision_status(reset=True)
    for p in ("f32", "f16x2"):
        ctx.set_precision(p)
        e = np.abs(net(x).cpu().numpy().astype(np.float64) - want) / scale
        out[p] = (np.sqrt((e ** 2).mean()), e.max())
    st = ctx.precision_status()
    print(f"{name:45s} f32 rms {out['f32'][0]:.2e} max {out['f32'][1]:.2e} | f16x2 rms {out['f16x2'][0]:.2e} max {out['f16x2'][1]:.2e}"
          f" | ratio rms {out['f16x2'][0]/out['f32'][0]:.2f} max {out['f16x2'][1]/out['f32'][1]:.2f} status {st}")

torch.manual_seed(5)
x = torch.rand(2048, 90, device="cuda") * 2 - 1
base = dict(synthetic.synthetic_state_dict(7))
run("plain", dict(base), x)
for k in (8, 13, 16, 20):
    sd = dict(base); w = np.asarray(sd["pts_linears.2.weight"]).copy()
    # one weight per row 2^k above the rest of its row (small weights in a row of large ones): every 8th column
    w[:, ::8] *= np.float32(2.0 ** k); sd["pts_linears.2.weight"] = w
    # keep activations sane: shrink the columns' inputs? no - leave, fp64 is the yardstick
    run(f"rows: 1/8 of weights x2^{k}", sd, x)
for k in (8, 13, 16, 20):
    sd = dict(base); w = np.asarray(sd["pts_linears.2.weight"]).copy()
    w[5, :] *= np.float32(2.0 ** k)                       # one ROW huge: all other rows 2^-k below the layer max
    w3 = np.asarray(sd["pts_linears.3.weight"]).copy(); w3[:, 5] = 0.0          # ... and its output is a dead end
    sd["pts_linears.2.weight"] = w; sd["pts_linears.3.weight"] = w3
    run(f"one row x2^{k}, dead-end output", sd, x)
for k in (8, 13, 16, 20):
    sd = dict(base); w = np.asarray(sd["pts_linears.2.weight"]).copy()
    w[5, :] *= np.float32(2.0 ** k)
    sd["pts_linears.2.weight"] = w
    run(f"one row x2^{k}, output used", sd, x)
