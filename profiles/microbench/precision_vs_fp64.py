import sys, numpy as np, torch
sys.path.insert(0, ".")
import nerf_projects_amd as N
from nerf_projects_amd import synthetic
ctx = N.get_context()
torch.manual_seed(0)

def ref64(sd, x, D, skips, use_viewdirs, input_ch=63):
    g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float64)
    x = x.double().cpu()
    pts, views = x[:, :input_ch], x[:, input_ch:]
    h = pts
    for i in range(D):
        h = torch.relu(h @ g(f"pts_linears.{i}.weight").T + g(f"pts_linears.{i}.bias"))
        if i in skips:
            h = torch.cat([pts, h], -1)
    if use_viewdirs:
        alpha = h @ g("alpha_linear.weight").T + g("alpha_linear.bias")
        feat = h @ g("feature_linear.weight").T + g("feature_linear.bias")
        h = torch.relu(torch.cat([feat, views], -1) @ g("views_linears.0.weight").T + g("views_linears.0.bias"))
        rgb = h @ g("rgb_linear.weight").T + g("rgb_linear.bias")
        return torch.cat([rgb, alpha], -1).numpy()
    return (h @ g("output_linear.weight").T + g("output_linear.bias")).numpy()

def run(name, x, **arch):
    kw = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True); kw.update(arch)
    sd = synthetic.synthetic_state_dict(7, **{k: v for k, v in kw.items() if k != "W"})
    net = N.NeRF(**kw).load_state_dict(sd)
    r = ref64(sd, x, kw["D"], kw["skips"], kw["use_viewdirs"])
    scale = np.abs(r).max(0)
    out = {}
    for p in ("f32", "f16x2"):
        ctx.set_precision(p)
        o = net(x).cpu().numpy().astype(np.float64)
        e = np.abs(o - r) / scale
        out[p] = (np.sqrt((e ** 2).mean()), e.max())
    print(f"{name:22s} rel-to-channel-max err  f32: rms {out['f32'][0]:.2e} max {out['f32'][1]:.2e} | f16x2: rms {out['f16x2'][0]:.2e} max {out['f16x2'][1]:.2e}", flush=True)

x = (torch.rand(4096, 90, device="cuda") * 2 - 1)
run("D1 noview out4", x, D=1, skips=[], use_viewdirs=False, output_ch=4)
run("D2 noview out4", x, D=2, skips=[], use_viewdirs=False, output_ch=4)
run("D3 noview skip0", x, D=3, skips=[0], use_viewdirs=False, output_ch=4)
run("D1 view", x, D=1, skips=[])
run("D8 view noskip", x, D=8, skips=[])
run("D8 view skip4", x)
run("D8 view skip4 x1e-3", x * 1e-3)
run("D8 view skip4 x30", x * 30)
