"""A host that is not Python drives the C ABI (SURVEY.md section 8 b: the boundary is a C library, the Python package is one
binding of it). tests/c_host/render_rays_host.c is plain C over include/nerf_mi355x.h - what a cgo / JNI / N-API stub would
call - built here with gcc against the in-tree library and run as its own process; its render of a chunk of rays must equal
the Python mirror's bit for bit (same library, same kernels, no Python in between)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nerf-projects_amd")


def _keys():
    keys = []
    for i in range(8):
        keys += [f"pts_linears.{i}.weight", f"pts_linears.{i}.bias"]
    return keys + ["views_linears.0.weight", "views_linears.0.bias", "feature_linear.weight", "feature_linear.bias",
                   "alpha_linear.weight", "alpha_linear.bias", "rgb_linear.weight", "rgb_linear.bias"]


def test_plain_c_host_renders_what_the_python_mirror_renders(weights_pair, tmp_path):
    import nerf_projects_amd as N
    import torch
    N.get_context().set_precision("f16x2")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = tmp_path / "render_rays_host"
    subprocess.run(["gcc", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
                    os.path.join(ROOT, "tests", "c_host", "render_rays_host.c"), "-o", str(exe),
                    os.path.join(PKG, "libnerf_mi355x.so"), "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                    "-Wl,-rpath," + PKG, "-Wl,-rpath," + os.path.join(rocm, "lib")], check=True)
    rays = load_golden("render_rays_lego")["rays"].astype(np.float32)            # [256, 11]
    rays = np.concatenate([rays, rays[:77] + np.float32(1e-3)])                 # 333 rays: a ragged last tile
    n = rays.shape[0]
    blob = np.concatenate([np.asarray(sd[k], dtype=np.float32).reshape(-1) for sd in weights_pair for k in _keys()])
    blob.tofile(tmp_path / "weights.bin")
    rays.tofile(tmp_path / "rays.bin")
    r = subprocess.run([str(exe), str(tmp_path / "weights.bin"), str(tmp_path / "rays.bin"), str(n), "64", "128",
                        str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    rgb, disp, acc, rgb0 = out[:3 * n].reshape(n, 3), out[3 * n:4 * n], out[4 * n:5 * n], out[5 * n:].reshape(n, 3)

    kw = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    net_c, net_f = (N.NeRF(**kw).load_state_dict(sd) for sd in weights_pair)
    q = N.make_network_query_fn(N.get_embedder(10, 0)[0], N.get_embedder(4, 0)[0])
    ret = N.render_rays(torch.from_numpy(rays).cuda(), net_c, q, 64, N_importance=128, network_fine=net_f, white_bkgd=True)
    for name, got in (("rgb_map", rgb), ("disp_map", disp), ("acc_map", acc), ("rgb0", rgb0)):
        want = ret[name].cpu().numpy()
        assert np.array_equal(got, want), (name, np.abs(got - want).max())
