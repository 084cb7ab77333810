#!/usr/bin/env python3
"""Per-shard render time of the N-GPU partition, measured on ONE GPU.

No 8-GPU node is available to the builder, so this is a PROJECTION, labelled as such: for world = 1, 2, 4, 8 every
rank's shard of the frame (shard_bounds = nerf_shard_bounds) is rendered by one nerf_render_shard call on this GPU and
timed; the projected N-GPU frame time is the slowest shard plus the measured cost of packing the [n, 5] gather buffer
(the collective itself - 1.6 MB per GPU over xGMI - is not measurable here and is priced from the link rate).
projected_efficiency = T(world 1) / (N * T_slowest_shard(N)). Run on the GPU box:  python profiles/shard_projection.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_projects_amd as N  # noqa: E402
from nerf_projects_amd import synthetic  # noqa: E402

XGMI_LINK_GBS = 153.0      # MI355X_MICROARCH.md: one xGMI link, per direction


def main():
    sd_c, sd_f = synthetic.synthetic_pair(0)
    mk = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    net_c, net_f = N.NeRF(**mk).load_state_dict(sd_c), N.NeRF(**mk).load_state_dict(sd_f)
    q = N.make_network_query_fn(N.get_embedder(10, 0)[0], N.get_embedder(4, 0)[0])
    out = {"note": "projection from one GPU: shards rendered one after another on the same device; no collective, "
                   "no inter-GPU effects", "precision": N.get_context().get_precision(), "workloads": {}}
    for name, (H, W, ndc, white, cam) in {"C3_lego_800x800_64c+128f": (800, 800, False, True, synthetic.lego_camera),
                                          "C4_fern_1008x756_ndc_64c+128f": (756, 1008, True, False, synthetic.fern_camera)}.items():
        K, c2w, near, far = cam(H, W)
        kw = dict(network_fn=net_c, network_query_fn=q, N_samples=64, N_importance=128, network_fine=net_f,
                  white_bkgd=white, perturb=0., raw_noise_std=0.)
        camkw = dict(c2w=c2w, ndc=ndc, near=near, far=far, use_viewdirs=True)
        rows = {}
        t1 = None
        for world in (1, 2, 4, 8):
            per_rank = []
            for rank in range(world):
                N.render_shard(H, W, K, world, rank, chunk=32768, **camkw, **kw)       # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    ret = N.render_shard(H, W, K, world, rank, chunk=32768, **camkw, **kw)
                    buf = torch.cat([ret["rgb_map"], ret["disp_map"][:, None], ret["acc_map"][:, None]], 1)
                torch.cuda.synchronize()
                per_rank.append((time.perf_counter() - t0) / reps * 1e3)
            n_shard = N.shard_bounds(H * W, world, 0)[1]
            slowest = max(per_rank)
            if world == 1:
                t1 = slowest
            gather_us = n_shard * 20 / (XGMI_LINK_GBS * 1e9) * 1e6 * (world > 1)     # one link into the root, lower bound
            rows[str(world)] = {"rays_per_shard": n_shard, "chunks_per_shard": -(-n_shard // 32768),
                                "last_chunk_rays": n_shard - (n_shard - 1) // 32768 * 32768,
                                "ms_per_shard_by_rank": [round(t, 3) for t in per_rank], "ms_slowest_shard": round(slowest, 3),
                                "gather_wire_time_us_estimate": round(gather_us, 1),
                                "projected_rays_per_s": H * W / (slowest * 1e-3),
                                "projected_efficiency": t1 / (world * slowest)}
            print(name, world, rows[str(world)], flush=True)
        out["workloads"][name] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
