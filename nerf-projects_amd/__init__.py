"""MI355X-native NeRF ray-chunk renderer.

Drop-in for the render path of isaacchunn/nerf-projects ``nerf/`` (``render_rays`` /
``raw2outputs`` / ``run_network`` / ``render`` ...): see DESIGN.md and INTEGRATION.md.
Importing the package never touches the GPU; the first call that needs it loads
``libnerf_mi355x.so`` and raises if the library or a gfx950 device is missing.
"""
from . import synthetic  # noqa: F401
from .host import (Adam, NeRF, NetworkQuery, train_on_batch, batchify, batchify_rays, calculate_lpips, calculate_metrics,  # noqa: F401
                   calculate_ssim, create_nerf, generate_rays, get_context, get_embedder, get_rays, get_rays_np,
                   img2mse, load_checkpoint, make_network_query_fn, mse2psnr, ndc_rays, pack_rays, raw2outputs,
                   render, render_path, render_rays, render_shard, run_network, sample_pdf, save_checkpoint, to8b, write_png)
from . import datasets  # noqa: F401
from .datasets import load_blender_data, load_llff_data, read_png  # noqa: F401
from .sharded import ensure_ipc_env, gather_frame, render_sharded, shard_bounds  # noqa: F401
from .batching import RayBatcher  # noqa: F401

__version__ = "0.1.0"
