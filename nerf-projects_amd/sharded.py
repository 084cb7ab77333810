"""Multi-GPU frame rendering: static ray partition + one gather at frame end.

The reference's ``nerf/`` path is single-device. Rays are
independent and cost the same (fixed 64+128 samples, no early termination), so the flat
``[H*W]`` ray index is cut into ``world_size`` contiguous, equally sized shards; each
rank (one process per GPU, ``torch.distributed`` with the ``nccl`` = RCCL backend over
xGMI) renders its shard with no data-path collective, and the only exchange is a gather
of ``rgb|disp|acc`` (20 B/ray, 1.6 MB per GPU for an 800x800 frame) to rank 0.
Weights (4.8 MB) are loaded by every rank from the same state dict.

RCCL between the processes of one node shares device buffers by IPC handles; the host driver of the MI355X pool this
was built on supports only the dmabuf form, which HSA selects when ``HSA_ENABLE_IPC_MODE_LEGACY=0`` is in the environment
at the first GPU call of the process. ``ensure_ipc_env()`` puts it there (never overriding a value the user set); it runs
when this module is imported - importing the package does not touch the GPU - so library users get the same environment
as ``bench.py`` in either launch form.
"""
import os

import torch
import torch.distributed as dist


def ensure_ipc_env():
    """``HSA_ENABLE_IPC_MODE_LEGACY=0`` unless the caller chose a value. Must precede the process's first GPU call to have
    any effect (HSA reads it once); returns the value in force."""
    return os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


ensure_ipc_env()


def shard_bounds(n, world_size, rank):
    """Contiguous partition of ``range(n)``: the first ``n % world_size`` ranks get one extra."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_frame(local, n_total, group=None, dst=0, force_collective=False):
    """Gather per-ray outputs ``{name: [n_local, ...]}`` of every rank to ``dst``.

    All fields are packed into one ``[n_max, C]`` fp32 buffer per rank (padded to the
    largest shard so a plain ``gather`` applies): one collective per frame. Returns the
    ``{name: [n_total, ...]}`` dict on ``dst`` and ``None`` elsewhere. ``force_collective``
    issues the gather even in a one-rank group (used to exercise the RCCL path on one GPU).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    names = sorted(local)
    widths = [int(torch.Size(local[k].shape[1:]).numel()) for k in names]      # 1 for [n] fields; works for empty shards
    shapes = [tuple(local[k].shape[1:]) for k in names]
    n_local = local[names[0]].shape[0]
    n_max = -(-n_total // world)
    dev = local[names[0]].device
    buf = torch.zeros((n_max, sum(widths)), device=dev, dtype=torch.float32)
    col = 0
    for k, w in zip(names, widths):
        buf[:n_local, col:col + w] = local[k].reshape(n_local, w)
        col += w
    if world == 1 and not force_collective:
        parts = [buf]
    else:
        dst_global = dist.get_global_rank(group, dst) if group is not None else dst
        # RCCL (backend "nccl") gathers device buffers; gloo - used to rehearse the N > 1 path where the ranks cannot
        # each have a GPU of their own - only moves host memory, so the packed buffer takes a detour through the host
        via_host = buf.is_cuda and dist.get_backend(group) == "gloo"
        send = buf.cpu() if via_host else buf
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst_global, group=group)
        if via_host and parts is not None:
            parts = [p.to(dev) for p in parts]
    if rank != dst:
        return None
    out = {}
    col = 0
    for k, w, shp in zip(names, widths, shapes):
        pieces = []
        for r in range(world):
            lo, hi = shard_bounds(n_total, world, r)
            pieces.append(parts[r][: hi - lo, col:col + w])
        out[k] = torch.cat(pieces, 0).reshape((n_total,) + shp)
        col += w
    return out


def _empty_outputs(shard, keys, kwargs):
    """What ``batchify_rays`` would return for zero rays if it knew the per-ray shapes (it returns ``{}``)."""
    widths = {"rgb_map": (3,), "rgb0": (3,)}
    return {k: torch.zeros((0,) + widths.get(k, ()), dtype=torch.float32, device=shard.device) for k in keys}


def render_sharded(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
                   c2w_staticcam=None, group=None, keys=("rgb_map", "disp_map", "acc_map"), render_chunks=None,
                   **kwargs):
    """``render()`` for one frame across the ranks of ``group``.

    Every rank builds the ray record of its own contiguous shard, renders it in ``chunk``-ray
    pieces and contributes to one gather. Rank 0 returns ``[rgb, disp, acc, {}]`` reshaped to
    the ray grid (the reference's ``render`` return shape); other ranks return ``None``.
    ``render_chunks(rays_shard, chunk, **kwargs) -> dict`` defaults to ``batchify_rays``.
    """
    from . import host
    if render_chunks is None:
        render_chunks = host.batchify_rays
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = None
    if 'network_fn' in kwargs and hasattr(kwargs['network_fn'], 'parameters'):
        device = next(kwargs['network_fn'].parameters()).device
    fused = (render_chunks is host.batchify_rays and c2w is not None and host._frame_call_applies(kwargs)
             and kwargs['network_query_fn'].matches(kwargs['network_fn'], bool(use_viewdirs))
             and all(k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std") for k in keys))
    if fused:
        # one C call per rank and frame (nerf_render_shard): the rank generates and renders only its own pixels
        n_total, sh = int(H) * int(W), (H, W, 3)
        ret = host.render_shard(H, W, K, world, rank, chunk, c2w=c2w, ndc=ndc, near=near, far=far,
                                use_viewdirs=use_viewdirs, c2w_staticcam=c2w_staticcam, **kwargs)
    else:
        if c2w is not None and isinstance(kwargs.get('network_fn'), host.NeRF):
            # each rank generates only its own shard of the frame's rays, on its GPU
            n_total, sh = int(H) * int(W), (H, W, 3)
            lo, hi = shard_bounds(n_total, world, rank)
            shard = host.generate_rays(H, W, K, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, first_pixel=lo,
                                       n_pixels=hi - lo, device=device)
        else:
            packed, sh = host.pack_rays(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, device=device)
            n_total = packed.shape[0]
            lo, hi = shard_bounds(n_total, world, rank)
            shard = packed[lo:hi]
        ret = render_chunks(shard, chunk, **kwargs)
        if not ret:                                  # a rank with an empty shard (n_total < world): shaped empties
            ret = _empty_outputs(shard, keys, kwargs)
    local = {k: ret[k] for k in keys}
    if world == 1:
        full = local
    else:
        full = gather_frame(local, n_total, group)
    if full is None:
        return None
    grid = list(sh[:-1])
    return [torch.reshape(full[k], grid + list(full[k].shape[1:])) for k in keys] + [{}]
