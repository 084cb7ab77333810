"""CPU oracle: numpy fp32 restatement of the reference NeRF ray-chunk renderer.

TEST INFRASTRUCTURE ONLY. This module is the *checker* for the HIP path; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it. Nothing under ``nerf-projects_amd/`` imports it and
the product path raises if its HIP library is missing - it never falls back here.

Pinned: yes. ``tests/golden/make_golden.py`` imports the reference itself
(``nerf/nerf.py``, ``nerf/embedder.py``, ``nerf/nerf_helpers.py`` and notebook
cells 8-12/15 of ``nerf/nerf.ipynb``) in the build container and stores its
outputs as fixtures; ``tests/test_oracle_golden.py`` checks every function here
against them (the reference holds no tests or golden vectors of its own for
this path).

Every function cites the reference lines it restates. ``nerf.ipynb:N`` is line N
of the raw notebook JSON (SURVEY.md citation convention). All arithmetic is
fp32 (``torch.set_default_dtype(torch.float32)``, nerf.ipynb:76); op order
follows the reference wherever order is observable at 1e-6.
"""
import numpy as np

F32 = np.float32

# GEMM backend of NeRF._linear. "numpy" (default: the pinned checker, whatever BLAS numpy links) or "torch":
# torch.addmm on CPU tensors sharing the numpy buffers - the same y = x W^T + b through PyTorch's CPU sgemm, i.e. the
# BLAS the reference itself runs on. Used by bench.py's cpu_baseline leg only, so that the reported CPU rate is not
# held back by numpy's OpenBLAS build (2.5x slower here at equal threads); tests/test_oracle_golden.py checks that
# the two backends agree to rounding.
_GEMM = {"backend": "numpy"}


def set_gemm_backend(name):
    if name not in ("numpy", "torch"):
        raise ValueError(name)
    _GEMM["backend"] = name


# ----------------------------------------------------------------------------------------------
# helpers that restate torch semantics the reference relies on
# ----------------------------------------------------------------------------------------------

def linspace_f32(start, end, steps):
    """``torch.linspace(start, end, steps)`` for fp32 on CPU.

    ATen fills the first half as ``start + step*i`` and the second half as
    ``end - step*(steps-1-i)``, each with a single rounding (fused multiply-add),
    ``step = (end-start)/(steps-1)`` in fp32. Checked bit-for-bit against
    torch 2.10 for the sizes the path uses (tests/test_oracle_golden.py).
    """
    start, end = F32(start), F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps, dtype=np.float64)
    lo = np.float64(start) + np.float64(step) * i
    hi = np.float64(end) - np.float64(step) * (steps - 1 - i)
    return np.where(i < steps // 2, lo, hi).astype(F32)


def _sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


# ----------------------------------------------------------------------------------------------
# R3  positional encoding (nerf/embedder.py:8-116)
# ----------------------------------------------------------------------------------------------

def get_embedder(multires, i=0):
    """Returns ``(embed_fn, out_dim)`` (nerf/embedder.py:82-116).

    gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]
    in 3-wide blocks, identity first (embedder.py:28-34, 54-65, 80); frequencies
    ``2.**linspace(0, L-1, L)`` are exact powers of two (embedder.py:48).
    ``i == -1`` is the identity with out_dim 3 (embedder.py:89-92).
    """
    if i == -1:
        return (lambda x: np.asarray(x, dtype=F32)), 3
    freqs = [F32(2.0 ** k) for k in range(multires)]

    def embed(x):
        x = np.asarray(x, dtype=F32)
        parts = [x]
        for f in freqs:
            xf = x * f
            parts.append(np.sin(xf))
            parts.append(np.cos(xf))
        return np.concatenate(parts, axis=-1).astype(F32, copy=False)

    return embed, 3 + 6 * multires


# ----------------------------------------------------------------------------------------------
# R5  the field MLP (nerf/nerf.py:8-111)
# ----------------------------------------------------------------------------------------------

class NeRF:
    """Forward-only restatement of the reference ``NeRF`` module (nerf/nerf.py:8-111).

    Holds a state-dict-shaped mapping of fp32 arrays; ``nn.Linear`` is
    ``y = x W^T + b`` with W stored [out, in].
    """

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=(4,),
                 use_viewdirs=False, state_dict=None):
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.output_ch = output_ch
        self.skips = tuple(skips)
        self.use_viewdirs = use_viewdirs
        self.sd = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd):
        self.sd = {k: np.ascontiguousarray(np.asarray(v), dtype=F32) for k, v in sd.items()}

    def state_dict(self):
        return self.sd

    def _linear(self, name, x):
        if _GEMM["backend"] == "torch":
            import torch
            xt = torch.from_numpy(np.ascontiguousarray(x, dtype=F32))
            y = torch.addmm(torch.from_numpy(self.sd[name + ".bias"]), xt, torch.from_numpy(self.sd[name + ".weight"]).t())
            return y.numpy()
        return (x @ self.sd[name + ".weight"].T + self.sd[name + ".bias"]).astype(F32, copy=False)

    def __call__(self, x):
        return self.forward(x)

    def forward(self, x):
        x = np.asarray(x, dtype=F32)
        input_pts = x[..., : self.input_ch]                                   # nerf.py:64
        input_views = x[..., self.input_ch: self.input_ch + self.input_ch_views]
        h = input_pts
        for i in range(self.D):                                               # nerf.py:70-80
            h = np.maximum(self._linear(f"pts_linears.{i}", h), F32(0))
            if i in self.skips:
                h = np.concatenate([input_pts, h], -1)                        # input first
        if self.use_viewdirs:
            alpha = self._linear("alpha_linear", h)                           # nerf.py:86
            feature = self._linear("feature_linear", h)                       # nerf.py:89 (no ReLU)
            h = np.concatenate([feature, input_views], -1)                    # nerf.py:93
            h = np.maximum(self._linear("views_linears.0", h), F32(0))        # nerf.py:96-98
            rgb = self._linear("rgb_linear", h)                               # nerf.py:101
            return np.concatenate([rgb, alpha], -1)                           # nerf.py:106
        return self._linear("output_linear", h)                               # nerf.py:109


# ----------------------------------------------------------------------------------------------
# R4  batchify / run_network (nerf.ipynb:224-244, 790-855)
# ----------------------------------------------------------------------------------------------

def batchify(fn, chunk):
    """nerf.ipynb:224-244."""
    if chunk is None:
        return fn

    def ret(inputs):
        return np.concatenate([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """nerf.ipynb:790-855: flatten, encode xyz, broadcast+encode dirs per sample,
    concat [gamma(xyz) | gamma(dir)], MLP in ``netchunk`` slices, reshape back."""
    inputs = np.asarray(inputs, dtype=F32)
    inputs_flat = inputs.reshape(-1, inputs.shape[-1])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = np.broadcast_to(np.asarray(viewdirs, dtype=F32)[:, None], inputs.shape)
        input_dirs_flat = input_dirs.reshape(-1, input_dirs.shape[-1])
        embedded = np.concatenate([embedded, embeddirs_fn(input_dirs_flat)], -1)
    outputs_flat = batchify(fn, netchunk)(embedded)
    return outputs_flat.reshape(list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


# ----------------------------------------------------------------------------------------------
# R6  raw2outputs (nerf.ipynb:254-349)
# ----------------------------------------------------------------------------------------------

def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, noise=None):
    """Sigma->alpha compositing. ``noise`` (optional, [N,S]) injects the additive
    sigma noise explicitly; ``pytest=True`` reproduces the reference's
    ``np.random.seed(0); np.random.rand`` path (nerf.ipynb:322-325)."""
    raw = np.asarray(raw, dtype=F32)
    z_vals = np.asarray(z_vals, dtype=F32)
    rays_d = np.asarray(rays_d, dtype=F32)
    dists = z_vals[..., 1:] - z_vals[..., :-1]                                      # :295
    dists = np.concatenate([dists, np.full(dists[..., :1].shape, 1e10, dtype=F32)], -1)   # :300
    norm = np.sqrt(np.sum(rays_d * rays_d, axis=-1, dtype=F32)).astype(F32)
    dists = dists * norm[..., None]                                                 # :305
    rgb = _sigmoid(raw[..., :3])                                                    # :308
    try:                                                                            # :312-315
        noise_std = float(raw_noise_std)
    except (TypeError, ValueError):
        noise_std = 0.0
    nz = F32(0.0)
    if noise is not None:
        nz = np.asarray(noise, dtype=F32)
    elif noise_std > 0.0:
        if pytest:
            np.random.seed(0)
            nz = (np.random.rand(*raw[..., 3].shape) * noise_std).astype(F32)       # :322-325
        else:
            nz = (np.random.randn(*raw[..., 3].shape) * noise_std).astype(F32)      # :319
    sigma = np.maximum(raw[..., 3] + nz, F32(0))
    alpha = (F32(1.0) - np.exp(-sigma * dists)).astype(F32)                         # :291, :328
    trans = np.concatenate([np.ones((alpha.shape[0], 1), F32),
                            (F32(1.0) - alpha) + F32(1e-10)], -1)
    # torch.cumprod on CPU accumulates fp32 inputs in double (ATen acc_type<float,false>)
    # and rounds every prefix to fp32.
    weights = (alpha * np.cumprod(trans, -1, dtype=np.float64).astype(F32)[:, :-1]).astype(F32)   # :329
    rgb_map = np.sum(weights[..., None] * rgb, axis=-2, dtype=F32)                  # :332
    depth_map = np.sum(weights * z_vals, axis=-1, dtype=F32)                        # :335
    acc_map = np.sum(weights, axis=-1, dtype=F32)                                   # :343
    denom = np.maximum(F32(1e-10), acc_map)                                         # :339
    disp_map = (F32(1.0) / np.maximum(depth_map / denom, F32(1e-10))).astype(F32)   # :340
    if white_bkgd:
        rgb_map = rgb_map + (F32(1.0) - acc_map[..., None])                         # :346-347
    return rgb_map.astype(F32), disp_map, acc_map, weights, depth_map


# ----------------------------------------------------------------------------------------------
# R7  sample_pdf (nerf/nerf_helpers.py:372-439)
# ----------------------------------------------------------------------------------------------

def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """Inverse-CDF sampling; ``u`` (optional [N,N_samples]) injects the uniforms."""
    bins = np.asarray(bins, dtype=F32)
    weights = np.asarray(weights, dtype=F32) + F32(1e-5)                            # :396
    pdf = weights / np.sum(weights, -1, keepdims=True, dtype=F32)                   # :397
    # torch.cumsum on CPU: sequential, accumulated in double, each prefix rounded to fp32
    cdf = np.cumsum(pdf, -1, dtype=np.float64).astype(F32)                          # :398
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)                    # :400
    shape = list(cdf.shape[:-1]) + [N_samples]
    if u is not None:
        u = np.asarray(u, dtype=F32)
    elif pytest:                                                                    # :410-418
        np.random.seed(0)
        if det:
            u = np.broadcast_to(np.linspace(0.0, 1.0, N_samples), shape).astype(F32)
        else:
            u = np.random.rand(*shape).astype(F32)
    elif det:
        u = np.broadcast_to(linspace_f32(0.0, 1.0, N_samples), shape)               # :404-405
    else:
        u = np.random.rand(*shape).astype(F32)                                      # :407
    u = np.ascontiguousarray(u, dtype=F32)
    M = cdf.shape[-1]
    # searchsorted(cdf, u, right=True): number of cdf entries <= u              :423
    inds = np.sum(cdf[:, None, :] <= u[:, :, None], axis=-1).astype(np.int64)
    below = np.maximum(0, inds - 1)                                                 # :424
    above = np.minimum(M - 1, inds)                                                 # :425
    cdf_lo = np.take_along_axis(cdf, below, -1)                                     # :429-431
    cdf_hi = np.take_along_axis(cdf, above, -1)
    bin_lo = np.take_along_axis(bins, below, -1)
    bin_hi = np.take_along_axis(bins, above, -1)
    denom = cdf_hi - cdf_lo                                                         # :434
    denom = np.where(denom < F32(1e-5), F32(1.0), denom)                            # :435
    t = (u - cdf_lo) / denom                                                        # :436
    return (bin_lo + t * (bin_hi - bin_lo)).astype(F32)                             # :437


def sample_pdf_tolerance(bins, weights, u, eps=5e-7, knot_tol=2.5e-7):
    """Conditioning of ``sample_pdf`` for parity tests: ``(tol, mask, bin_lo, bin_hi)``.

    Inverse-CDF sampling amplifies rounding: a perturbation d of the fp32 CDF (a few
    ulps of an O(1) number; the reference's ``torch.sum`` order is even host-SIMD
    dependent) moves a sample by ``d * bin_width / bin_mass``. ``tol`` is that bound
    with ``d = eps`` per sample, evaluated in fp64.

    Two genuine discontinuities get ``mask=True`` and are only required to stay
    inside ``[bin_lo, bin_hi]`` (one bin either side of the bracketing one):
    * ``denom < 1e-5 -> 1`` (nerf_helpers.py:435): a bin whose mass is within
      ``knot_tol`` of 1e-5 takes either branch;
    * a sample whose ``u`` is within ``knot_tol`` of a CDF knot next to a degenerate
      (mass < 1e-5) bin jumps across that bin (includes det ``u = 1.0`` vs
      ``cdf[-1]`` rounding above or below 1.0).
    """
    bins = np.asarray(bins, dtype=np.float64)
    w = np.asarray(weights, dtype=F32).astype(np.float64) + np.float64(F32(1e-5))
    pdf = w / np.sum(w, -1, keepdims=True)
    cdf = np.concatenate([np.zeros_like(pdf[..., :1]), np.cumsum(pdf, -1)], -1)
    u = np.asarray(u, dtype=np.float64)
    M = cdf.shape[-1]
    inds = np.sum(cdf[:, None, :] <= u[:, :, None], axis=-1)
    below = np.maximum(0, inds - 1)
    above = np.minimum(M - 1, inds)
    denom = np.take_along_axis(cdf, above, -1) - np.take_along_axis(cdf, below, -1)
    mass = np.diff(cdf, axis=-1)                                # [N, M-1]
    tol_bin = eps * np.diff(bins, axis=-1) / np.maximum(mass, 1e-5)
    tol = np.zeros_like(u)
    for off in (-1, 0, 1):                                      # bracketing bin and its neighbours
        tol = np.maximum(tol, np.take_along_axis(tol_bin, np.clip(below + off, 0, M - 2), -1))
    degenerate = mass < 1e-5 + knot_tol
    mask = np.abs(denom - 1e-5) < knot_tol
    near_knot = (np.abs(u - np.take_along_axis(cdf, below, -1)) < knot_tol) | \
                (np.abs(u - np.take_along_axis(cdf, above, -1)) < knot_tol)
    nb = np.zeros_like(mask)
    for off in (-1, 0, 1):
        nb |= np.take_along_axis(degenerate, np.clip(below + off, 0, M - 2), -1)
    mask |= near_knot & nb
    bin_lo = np.take_along_axis(bins, np.maximum(0, below - 1), -1)
    bin_hi = np.take_along_axis(bins, np.minimum(M - 1, above + 1), -1)
    return tol, mask, bin_lo, bin_hi


# ----------------------------------------------------------------------------------------------
# R2 / R8 / R9  render_rays (nerf.ipynb:359-492)
# ----------------------------------------------------------------------------------------------

def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False,
                perturb=0., N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.,
                verbose=False, pytest=False, _extras=None, _inject=None):
    """The per-ray-chunk renderer. ``_extras`` (a dict) receives intermediates
    (coarse/fine z_vals, weights); ``_inject`` may carry explicit random arrays
    ``t_rand`` [N,S_c], ``u`` [N,S_i], ``noise0`` [N,S_c], ``noise`` [N,S_c+S_i]
    so that perturbed paths can be compared without sharing an RNG, and ``z_fine`` [N,S_c+S_i]: fine depths to
    evaluate the fine pass at (in place of the merged ones; z_std still describes the own samples)."""
    inj = _inject or {}
    ray_batch = np.asarray(ray_batch, dtype=F32)
    N_rays = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]                           # :410
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None               # :413
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]                                # :414-415

    t_vals = linspace_f32(0., 1., N_samples)                                        # :418
    if not lindisp:
        z_vals = near * (F32(1.) - t_vals) + far * t_vals                           # :421
    else:
        z_vals = F32(1.) / (F32(1.) / near * (F32(1.) - t_vals) + F32(1.) / far * t_vals)   # :424
    z_vals = np.broadcast_to(z_vals, (N_rays, N_samples)).astype(F32)

    if perturb > 0.:                                                                # :428-444
        mids = F32(.5) * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = np.concatenate([mids, z_vals[..., -1:]], -1)
        lower = np.concatenate([z_vals[..., :1], mids], -1)
        if "t_rand" in inj:
            t_rand = np.asarray(inj["t_rand"], dtype=F32)
        elif pytest:
            np.random.seed(0)
            t_rand = np.random.rand(*z_vals.shape).astype(F32)
        else:
            t_rand = np.random.rand(*z_vals.shape).astype(F32)
        z_vals = (lower + (upper - lower) * t_rand).astype(F32)

    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]        # :447
    raw = network_query_fn(pts, viewdirs, network_fn)                               # :451
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(
        raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest, noise=inj.get("noise0"))
    if _extras is not None:
        _extras.update(z_coarse=z_vals, weights_coarse=weights, raw_coarse=raw, depth0=depth_map)

    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        z_vals_mid = F32(.5) * (z_vals[..., 1:] + z_vals[..., :-1])                 # :460
        z_samples = sample_pdf(z_vals_mid, weights[..., 1:-1], N_importance,
                               det=(perturb == 0.), pytest=pytest, u=inj.get("u"))  # :462
        z_vals = np.sort(np.concatenate([z_vals, z_samples], -1), -1)               # :467
        if "z_fine" in inj:                      # parity tests: evaluate the fine pass at given depths
            z_vals = np.ascontiguousarray(inj["z_fine"], dtype=F32)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]    # :468
        run_fn = network_fn if network_fine is None else network_fine               # :471
        raw = network_query_fn(pts, viewdirs, run_fn)                               # :473
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(
            raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest, noise=inj.get("noise"))
        if _extras is not None:
            _extras.update(z_samples=z_samples, z_fine=z_vals, weights_fine=weights, depth=depth_map)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}           # :477
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'] = rgb_map_0
        ret['disp0'] = disp_map_0
        ret['acc0'] = acc_map_0
        m = np.mean(z_samples, -1, keepdims=True, dtype=F32)
        ret['z_std'] = np.sqrt(np.mean((z_samples - m) ** 2, -1, dtype=F32)).astype(F32)   # :486
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """nerf.ipynb:514-548."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: np.concatenate(all_ret[k], 0) for k in all_ret}


# ----------------------------------------------------------------------------------------------
# R0  ray generation and packing (nerf_helpers.py:222-369, nerf.ipynb:558-640)
# ----------------------------------------------------------------------------------------------

def get_rays(H, W, K, c2w):
    """nerf_helpers.py:222-296: integer pixel coordinates (no +0.5), camera looks
    down -z, ``rays_d = R @ dirs`` (not normalised), ``rays_o = t``. Pixel grids come
    from ``torch.linspace(0, W-1, W)`` (exact integers in fp32)."""
    c2w = np.asarray(c2w, dtype=F32)
    i, j = np.meshgrid(linspace_f32(0, W - 1, W), linspace_f32(0, H - 1, H), indexing='xy')
    # K may hold python/numpy float64 scalars; torch keeps the fp32 tensor dtype.
    dirs = np.stack([(i - F32(K[0][2])) / F32(K[0][0]),
                     -(j - F32(K[1][2])) / F32(K[1][1]),
                     -np.ones_like(i)], -1).astype(F32)
    prod = dirs[..., None, :] * c2w[:3, :3]
    rays_d = ((prod[..., 0] + prod[..., 1]) + prod[..., 2]).astype(F32)             # torch.sum over 3
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """nerf_helpers.py:311-369."""
    rays_o = np.asarray(rays_o, dtype=F32)
    rays_d = np.asarray(rays_d, dtype=F32)
    near = F32(near)
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    cw = F32(-1. / (W / (2. * focal)))
    ch = F32(-1. / (H / (2. * focal)))
    o0 = cw * rays_o[..., 0] / rays_o[..., 2]
    o1 = ch * rays_o[..., 1] / rays_o[..., 2]
    o2 = F32(1.) + F32(2.) * near / rays_o[..., 2]
    d0 = cw * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = ch * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = F32(-2.) * near / rays_o[..., 2]
    return np.stack([o0, o1, o2], -1).astype(F32), np.stack([d0, d1, d2], -1).astype(F32)


def pack_rays(H, W, K, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
              c2w_staticcam=None):
    """The ray record ``render()`` builds before ``batchify_rays`` (nerf.ipynb:596-629):
    ``[o(3), d(3), near, far, viewdir(3)]``; viewdir is the unit world-space direction taken
    before the NDC warp and before the ``c2w_staticcam`` override."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
    rays_o, rays_d = np.asarray(rays_o, dtype=F32), np.asarray(rays_d, dtype=F32)
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
        nrm = np.sqrt(np.sum(viewdirs * viewdirs, -1, keepdims=True, dtype=F32)).astype(F32)
        viewdirs = (viewdirs / nrm).reshape(-1, 3).astype(F32)
    sh = rays_d.shape
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    rays_o = rays_o.reshape(-1, 3).astype(F32)
    rays_d = rays_d.reshape(-1, 3).astype(F32)
    near_c = F32(near) * np.ones_like(rays_d[..., :1])
    far_c = F32(far) * np.ones_like(rays_d[..., :1])
    packed = np.concatenate([rays_o, rays_d, near_c, far_c], -1)
    if use_viewdirs:
        packed = np.concatenate([packed, viewdirs], -1)
    return np.ascontiguousarray(packed, dtype=F32), sh


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1.,
           use_viewdirs=False, c2w_staticcam=None, **kwargs):
    """nerf.ipynb:558-640: returns ``[rgb, disp, acc, extras]`` reshaped to the ray grid."""
    packed, sh = pack_rays(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam)
    all_ret = batchify_rays(packed, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = all_ret[k].reshape(list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]


def make_query_fn(embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """The ``network_query_fn`` lambda ``create_nerf`` binds (nerf.ipynb:899-902)."""
    return lambda inputs, viewdirs, network_fn: run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=netchunk)


def calculate_ssim(img1, img2, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """Mean SSIM of two [H,W,3] images, restating nerf_helpers.py:21-111 (tf.image.ssim style):
    separable Gaussian window, zero padding, sigma clamps, mean over all pixels and channels."""
    a = np.clip(np.asarray(img1, dtype=F32), 0, max_val)
    b = np.clip(np.asarray(img2, dtype=F32), 0, max_val)
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    f_i = ((np.arange(filter_size) - hw + shift) / filter_sigma).astype(F32) ** 2           # :59
    filt = np.exp(F32(-0.5) * f_i).astype(F32)
    filt = (filt / np.sum(filt, dtype=F32)).astype(F32)                                     # :60-62

    def conv(z, axis):                                                                      # zero-padded 'same'
        pad = [(0, 0)] * z.ndim
        pad[axis] = (hw, hw)
        zp = np.pad(z, pad)
        out = np.zeros_like(z)
        for k in range(filter_size):
            sl = [slice(None)] * z.ndim
            sl[axis] = slice(k, k + z.shape[axis])
            out = out + filt[k] * zp[tuple(sl)]
        return out.astype(F32)

    def filt_fn(z):                                                                         # :63-73
        return conv(conv(z, 1), 0)

    mu0, mu1 = filt_fn(a), filt_fn(b)
    mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
    s00 = np.maximum(filt_fn(a * a) - mu00, F32(0))                                         # :80-86
    s11 = np.maximum(filt_fn(b * b) - mu11, F32(0))
    s01 = filt_fn(a * b) - mu01
    s01 = np.sign(s01) * np.minimum(np.sqrt(s00 * s11), np.abs(s01))                        # :87-89
    c1, c2 = F32((k1 * max_val) ** 2), F32((k2 * max_val) ** 2)
    numer = (F32(2) * mu01 + c1) * (F32(2) * s01 + c2)
    denom = (mu00 + mu11 + c1) * (s00 + s11 + c2)
    return float(np.mean((numer / denom).astype(np.float64)))                               # :93-99


def calculate_metrics(img1, img2):
    """mse / psnr / ssim of nerf_helpers.py:148-214 (LPIPS needs the absent ``lpips`` package)."""
    a = np.clip(np.asarray(img1, dtype=F32), 0, 1)
    b = np.clip(np.asarray(img2, dtype=F32), 0, 1)
    mse = float(img2mse(a, b))
    return {'mse': mse, 'psnr': float(mse2psnr(mse)), 'ssim': calculate_ssim(img1, img2)}


def img2mse(x, y):
    """nerf_helpers.py:8-11."""
    return np.mean((np.asarray(x, F32) - np.asarray(y, F32)) ** 2, dtype=F32)


def mse2psnr(x):
    """nerf_helpers.py:12-14."""
    return F32(-10.) * np.log(F32(x)) / np.log(F32(10.))


# ---- random-ray batching of train() (nerf/nerf.ipynb cell 19; "train:N" = line N of the cell source) ------------
def get_rays_np(H, W, K, c2w):
    """nerf_helpers.py:301-308 (the numpy twin of get_rays the global-batching mode uses)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def global_ray_batches(images, poses, H, W, K, i_train, N_rand, n_batches, rng):
    """use_batching mode (train:190-203, 230-242): all rays of the training images, shuffled once with the numpy
    stream, consumed in consecutive windows. Returns a list of (batch_rays [2,N,3], target_s [N,3]).
    (The epoch-end ``torch.randperm`` reshuffle is not restated: it needs torch's RNG.)"""
    rays = np.stack([np.stack(get_rays_np(H, W, K, p), 0) for p in poses[:, :3, :4]], 0)      # [N, ro+rd, H, W, 3]
    rays_rgb = np.concatenate([rays, images[:, None]], 1)                                     # [N, ro+rd+rgb, H, W, 3]
    rays_rgb = np.transpose(rays_rgb, [0, 2, 3, 1, 4])
    rays_rgb = np.stack([rays_rgb[i] for i in i_train], 0)
    rays_rgb = np.reshape(rays_rgb, [-1, 3, 3]).astype(np.float32)
    rng.shuffle(rays_rgb)
    out, i_batch = [], 0
    for _ in range(n_batches):
        batch = np.transpose(rays_rgb[i_batch:i_batch + N_rand], [1, 0, 2])
        out.append((batch[:2], batch[2]))
        i_batch += N_rand
    return out


def per_image_ray_batch(images, poses, H, W, K, i_train, N_rand, i, precrop_iters, precrop_frac, rng):
    """no_batching mode (train:243-276): random training image, optional centre crop, N_rand distinct pixels."""
    img_i = rng.choice(i_train)
    target = images[img_i]
    rays_o, rays_d = get_rays(H, W, K, poses[img_i, :3, :4])               # train:252 uses the torch get_rays
    if i < precrop_iters:
        dH = int(H // 2 * precrop_frac)
        dW = int(W // 2 * precrop_frac)
        rows = np.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH, dtype=np.float32)
        cols = np.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW, dtype=np.float32)
    else:
        rows, cols = np.linspace(0, H - 1, H, dtype=np.float32), np.linspace(0, W - 1, W, dtype=np.float32)
    coords = np.stack(np.meshgrid(rows, cols, indexing="ij"), -1).reshape(-1, 2)
    select_inds = rng.choice(coords.shape[0], size=[N_rand], replace=False)
    sel = coords[select_inds].astype(np.int64)
    return (np.stack([rays_o[sel[:, 0], sel[:, 1]], rays_d[sel[:, 0], sel[:, 1]]], 0), target[sel[:, 0], sel[:, 1]],
            int(img_i), sel)
