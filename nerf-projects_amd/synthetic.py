"""Synthetic inputs for tests and benchmarks (no dataset or checkpoint exists offline).

Two things live here, both plain numpy so that the oracle, the tests and the
HIP path all consume byte-identical inputs:

* ``synthetic_state_dict`` - seeded weights with the reference ``NeRF`` module's
  ``state_dict()`` keys and shapes (nerf/nerf.py:32-55), shaped so that the
  field behaves like a trained white-background Blender scene: a random,
  spectrally-decaying density blob inside the unit cube and robustly negative
  sigma outside it (so the far-bound sample, whose ``dists`` is 1e10
  (nerf/nerf.ipynb:300), has a robust sign).
* Blender / LLFF camera helpers: ``pose_spherical`` (nerf/load_blender.py:10-34),
  intrinsics ``K`` (nerf/nerf.ipynb:1088-1094) and the lego focal length
  (nerf/load_blender.py:72-73).

``numpy.random.RandomState`` is the legacy generator and is bit-stable across
numpy versions, so fixtures never store weights, only seeds.
"""
from collections import OrderedDict

import numpy as np

LEGO_CAMERA_ANGLE_X = 0.6911112070083618  # transforms_*.json of Blender lego (nerf.ipynb:1458 output)


def encoded_channels(multires, i_embed=0):
    """Width of gamma(x) for 3 inputs (nerf/embedder.py:82-116)."""
    return 3 if i_embed == -1 else 3 + 6 * multires


def state_dict_shapes(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4,
                      skips=(4,), use_viewdirs=True):
    """Ordered ``name -> shape`` exactly as ``NeRF(...).state_dict()`` lists them.

    nn.Linear stores weight as [out, in] (nerf/nerf.py:32-55).
    """
    shapes = OrderedDict()
    for i in range(D):
        if i == 0:
            fan_in = input_ch
        elif (i - 1) in skips:
            fan_in = W + input_ch
        else:
            fan_in = W
        shapes[f"pts_linears.{i}.weight"] = (W, fan_in)
        shapes[f"pts_linears.{i}.bias"] = (W,)
    shapes["views_linears.0.weight"] = (W // 2, input_ch_views + W)
    shapes["views_linears.0.bias"] = (W // 2,)
    if use_viewdirs:
        shapes["feature_linear.weight"] = (W, W)
        shapes["feature_linear.bias"] = (W,)
        shapes["alpha_linear.weight"] = (1, W)
        shapes["alpha_linear.bias"] = (1,)
        shapes["rgb_linear.weight"] = (3, W // 2)
        shapes["rgb_linear.bias"] = (3,)
    else:
        shapes["output_linear.weight"] = (output_ch, W)
        shapes["output_linear.bias"] = (output_ch,)
    return shapes


def _pe_column_gain(input_ch, decay):
    """Per-input-column gain 2^(-decay*k) for the sin/cos block of frequency 2^k.

    Column order is [x y z | sin f0 (3) cos f0 (3) | sin f1 ...] (embedder.py:28-65).
    """
    g = np.ones(input_ch, dtype=np.float64)
    for c in range(3, input_ch):
        k = (c - 3) // 6
        g[c] = 2.0 ** (-decay * k)
    return g


def synthetic_state_dict(seed, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4,
                         skips=(4,), use_viewdirs=True, sigma_scale=12.0, sigma_bias=None,
                         occupied=0.25, pe_decay=0.9, cube=1.0, wall=40.0, rgb_scale=2.0):
    """Seeded fp32 weights keyed like the reference ``state_dict``.

    Trunk layers use a variance-preserving uniform init (He) so the random field
    keeps O(1) amplitude through 8 ReLU layers; columns that read the positional
    encoding of frequency 2^k are damped by 2^(-pe_decay*k) (trained NeRFs are
    smooth at this scale; an undamped random net crosses sigma=0 every few
    millimetres, which makes the fine pass ill-conditioned in *any* fp32
    implementation - SURVEY.md section 7).

    "Empty space outside the unit cube": when a skip connection exists, six
    trunk channels of the layer that reads ``[input_pts, h]`` (nerf/nerf.py:79-80)
    are rewired to ``relu(+-x_c - cube)`` and carried by identity through the
    remaining trunk layers; ``alpha_linear`` (or the sigma row of
    ``output_linear``) subtracts ``wall`` times their sum, so sigma is strongly
    negative wherever max|x_c| > cube. This is data, not code: the product path
    sees an ordinary state dict.

    ``sigma_bias=None`` calibrates the density bias so that a fraction
    ``occupied`` of the unit cube has sigma > 0: the field is evaluated in fp64 on
    4096 seeded points and the bias is rounded to 1e-2, which keeps the result
    independent of the host's BLAS / libm rounding (``state_dict_digest`` is
    stored in the fixtures and checked on load in case it ever is not).
    """
    rs = np.random.RandomState(seed)
    shapes = state_dict_shapes(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs)
    sd = OrderedDict()
    for name, shape in shapes.items():
        if name.endswith(".weight"):
            fan_in = shape[1]
            bound = np.sqrt(6.0 / fan_in)
            sd[name] = rs.uniform(-bound, bound, size=shape)
        else:
            sd[name] = rs.uniform(-0.1, 0.1, size=shape)

    gain = _pe_column_gain(input_ch, pe_decay)
    sd["pts_linears.0.weight"] = sd["pts_linears.0.weight"] * gain[None, :] * np.sqrt(input_ch / 9.0)
    for s in skips:
        if s + 1 < D:
            name = f"pts_linears.{s + 1}.weight"
            sd[name][:, :input_ch] *= gain[None, :]

    sig_name = "alpha_linear" if use_viewdirs else "output_linear"
    sig_row = 0 if use_viewdirs else 3
    head_w = sd[f"{sig_name}.weight"]
    head_b = sd[f"{sig_name}.bias"]
    head_w[sig_row] *= sigma_scale / np.sqrt(2.0)
    head_b[sig_row] = 0.0 if sigma_bias is None else sigma_bias
    if use_viewdirs:
        sd["rgb_linear.weight"] *= rgb_scale
    else:
        head_w[:3] *= rgb_scale

    usable_skips = [s for s in skips if s + 1 < D]
    if usable_skips and input_ch >= 3:
        L = usable_skips[-1] + 1          # the layer whose input is [input_pts, h]
        ch = list(range(W - 6, W))        # six dedicated trunk channels
        wl = sd[f"pts_linears.{L}.weight"]
        bl = sd[f"pts_linears.{L}.bias"]
        for n, c in enumerate(ch):
            axis, sign = n % 3, (1.0 if n < 3 else -1.0)
            wl[c, :] = 0.0
            wl[c, axis] = sign            # column `axis` of input_pts is the raw coordinate
            bl[c] = -cube
        for j in range(L + 1, D):
            wj = sd[f"pts_linears.{j}.weight"]
            bj = sd[f"pts_linears.{j}.bias"]
            for c in ch:
                wj[c, :] = 0.0
                wj[c, c] = 1.0
                bj[c] = 0.0
                wj[: W - 6, c] = 0.0      # keep the wall channels out of the random field
        head_w[sig_row, ch] = -wall
        if use_viewdirs:
            sd["feature_linear.weight"][:, ch] = 0.0
        else:
            others = [r for r in range(head_w.shape[0]) if r != sig_row]
            head_w[np.ix_(others, ch)] = 0.0
    if sigma_bias is None:
        pts = np.random.RandomState(seed + 7919).uniform(-cube, cube, size=(4096, 3))
        sig = _sigma_fp64(sd, pts, D, W, input_ch, skips, use_viewdirs)
        head_b[sig_row] = -float(np.round(np.quantile(sig, 1.0 - occupied), 2))
    return OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items())


def _sigma_fp64(sd, pts, D, W, input_ch, skips, use_viewdirs):
    """fp64 trunk evaluation used only to calibrate the density bias."""
    L = (input_ch - 3) // 6
    enc = [pts] + [fn(pts * 2.0 ** k) for k in range(L) for fn in (np.sin, np.cos)]
    x = np.concatenate(enc, -1)[:, :input_ch]
    h = x
    for i in range(D):
        h = np.maximum(h @ sd[f"pts_linears.{i}.weight"].T + sd[f"pts_linears.{i}.bias"], 0.0)
        if i in skips:
            h = np.concatenate([x, h], -1)
    if use_viewdirs:
        return (h @ sd["alpha_linear.weight"].T + sd["alpha_linear.bias"])[:, 0]
    return (h @ sd["output_linear.weight"].T + sd["output_linear.bias"])[:, 3]


def perturbed_copy(sd, seed, rel=0.05):
    """A multiplicatively perturbed copy: ``w * (1 + rel*U(-1,1))``. Used for the
    'fine' network so that, as in a trained model, it describes the same scene as
    the coarse one while being a different set of numbers."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for k, v in sd.items():
        out[k] = np.ascontiguousarray(v * (1.0 + rel * rs.uniform(-1.0, 1.0, size=v.shape)), dtype=np.float32)
    return out


def synthetic_pair(seed=0, **arch):
    """(coarse, fine) state dicts for the hierarchical renderer."""
    coarse = synthetic_state_dict(seed, **arch)
    return coarse, perturbed_copy(coarse, seed + 1)


def default_init_state_dict(seed, **arch):
    """What a freshly constructed reference ``NeRF`` holds, up to the random stream: ``nn.Linear``'s default
    initialisation draws weight AND bias from U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (kaiming_uniform_ with a = sqrt(5);
    the modules of nerf/nerf.py:32-55 are never re-initialised). The state a training run starts from."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    bound = 0.0
    for k, shape in state_dict_shapes(**arch).items():
        if k.endswith(".weight"):
            bound = 1.0 / np.sqrt(shape[1])
        out[k] = rs.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def state_dict_digest(sd):
    """sha256 over names, shapes and bytes; fixtures store it to detect a host on
    which the seeded generator does not reproduce the build container's weights."""
    import hashlib
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(str(v.shape).encode())
        h.update(np.ascontiguousarray(v, dtype=np.float32).tobytes())
    return h.hexdigest()


# ----------------------------------------------------------------------------------------------
# cameras
# ----------------------------------------------------------------------------------------------

def blender_focal(W, camera_angle_x=LEGO_CAMERA_ANGLE_X):
    """focal = .5*W/tan(.5*camera_angle_x) (nerf/load_blender.py:72-73)."""
    return 0.5 * W / np.tan(0.5 * camera_angle_x)


def intrinsics(H, W, focal):
    """K = [[f,0,W/2],[0,f,H/2],[0,0,1]] (nerf/nerf.ipynb:1088-1094)."""
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float64)


def pose_spherical(theta, phi, radius):
    """Camera-to-world [4,4] on a sphere, restating nerf/load_blender.py:10-34."""
    def trans_t(t):
        return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], dtype=np.float32)

    def rot_phi(p):
        return np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0],
                         [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]], dtype=np.float32)

    def rot_theta(th):
        return np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                         [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], dtype=np.float32)

    c2w = trans_t(radius)
    c2w = rot_phi(phi / 180.0 * np.pi) @ c2w
    c2w = rot_theta(theta / 180.0 * np.pi) @ c2w
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32) @ c2w
    return c2w.astype(np.float32)


def llff_like_pose(offset=(0.15, -0.1, 0.05)):
    """A forward-facing camera slightly off the origin (an origin-centred identity
    pose degenerates NDC directions to (0,0,2))."""
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = np.asarray(offset, dtype=np.float32)
    return c2w


def lego_camera(H=800, W=800, theta=30.0, phi=-30.0, radius=4.0):
    """(K, c2w[3,4], near, far) of the Blender-lego render path
    (nerf/load_blender.py:75, nerf/nerf.ipynb:1043-1044)."""
    f = blender_focal(W)
    return intrinsics(H, W, f), pose_spherical(theta, phi, radius)[:3, :4], 2.0, 6.0


def fern_camera(H=756, W=1008, focal=815.0):
    """(K, c2w[3,4], near, far) for the NDC (LLFF) configuration C4."""
    return intrinsics(H, W, focal), llff_like_pose()[:3, :4], 0.0, 1.0
