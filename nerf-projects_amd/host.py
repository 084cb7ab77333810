"""Host-side mirror of the reference's ray-chunk renderer call surface.

Same names, argument meaning and error behaviour as the notebook cells of
``nerf/nerf.ipynb`` (``batchify`` :224, ``raw2outputs`` :254, ``render_rays`` :359,
``batchify_rays`` :514, ``render`` :558, ``run_network`` :790), ``nerf/nerf.py`` (``NeRF``),
``nerf/embedder.py`` (``get_embedder``) and ``nerf/nerf_helpers.py`` (``sample_pdf``,
``get_rays``, ``ndc_rays``), but every numeric stage runs in hand-written HIP kernels
behind the C ABI of ``include/nerf_mi355x.h``. PyTorch is plumbing only: device memory,
streams, ``torch.distributed``. There is no fallback path: without the built library and a
gfx950 GPU every entry point raises.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import Camera, FrameArgs, NerfArch, RenderArgs, TrainArgs, check

__all__ = [
    "NeRF", "get_embedder", "batchify", "run_network", "raw2outputs", "sample_pdf", "render_rays",
    "batchify_rays", "render", "get_rays", "get_rays_np", "ndc_rays", "make_network_query_fn",
    "get_context", "img2mse", "mse2psnr", "to8b", "generate_rays", "render_path", "calculate_ssim",
    "calculate_lpips", "calculate_metrics", "create_nerf", "load_checkpoint", "Adam", "train_on_batch",
]


# ----------------------------------------------------------------------------------------------
# context
# ----------------------------------------------------------------------------------------------

class Context:
    """One ``nerf_ctx`` per GPU (owns packed weights and the render workspace)."""

    def __init__(self, device_index):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible to PyTorch: the MI355X renderer has no CPU fallback")
        self.lib = lib
        self.device = torch.device("cuda", device_index)
        handle = C.c_void_p()
        check(lib.nerf_ctx_create(device_index, C.byref(handle)))
        self.handle = handle
        self._slots = [None] * _lib.NERF_NUM_SLOTS
        self._chunk_loop = 0       # > 0 while batchify_rays owns the precision guard's events (see _peek_precision)
        default = os.environ.get("NERF_PRECISION")
        if default:
            self.set_precision(default)

    PRECISIONS = {"f32": 0, "f16x2": 1}

    def set_precision(self, name):
        """Arithmetic of the fused MLP kernel: "f32" (fp32 MFMA) or "f16x2" (exact fp16-pair split, 3 MFMAs per term)."""
        if name not in self.PRECISIONS:
            raise ValueError(f"precision {name!r}: expected one of {sorted(self.PRECISIONS)}")
        check(self.lib.nerf_set_precision(self.handle, self.PRECISIONS[name]))

    def set_render_precision(self, name):
        """The same for the rendering calls only: the training step keeps its arithmetic and any fp32 fallback it is in
        (nerf_set_render_precision); ``get_precision`` reports this one."""
        if name not in self.PRECISIONS:
            raise ValueError(f"precision {name!r}: expected one of {sorted(self.PRECISIONS)}")
        check(self.lib.nerf_set_render_precision(self.handle, self.PRECISIONS[name]))

    def precision_status(self, reset=True):
        """Number of (wavefront, layer) events since the last reset in which the fp16-pair kernel's a-priori output
        bound was >= 2^12 too wide (0 for NeRF-like weights; otherwise prefer ``set_precision("f32")``). Synchronises."""
        n = C.c_int64()
        check(self.lib.nerf_precision_status(self.handle, C.byref(n), int(bool(reset))))
        return n.value

    def precision_detail(self, reset=True):
        """[guard counter, then the backward-data kernel's events by overshoot 2^12-13, 2^14-15, ..., >= 2^24]
        (nerf_precision_detail). Synchronises."""
        v = (C.c_int64 * 8)()
        check(self.lib.nerf_precision_detail(self.handle, v, int(bool(reset))))
        return list(v)

    def precision_peek(self):
        """Loose-bound events of COMPLETED work that no call has reported yet (no synchronisation: a pinned mirror of the
        counter follows every render / training call). Marks them reported."""
        n = C.c_int64()
        check(self.lib.nerf_precision_peek(self.handle, C.byref(n)))
        return n.value

    def precision_check(self):
        """The same after waiting for the current stream: events of everything enqueued so far."""
        n = C.c_int64()
        check(self.lib.nerf_precision_check(self.handle, self.stream(), C.byref(n)))
        return n.value

    def get_precision(self):
        code = self.lib.nerf_get_precision(self.handle)
        return {v: k for k, v in self.PRECISIONS.items()}[code]

    def alloc_slot(self, owner):
        for i, o in enumerate(self._slots):
            if o is None or o() is None:
                import weakref
                self._slots[i] = weakref.ref(owner)
                return i
        raise RuntimeError(f"all {_lib.NERF_NUM_SLOTS} network slots of the context are in use")

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def profile_enable(self, on=True):
        check(self.lib.nerf_profile_enable(self.handle, int(bool(on))))

    def profile_read(self, reset=True):
        """(ms, launches, points) of the fused encode+MLP kernel since the last reset."""
        ms, n, pts = C.c_double(), C.c_int64(), C.c_int64()
        check(self.lib.nerf_profile_read(self.handle, C.byref(ms), C.byref(n), C.byref(pts), int(reset)))
        return ms.value, n.value, pts.value

    def profile_read_train(self, reset=True):
        """{kind: (ms, launches, points)} of the training step's kernels since the last reset (while profile_enable is on):
        forward / backward_data / weight_gradients_hidden / weight_gradients_other."""
        ms, n, pts = (C.c_double * 4)(), (C.c_int64 * 4)(), (C.c_int64 * 4)()
        check(self.lib.nerf_profile_read_train(self.handle, ms, n, pts, int(reset)))
        names = ("forward", "backward_data", "weight_gradients_hidden", "weight_gradients_other")
        return {k: (ms[i], n[i], pts[i]) for i, k in enumerate(names)}

    def workspace_bytes(self):
        return int(self.lib.nerf_workspace_bytes(self.handle))


_LOOSE = ("the fp16-pair MLP kernel's output-scale bound was loose in {n} (wavefront, layer) cases, i.e. some activations "
          "kept fewer than 24 bits with these weights")


def _peek_precision(ctx, where):
    """Entry of an asynchronous call: report (never hide) what completed work has counted since the last look. Inside
    ``batchify_rays``' chunk loop the events belong to that call - it checks behind its last chunk and renders the rays
    again in fp32 - so a chunk's entry must not mark an earlier chunk's events as reported."""
    if ctx._chunk_loop or ctx.get_precision() != "f16x2":
        return 0
    n = ctx.precision_peek()
    if n:
        import warnings
        warnings.warn(f"{where}: in earlier calls " + _LOOSE.format(n=n) + "; get_context().set_precision('f32') "
                      "evaluates them as the reference does", RuntimeWarning, stacklevel=3)
    return n


def _warn_if_scale_bound_was_loose(ctx, where):
    """The fp16-pair kernel counts (never hides) the cases in which its a-priori per-point scale bound was >= 2^12 too
    wide for the weights at hand (include/nerf_mi355x.h, nerf_precision_status); surfaced at natural sync points."""
    if ctx.get_precision() != "f16x2":
        return 0
    n = ctx.precision_status(reset=True)
    if n:
        import warnings
        warnings.warn(f"{where}: the fp16-pair MLP kernel's output-scale bound was loose in {n} (wavefront, layer) "
                      "cases since the last check, i.e. some activations kept fewer than 24 bits with these weights; "
                      "use get_context().set_precision('f32') for them", RuntimeWarning, stacklevel=3)
    return n


_contexts = {}


def get_context(device=None):
    if device is None:
        if not torch.cuda.is_available():
            _lib.load()
            raise RuntimeError("no GPU visible to PyTorch: the MI355X renderer has no CPU fallback")
        index = torch.cuda.current_device()
    else:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"device {device} is not a GPU: the MI355X renderer has no CPU fallback")
        index = device.index if device.index is not None else torch.cuda.current_device()
    if index not in _contexts:
        _contexts[index] = Context(index)
    return _contexts[index]


def _dev(t, ctx, dtype=torch.float32):
    """fp32, contiguous, on the context's GPU (the reference does ``.float().to(model_device)``)."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t))
    return t.detach().to(device=ctx.device, dtype=dtype).contiguous()


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


# ----------------------------------------------------------------------------------------------
# NeRF module (nerf/nerf.py:8-146)
# ----------------------------------------------------------------------------------------------

class NeRF:
    """Forward-only stand-in for the reference ``NeRF`` ``nn.Module`` (nerf/nerf.py:8-111).

    Same constructor; ``load_state_dict`` accepts the reference's state dict (torch tensors
    or numpy arrays, keys ``pts_linears.i.weight`` ...) and repacks it for the MFMA kernel.
    ``__call__(x)`` is ``forward`` on already-encoded rows ``[B, input_ch + input_ch_views]``.
    """

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 device=None):
        self.D, self.W = int(D), int(W)
        self.input_ch, self.input_ch_views = int(input_ch), int(input_ch_views)
        self.output_ch = int(output_ch)
        self.skips = list(skips)
        self.use_viewdirs = bool(use_viewdirs)
        self.ctx = get_context(device)
        self.slot = self.ctx.alloc_slot(self)
        self._sd = None

    # -- state dict ------------------------------------------------------------------------
    def state_dict_keys(self):
        keys = []
        for i in range(self.D):
            keys += [f"pts_linears.{i}.weight", f"pts_linears.{i}.bias"]
        keys += ["views_linears.0.weight", "views_linears.0.bias"]
        if self.use_viewdirs:
            keys += ["feature_linear.weight", "feature_linear.bias", "alpha_linear.weight", "alpha_linear.bias",
                     "rgb_linear.weight", "rgb_linear.bias"]
        else:
            keys += ["output_linear.weight", "output_linear.bias"]
        return keys

    def _expected_shape(self, key):
        W, D = self.W, self.D
        name, kind = key.rsplit(".", 1)
        if name.startswith("pts_linears"):
            i = int(name.split(".")[1])
            fan_in = self.input_ch if i == 0 else (W + self.input_ch if (i - 1) in self.skips else W)
            shape = (W, fan_in)
        elif name == "views_linears.0":
            shape = (W // 2, self.input_ch_views + W)
        elif name == "feature_linear":
            shape = (W, W)
        elif name == "alpha_linear":
            shape = (1, W)
        elif name == "rgb_linear":
            shape = (3, W // 2)
        else:
            shape = (self.output_ch, W)
        return shape if kind == "weight" else shape[:1]

    def load_state_dict(self, state_dict, strict=True):
        keys = self.state_dict_keys()
        missing = [k for k in keys if k not in state_dict]
        unexpected = [k for k in state_dict if k not in keys]
        if missing or (strict and unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for NeRF: missing {missing}, unexpected {unexpected}")
        arrays = []
        for k in keys:
            v = state_dict[k]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            v = np.ascontiguousarray(v, dtype=np.float32)
            if tuple(v.shape) != self._expected_shape(k):
                raise RuntimeError(f"size mismatch for {k}: got {tuple(v.shape)}, expected {self._expected_shape(k)}")
            arrays.append(v)
        arch = NerfArch()
        arch.D, arch.W = self.D, self.W
        arch.input_ch, arch.input_ch_views, arch.output_ch = self.input_ch, self.input_ch_views, self.output_ch
        if len(self.skips) > _lib.NERF_MAX_SKIPS:
            raise RuntimeError(f"at most {_lib.NERF_MAX_SKIPS} skip connections are supported")
        arch.n_skips = len(self.skips)
        for i, s in enumerate(self.skips):
            arch.skips[i] = int(s)
        arch.use_viewdirs = int(self.use_viewdirs)
        ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        check(self.ctx.lib.nerf_load_weights(self.ctx.handle, self.slot, C.byref(arch), ptrs, len(arrays)))
        self._sd = dict(zip(keys, arrays))
        return self

    def _read_flat(self, fn):
        keys = self.state_dict_keys()
        arrays = [np.empty(self._expected_shape(k), dtype=np.float32) for k in keys]
        ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        check(fn(self.ctx.handle, self.slot, ptrs, len(arrays)))
        return dict(zip(keys, arrays))

    def state_dict(self):
        """The current weights (read back from the device, so training updates are visible)."""
        if self._sd is None:
            raise RuntimeError("no weights loaded")
        self._sd = self._read_flat(self.ctx.lib.nerf_get_weights)
        return {k: torch.from_numpy(v.copy()) for k, v in self._sd.items()}

    def grad_dict(self):
        """Gradients of the last training step, keyed like the state dict (``param.grad``)."""
        return {k: torch.from_numpy(v) for k, v in self._read_flat(self.ctx.lib.nerf_get_gradients).items()}

    def adam_state(self):
        """(exp_avg, exp_avg_sq) of this model's Adam state, each keyed like the state dict."""
        keys = self.state_dict_keys()
        m = [np.empty(self._expected_shape(k), dtype=np.float32) for k in keys]
        v = [np.empty(self._expected_shape(k), dtype=np.float32) for k in keys]
        pm = (C.c_void_p * len(m))(*[a.ctypes.data for a in m])
        pv = (C.c_void_p * len(v))(*[a.ctypes.data for a in v])
        check(self.ctx.lib.nerf_get_adam_state(self.ctx.handle, self.slot, pm, pv, len(m)))
        return dict(zip(keys, m)), dict(zip(keys, v))

    def load_adam_state(self, exp_avg, exp_avg_sq):
        """Inverse of :meth:`adam_state`; the sequences are in state-dict order."""
        keys = self.state_dict_keys()
        m = [np.ascontiguousarray(np.asarray(a, dtype=np.float32)).reshape(self._expected_shape(k)) for k, a in zip(keys, exp_avg)]
        v = [np.ascontiguousarray(np.asarray(a, dtype=np.float32)).reshape(self._expected_shape(k)) for k, a in zip(keys, exp_avg_sq)]
        if len(m) != len(keys) or len(v) != len(keys):
            raise ValueError(f"expected {len(keys)} tensors per moment")
        pm = (C.c_void_p * len(m))(*[a.ctypes.data for a in m])
        pv = (C.c_void_p * len(v))(*[a.ctypes.data for a in v])
        check(self.ctx.lib.nerf_set_adam_state(self.ctx.handle, self.slot, pm, pv, len(m)))

    def load_weights_from_keras(self, weights):
        """nerf/nerf.py:113-146: weights of the original TensorFlow NeRF as a flat list
        ``[kernel, bias] * D, feature, views, rgb, alpha`` with Keras ``[in, out]`` kernels."""
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"
        sd = {}
        for i in range(self.D):
            sd[f"pts_linears.{i}.weight"] = np.transpose(np.asarray(weights[2 * i]))
            sd[f"pts_linears.{i}.bias"] = np.transpose(np.asarray(weights[2 * i + 1]))
        for name, idx in (("feature_linear", 2 * self.D), ("views_linears.0", 2 * self.D + 2),
                          ("rgb_linear", 2 * self.D + 4), ("alpha_linear", 2 * self.D + 6)):
            sd[name + ".weight"] = np.transpose(np.asarray(weights[idx]))
            sd[name + ".bias"] = np.transpose(np.asarray(weights[idx + 1]))
        return self.load_state_dict(sd)

    def parameters(self):
        """Yields device placeholders so ``next(fn.parameters()).device`` works (nerf.ipynb:598, :848)."""
        yield torch.empty(0, device=self.ctx.device)

    @property
    def device(self):
        return self.ctx.device

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    @property
    def out_channels(self):
        return 4 if self.use_viewdirs else self.output_ch

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x):
        if self._sd is None:
            raise RuntimeError("NeRF.forward before load_state_dict")
        width = self.input_ch + self.input_ch_views
        if x.shape[-1] != width:
            raise RuntimeError(f"split_with_sizes expects the last dimension to be {width}, got {x.shape[-1]}")
        lead = list(x.shape[:-1])
        xf = _dev(x, self.ctx).reshape(-1, width)
        out = torch.empty((xf.shape[0], self.out_channels), device=self.ctx.device, dtype=torch.float32)
        check(self.ctx.lib.nerf_mlp_forward(self.ctx.handle, self.slot, _ptr(xf), xf.shape[0], _ptr(out),
                                            self.ctx.stream()))
        return out.reshape(lead + [self.out_channels])

    __call__ = forward


# ----------------------------------------------------------------------------------------------
# positional encoding (nerf/embedder.py:82-116)
# ----------------------------------------------------------------------------------------------

class _Embed:
    """``embed_fn`` returned by :func:`get_embedder`; callable like the reference's lambda."""

    def __init__(self, multires):
        self.multires = int(multires)
        self.out_dim = 3 + 6 * self.multires

    def __call__(self, x):
        ctx = get_context(x.device if torch.is_tensor(x) and x.is_cuda else None)
        if x.shape[-1] != 3:
            raise RuntimeError(f"embedder expects 3 input dims, got {x.shape[-1]}")
        lead = list(x.shape[:-1])
        xf = _dev(x, ctx).reshape(-1, 3)
        out = torch.empty((xf.shape[0], self.out_dim), device=ctx.device, dtype=torch.float32)
        check(ctx.lib.nerf_embed(ctx.handle, _ptr(xf), xf.shape[0], self.multires, _ptr(out), ctx.stream()))
        return out.reshape(lead + [self.out_dim])


def get_embedder(multires, i=0):
    """``(embed_fn, out_dim)``; ``i == -1`` is the identity with out_dim 3 (embedder.py:89-92)."""
    if i == -1:
        e = _Embed(0)
        return e, 3
    e = _Embed(multires)
    return e, e.out_dim


# ----------------------------------------------------------------------------------------------
# batchify / run_network (nerf.ipynb:224-244, 790-855)
# ----------------------------------------------------------------------------------------------

def batchify(fn, chunk):
    """nerf.ipynb:224-244."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """nerf.ipynb:790-855. With this package's embedders and ``NeRF`` the encoding, the
    per-sample viewdir broadcast and the MLP run as ONE fused kernel (``netchunk`` only bounded
    memory in the reference and results do not depend on it); any other ``fn`` /
    embedder combination is composed stage by stage exactly as the reference does."""
    fused = (isinstance(fn, NeRF) and isinstance(embed_fn, _Embed)
             and embed_fn.out_dim == fn.input_ch and inputs.shape[-1] == 3
             and (viewdirs is None or (isinstance(embeddirs_fn, _Embed)
                                       and embeddirs_fn.out_dim == fn.input_ch_views)))
    if fused and (viewdirs is not None) == fn.use_viewdirs:
        ctx = fn.ctx
        pts = _dev(inputs, ctx)
        lead = list(pts.shape[:-1])
        if viewdirs is not None:
            n_rays = viewdirs.shape[0]
            if pts.dim() < 2 or pts.shape[0] != n_rays:
                raise RuntimeError(f"viewdirs [{n_rays},3] cannot be expanded to inputs {tuple(pts.shape)}")
            vd = _dev(viewdirs, ctx).reshape(-1, 3)
        else:
            n_rays = pts.shape[0] if pts.dim() > 1 else 1
            vd = None
        flat = pts.reshape(-1, 3)
        n_samples = flat.shape[0] // max(n_rays, 1)
        out = torch.empty((flat.shape[0], fn.out_channels), device=ctx.device, dtype=torch.float32)
        check(ctx.lib.nerf_run_network(ctx.handle, fn.slot, _ptr(flat), _ptr(vd), n_rays, max(n_samples, 1),
                                       _ptr(out), ctx.stream()))
        return out.reshape(lead + [fn.out_channels])
    # generic composition (nerf.ipynb:827-855)
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape)
        input_dirs_flat = torch.reshape(input_dirs, [-1, input_dirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


class NetworkQuery:
    """The ``network_query_fn`` that ``create_nerf`` builds as a lambda (nerf.ipynb:899-902),
    as an introspectable callable so that ``render_rays`` can take the fully fused path."""

    def __init__(self, embed_fn, embeddirs_fn, netchunk=1024 * 64):
        self.embed_fn, self.embeddirs_fn, self.netchunk = embed_fn, embeddirs_fn, netchunk

    def __call__(self, inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=self.embed_fn,
                           embeddirs_fn=self.embeddirs_fn, netchunk=self.netchunk)

    def matches(self, net, has_viewdirs):
        return (isinstance(net, NeRF) and isinstance(self.embed_fn, _Embed)
                and self.embed_fn.out_dim == net.input_ch and net.use_viewdirs == has_viewdirs
                and (not has_viewdirs or (isinstance(self.embeddirs_fn, _Embed)
                                          and self.embeddirs_fn.out_dim == net.input_ch_views)))


def make_network_query_fn(embed_fn, embeddirs_fn, netchunk=1024 * 64):
    return NetworkQuery(embed_fn, embeddirs_fn, netchunk)


# ----------------------------------------------------------------------------------------------
# raw2outputs (nerf.ipynb:254-349)
# ----------------------------------------------------------------------------------------------

def _noise(shape, raw_noise_std, pytest, ctx):
    """Sigma noise exactly as the reference draws it (nerf.ipynb:312-325), or None."""
    try:
        noise_std = float(raw_noise_std)      # YAML may hand over the string '1e0'
    except (TypeError, ValueError):
        noise_std = 0.0
    if noise_std <= 0.0:
        return None
    if pytest:
        np.random.seed(0)
        return _dev(np.random.rand(*shape) * noise_std, ctx)
    noise = torch.randn(shape, device=ctx.device, dtype=torch.float32)
    return noise if noise_std == 1.0 else noise * noise_std      # (every YAML of the reference has raw_noise_std 1e0: x * 1 = x)


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """Returns ``(rgb_map, disp_map, acc_map, weights, depth_map)`` (nerf.ipynb:254-349)."""
    ctx = get_context(raw.device if torch.is_tensor(raw) and raw.is_cuda else None)
    raw, z_vals, rays_d = _dev(raw, ctx), _dev(z_vals, ctx), _dev(rays_d, ctx)
    N, S, Cc = raw.shape
    if Cc < 4 or tuple(z_vals.shape) != (N, S) or tuple(rays_d.shape) != (N, 3):
        raise RuntimeError(f"raw2outputs: inconsistent shapes raw {tuple(raw.shape)}, z_vals "
                           f"{tuple(z_vals.shape)}, rays_d {tuple(rays_d.shape)}")
    noise = _noise((N, S), raw_noise_std, pytest, ctx)
    o = dict(device=ctx.device, dtype=torch.float32)
    rgb, disp, acc = torch.empty((N, 3), **o), torch.empty((N,), **o), torch.empty((N,), **o)
    weights, depth = torch.empty((N, S), **o), torch.empty((N,), **o)
    check(ctx.lib.nerf_raw2outputs(ctx.handle, _ptr(raw), Cc, _ptr(z_vals), _ptr(rays_d), _ptr(noise),
                                   int(bool(white_bkgd)), N, S, _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(weights),
                                   _ptr(depth), ctx.stream()))
    return rgb, disp, acc, weights, depth


# ----------------------------------------------------------------------------------------------
# sample_pdf (nerf/nerf_helpers.py:372-439)
# ----------------------------------------------------------------------------------------------

def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    ctx = get_context(bins.device if torch.is_tensor(bins) and bins.is_cuda else None)
    bins, weights = _dev(bins, ctx), _dev(weights, ctx)
    lead = list(bins.shape[:-1])
    M = bins.shape[-1]
    if weights.shape[-1] != M - 1 or list(weights.shape[:-1]) != lead:
        raise RuntimeError(f"sample_pdf: bins {tuple(bins.shape)} need weights [..., {M - 1}], got "
                           f"{tuple(weights.shape)}")
    b2, w2 = bins.reshape(-1, M), weights.reshape(-1, M - 1)
    N = b2.shape[0]
    u = None
    if pytest:                                   # nerf_helpers.py:410-418
        np.random.seed(0)
        if det:
            u = _dev(np.broadcast_to(np.linspace(0.0, 1.0, N_samples), (N, N_samples)).copy(), ctx)
        else:
            u = _dev(np.random.rand(*(lead + [N_samples])).reshape(N, N_samples), ctx)
    elif not det:
        u = torch.rand((N, N_samples), device=ctx.device, dtype=torch.float32)
    out = torch.empty((N, N_samples), device=ctx.device, dtype=torch.float32)
    check(ctx.lib.nerf_sample_pdf(ctx.handle, _ptr(b2), _ptr(w2), _ptr(u), N, M, int(N_samples), _ptr(out),
                                  ctx.stream()))
    return out.reshape(lead + [N_samples])


# ----------------------------------------------------------------------------------------------
# render_rays (nerf.ipynb:359-492)
# ----------------------------------------------------------------------------------------------

def _render_rays_fused(ctx, ray_batch, net_c, net_f, N_samples, N_importance, retraw, lindisp, perturb,
                       white_bkgd, raw_noise_std, pytest, extras=None, z_vals_fine_in=None):
    N, stride = ray_batch.shape
    Sc, Si = int(N_samples), int(N_importance)
    o = dict(device=ctx.device, dtype=torch.float32)
    a = RenderArgs()
    a.rays, a.n_rays, a.ray_stride = ray_batch.data_ptr(), N, stride
    a.N_samples, a.N_importance = Sc, Si
    a.slot_coarse = net_c.slot
    a.slot_fine = net_f.slot if net_f is not None else -1
    a.lindisp, a.white_bkgd = int(bool(lindisp)), int(bool(white_bkgd))
    keep = []
    if perturb > 0.:
        a.perturb = 1
        if pytest:                               # nerf.ipynb:439-442: np.random.seed(0) before each draw
            np.random.seed(0)
            t_rand = _dev(np.random.rand(N, Sc), ctx)
        else:
            t_rand = torch.rand((N, Sc), **o)
        keep.append(t_rand)
        a.t_rand = t_rand.data_ptr()
    n0 = _noise((N, Sc), raw_noise_std, pytest, ctx)
    if n0 is not None:
        keep.append(n0)
        a.noise0 = n0.data_ptr()
    if Si > 0:
        if perturb > 0.:
            if pytest:
                np.random.seed(0)
                u = _dev(np.random.rand(N, Si), ctx)
            else:
                u = torch.rand((N, Si), **o)
            keep.append(u)
            a.u_rand = u.data_ptr()
        n1 = _noise((N, Sc + Si), raw_noise_std, pytest, ctx)
        if n1 is not None:
            keep.append(n1)
            a.noise = n1.data_ptr()
    ret = {"rgb_map": torch.empty((N, 3), **o), "disp_map": torch.empty((N,), **o),
           "acc_map": torch.empty((N,), **o)}
    a.rgb_map, a.disp_map, a.acc_map = (ret[k].data_ptr() for k in ("rgb_map", "disp_map", "acc_map"))
    if retraw:
        last = net_f if (Si > 0 and net_f is not None) else net_c
        ret["raw"] = torch.empty((N, Sc + Si, last.out_channels), **o)
        a.raw = ret["raw"].data_ptr()
    if Si > 0:
        ret["rgb0"], ret["disp0"], ret["acc0"] = torch.empty((N, 3), **o), torch.empty((N,), **o), torch.empty((N,), **o)
        ret["z_std"] = torch.empty((N,), **o)
        a.rgb0, a.disp0, a.acc0, a.z_std = (ret[k].data_ptr() for k in ("rgb0", "disp0", "acc0", "z_std"))
    if extras is not None:
        extras["z_coarse"] = torch.empty((N, Sc), **o)
        extras["weights_coarse"] = torch.empty((N, Sc), **o)
        a.z_vals_coarse, a.weights_coarse = extras["z_coarse"].data_ptr(), extras["weights_coarse"].data_ptr()
        if Si > 0:
            extras["z_samples"] = torch.empty((N, Si), **o)
            extras["z_fine"] = torch.empty((N, Sc + Si), **o)
            extras["weights_fine"] = torch.empty((N, Sc + Si), **o)
            a.z_samples, a.z_vals_fine = extras["z_samples"].data_ptr(), extras["z_fine"].data_ptr()
            a.weights_fine = extras["weights_fine"].data_ptr()
    if z_vals_fine_in is not None:
        zin = _dev(z_vals_fine_in, ctx)
        keep.append(zin)
        a.z_vals_fine_in = zin.data_ptr()
    a.stream = ctx.stream().value
    check(ctx.lib.nerf_render_rays(ctx.handle, C.byref(a)))
    # `keep` tensors are consumed by work already enqueued on the current stream; PyTorch's
    # caching allocator only reuses their memory for later work on that same stream.
    return ret


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False,
                pytest=False, _extras=None, _z_vals_fine=None):
    """Volume-render one chunk of rays; same arguments and return dict as nerf.ipynb:359-492.

    When ``network_query_fn`` is this package's :class:`NetworkQuery` over this package's
    embedders and ``NeRF`` models, the whole chunk (sampling, encoding, both MLP passes,
    compositing, resampling) runs inside one C call with no host synchronisation. Any other
    ``network_query_fn`` is honoured as an opaque callable and the stages are composed
    around it as in the reference, each stage still a HIP kernel.
    """
    if not isinstance(network_fn, NeRF):
        raise TypeError("render_rays needs this package's NeRF for network_fn (no PyTorch fallback exists)")
    ctx = network_fn.ctx
    _peek_precision(ctx, "render_rays")
    ray_batch = _dev(ray_batch, ctx)
    if ray_batch.dim() != 2 or ray_batch.shape[-1] not in (8, 11):
        raise RuntimeError(f"ray_batch must be [N, 8|11], got {tuple(ray_batch.shape)}")
    has_dirs = ray_batch.shape[-1] > 8
    perturb = float(perturb)
    fused = (isinstance(network_query_fn, NetworkQuery) and network_query_fn.matches(network_fn, has_dirs)
             and (network_fine is None or network_query_fn.matches(network_fine, has_dirs)))
    if fused:
        return _render_rays_fused(ctx, ray_batch, network_fn, network_fine, N_samples, N_importance, retraw,
                                  lindisp, perturb, white_bkgd, raw_noise_std, pytest, _extras, _z_vals_fine)

    # ---- staged route: an opaque network_query_fn is honoured; every other stage is a kernel ----------
    return _render_rays_staged(ctx, ray_batch, network_fn, network_query_fn, int(N_samples), int(N_importance),
                               network_fine, retraw, lindisp, perturb, white_bkgd, raw_noise_std, pytest, _extras)


def _uniforms(shape, pytest, ctx):
    """U[0,1) draws in the reference's order; under ``pytest`` its ``np.random.seed(0)`` sequence."""
    if pytest:
        np.random.seed(0)
        return _dev(np.random.rand(*shape), ctx)
    return torch.rand(shape, device=ctx.device, dtype=torch.float32)


def _stage_depths(ctx, ray_batch, n_samples, lindisp, t_rand):
    z = torch.empty((ray_batch.shape[0], n_samples), device=ctx.device, dtype=torch.float32)
    check(ctx.lib.nerf_stratified_z(ctx.handle, _ptr(ray_batch), ray_batch.shape[1], ray_batch.shape[0], n_samples,
                                    int(bool(lindisp)), _ptr(t_rand), _ptr(z), ctx.stream()))
    return z


def _stage_resample(ctx, z_vals, weights, n_importance, u):
    N, S = z_vals.shape
    o = dict(device=ctx.device, dtype=torch.float32)
    z_samples, z_merged, z_std = torch.empty((N, n_importance), **o), torch.empty((N, S + n_importance), **o), \
        torch.empty((N,), **o)
    check(ctx.lib.nerf_resample(ctx.handle, _ptr(z_vals), _ptr(weights.contiguous()), _ptr(u), N, S, n_importance,
                                _ptr(z_samples), _ptr(z_merged), _ptr(z_std), ctx.stream()))
    return z_samples, z_merged, z_std


def _render_rays_staged(ctx, ray_batch, network_fn, query, Sc, Si, network_fine, retraw, lindisp, perturb, white_bkgd,
                        raw_noise_std, pytest, extras):
    """render_rays with the network evaluation delegated to ``query(pts, viewdirs, net)``
    (nerf.ipynb:447-475): depth sampling, compositing, resampling and merging are the package's kernels;
    only ``pts = o + d*z`` is plain tensor arithmetic on the device."""
    origins, dirs = ray_batch[:, None, 0:3], ray_batch[:, None, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    jitter = _uniforms((ray_batch.shape[0], Sc), pytest, ctx) if perturb > 0. else None

    def shade(z_vals, net):
        raw = query(origins + dirs * z_vals[..., :, None], viewdirs, net)
        return raw, raw2outputs(raw, z_vals, ray_batch[:, 3:6], raw_noise_std, white_bkgd, pytest=pytest)

    z_vals = _stage_depths(ctx, ray_batch, Sc, lindisp, jitter)
    raw, (rgb, disp, acc, weights, _) = shade(z_vals, network_fn)
    if extras is not None:
        extras.update(z_coarse=z_vals, weights_coarse=weights)
    ret = {}
    if Si > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc)
        u = _uniforms((ray_batch.shape[0], Si), pytest, ctx) if perturb > 0. else None      # det = (perturb == 0)
        z_samples, z_vals, ret['z_std'] = _stage_resample(ctx, z_vals, weights, Si, u)
        raw, (rgb, disp, acc, weights, _) = shade(z_vals, network_fn if network_fine is None else network_fine)
        if extras is not None:
            extras.update(z_samples=z_samples, z_fine=z_vals, weights_fine=weights)
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc)
    if retraw:
        ret['raw'] = raw
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """nerf.ipynb:514-548."""
    def run():
        all_ret = {}
        for i in range(0, rays_flat.shape[0], chunk):
            ret = render_rays(rays_flat[i:i + chunk], **kwargs)
            for k in ret:
                all_ret.setdefault(k, []).append(ret[k])
        return {k: torch.cat(all_ret[k], dim=0) for k in all_ret}

    net = kwargs.get('network_fn')
    ctx = getattr(net, 'ctx', None)
    if ctx is None or ctx.get_precision() != "f16x2":
        return run()
    # Precision guard: the reference evaluates the network in fp32 (nerf.ipynb:76). If the fp16-pair kernel counted a loose
    # scale bound on these rays, they are rendered again by the fp32 kernel - one look at the counter behind the last chunk.
    # What earlier calls counted is reported here, once; from then on every event up to the check belongs to these rays,
    # whichever chunk it came from and however far the GPU has got when the next chunk is entered.
    _peek_precision(ctx, "batchify_rays")
    ctx._chunk_loop += 1
    try:
        out = run()
        n = ctx.precision_check() if out else 0
        if n:
            import warnings
            warnings.warn("batchify_rays: " + _LOOSE.format(n=n) + "; these rays were rendered again with the fp32 kernel",
                          RuntimeWarning, stacklevel=2)
            ctx.set_render_precision("f32")      # (rendering only: a training loop around this call keeps its arithmetic)
            try:
                out = run()
            finally:
                ctx.set_render_precision("f16x2")
    finally:
        ctx._chunk_loop -= 1
    return out


# ----------------------------------------------------------------------------------------------
# ray generation and render() (nerf_helpers.py:222-369, nerf.ipynb:558-640)
# ----------------------------------------------------------------------------------------------

def get_rays(H, W, K, c2w):
    """nerf_helpers.py:222-296 (torch ops on ``c2w``'s device; ray generation is a caller of the
    hot path)."""
    device, dtype = c2w.device, c2w.dtype
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device, dtype=dtype),
                          torch.linspace(0, H - 1, H, device=device, dtype=dtype), indexing="ij")
    i, j = i.t(), j.t()
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], dim=-1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], dim=-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H, W, K, c2w):
    """nerf_helpers.py:301-308."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """nerf_helpers.py:311-369."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], dim=-1), torch.stack([d0, d1, d2], dim=-1)


def pack_rays(H, W, K, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
              c2w_staticcam=None, device=None):
    """The ``[N, 8|11]`` ray record ``render()`` builds (nerf.ipynb:596-629) and the grid shape."""
    if c2w is not None:
        if not torch.is_tensor(c2w):
            c2w = torch.as_tensor(np.asarray(c2w), dtype=torch.float32)
        if device is not None:
            c2w = c2w.to(device)
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
        if not torch.is_tensor(rays_o):
            rays_o, rays_d = torch.as_tensor(np.asarray(rays_o)), torch.as_tensor(np.asarray(rays_d))
        if device is not None:
            rays_o, rays_d = rays_o.to(device), rays_d.to(device)
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            if not torch.is_tensor(c2w_staticcam):
                c2w_staticcam = torch.as_tensor(np.asarray(c2w_staticcam), dtype=torch.float32)
            rays_o, rays_d = get_rays(H, W, K, c2w_staticcam.to(rays_d.device))
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    sh = rays_d.shape
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    # near * ones_like(...), far * ones_like(...), two cats (nerf.ipynb:622-629): the same values in three launches less
    near_c = torch.full_like(rays_d[..., :1], float(near))
    far_c = torch.full_like(rays_d[..., :1], float(far))
    packed = torch.cat([rays_o, rays_d, near_c, far_c] + ([viewdirs] if use_viewdirs else []), -1)
    return packed, sh


def _mat34(m):
    m = m.detach().cpu().numpy() if torch.is_tensor(m) else np.asarray(m)
    m = np.asarray(m, dtype=np.float32)
    if m.shape[0] < 3 or m.shape[1] < 4:
        raise RuntimeError(f"camera-to-world matrix must be at least [3,4], got {m.shape}")
    return np.ascontiguousarray(m[:3, :4]).reshape(-1)


def _camera(H, W, K, c2w, ndc, near, far, use_viewdirs, c2w_staticcam):
    cam = Camera()
    cam.H, cam.W = int(H), int(W)
    # torch computes (i - K[0][2]) / K[0][0] in fp32 with the Python/numpy scalars cast to fp32
    cam.fx, cam.fy, cam.cx, cam.cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    cam.c2w = (C.c_float * 12)(*_mat34(c2w))
    if c2w_staticcam is not None and use_viewdirs:
        cam.c2w_static = (C.c_float * 12)(*_mat34(c2w_staticcam))
        cam.has_static = 1
    cam.ndc, cam.ndc_focal = int(bool(ndc)), float(K[0][0])
    cam.near, cam.far, cam.use_viewdirs = float(near), float(far), int(bool(use_viewdirs))
    return cam


def generate_rays(H, W, K, c2w, ndc=True, near=0., far=1., use_viewdirs=False, c2w_staticcam=None,
                  first_pixel=0, n_pixels=None, device=None):
    """The ``[n_pixels, 8|11]`` ray record of ``render()`` (nerf.ipynb:596-629) generated on the GPU by
    one kernel (``get_rays`` + viewdir normalisation + optional ``ndc_rays`` + near/far columns) for the
    flat pixel range ``[first_pixel, first_pixel + n_pixels)``: no host ray generation, no 28 MB H2D copy,
    and a rank of a sharded render only ever materialises its own shard."""
    ctx = get_context(device)
    cam = _camera(H, W, K, c2w, ndc, near, far, use_viewdirs, c2w_staticcam)
    n = cam.H * cam.W - first_pixel if n_pixels is None else int(n_pixels)
    out = torch.empty((n, 11 if use_viewdirs else 8), device=ctx.device, dtype=torch.float32)
    check(ctx.lib.nerf_generate_rays(ctx.handle, C.byref(cam), int(first_pixel), n, _ptr(out), ctx.stream()))
    return out


def _render_frame_fused(ctx, cam, first_pixel, n_pixels, chunk, net_c, net_f, N_samples, N_importance, lindisp,
                        white_bkgd, shard=None):
    """One ``nerf_render_frame`` call: ray generation + chunk loop + render_rays, nothing but kernels enqueued.
    With ``shard = (world, rank)`` the pixel range is the rank's shard and the call is ``nerf_render_shard``."""
    if shard is not None:
        lo, cnt = C.c_int64(), C.c_int64()
        check(ctx.lib.nerf_shard_bounds(int(cam.H) * int(cam.W), int(shard[0]), int(shard[1]), C.byref(lo), C.byref(cnt)))
        first_pixel, n_pixels = lo.value, cnt.value
    o = dict(device=ctx.device, dtype=torch.float32)
    ret = {"rgb_map": torch.empty((n_pixels, 3), **o), "disp_map": torch.empty((n_pixels,), **o),
           "acc_map": torch.empty((n_pixels,), **o)}
    f = FrameArgs()
    f.cam, f.first_pixel, f.n_pixels, f.chunk = cam, int(first_pixel), int(n_pixels), int(chunk)
    f.N_samples, f.N_importance = int(N_samples), int(N_importance)
    f.slot_coarse, f.slot_fine = net_c.slot, (net_f.slot if net_f is not None else -1)
    f.lindisp, f.white_bkgd = int(bool(lindisp)), int(bool(white_bkgd))
    f.rgb_map, f.disp_map, f.acc_map = (ret[k].data_ptr() for k in ("rgb_map", "disp_map", "acc_map"))
    if N_importance > 0:
        ret.update(rgb0=torch.empty((n_pixels, 3), **o), disp0=torch.empty((n_pixels,), **o),
                   acc0=torch.empty((n_pixels,), **o), z_std=torch.empty((n_pixels,), **o))
        f.rgb0, f.disp0, f.acc0, f.z_std = (ret[k].data_ptr() for k in ("rgb0", "disp0", "acc0", "z_std"))
    f.stream = ctx.stream().value
    f.precision_guard = _lib.NERF_GUARD_FALLBACK      # a frame whose scale bound was loose comes back from the fp32 kernel
    if shard is not None:
        check(ctx.lib.nerf_render_shard(ctx.handle, C.byref(f), int(shard[0]), int(shard[1]), None, None))
    else:
        check(ctx.lib.nerf_render_frame(ctx.handle, C.byref(f)))
    return ret


def render_shard(H, W, K, world, rank, chunk=1024 * 32, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
                 c2w_staticcam=None, first_pixel=None, n_pixels=None, **kwargs):
    """Rank ``rank``'s contiguous shard of ``render(H, W, K, chunk, c2w=c2w, ...)`` (deterministic kwargs,
    ``render_kwargs_test``): one ``nerf_render_shard`` call - the rank generates and renders only its own pixels.
    Returns the flat ``{rgb_map, disp_map, acc_map, (rgb0, disp0, acc0, z_std)}`` dict of ``[n_shard, ...]`` tensors;
    row i is flat pixel ``shard_bounds(H*W, world, rank)[0] + i``. An explicit ``first_pixel`` / ``n_pixels`` range
    (any partition) goes through ``nerf_render_frame`` instead."""
    if c2w is None or not _frame_call_applies(kwargs) or \
            not kwargs['network_query_fn'].matches(kwargs['network_fn'], bool(use_viewdirs)):
        raise RuntimeError("render_shard needs c2w, this package's networks / NetworkQuery and deterministic kwargs "
                           "(perturb = 0, raw_noise_std = 0, retraw = False)")
    net = kwargs['network_fn']
    cam = _camera(H, W, K, c2w, ndc, near, far, use_viewdirs, c2w_staticcam)
    args = (chunk, net, kwargs.get('network_fine'), kwargs['N_samples'], kwargs.get('N_importance', 0),
            kwargs.get('lindisp', False), kwargs.get('white_bkgd', False))
    if first_pixel is not None:
        return _render_frame_fused(net.ctx, cam, int(first_pixel), int(n_pixels), *args)
    return _render_frame_fused(net.ctx, cam, 0, 0, *args, shard=(world, rank))


def _frame_call_applies(kwargs):
    """render() can hand the whole frame to one C call when nothing random or opaque is involved."""
    q, net, fine = kwargs.get('network_query_fn'), kwargs.get('network_fn'), kwargs.get('network_fine')
    try:
        noise = float(kwargs.get('raw_noise_std', 0.))
    except (TypeError, ValueError):
        noise = 0.0
    return (isinstance(q, NetworkQuery) and isinstance(net, NeRF) and (fine is None or isinstance(fine, NeRF))
            and float(kwargs.get('perturb', 0.)) == 0. and noise == 0. and not kwargs.get('retraw', False)
            and not kwargs.get('pytest', False))


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """``[rgb_map, disp_map, acc_map, extras]`` reshaped to the ray grid (nerf.ipynb:558-640).
    With ``c2w`` the rays are generated on the GPU (:func:`generate_rays`); a ``rays`` tuple is
    packed with torch ops exactly as the reference does."""
    model_device = next(kwargs['network_fn'].parameters()).device
    if c2w is not None and _frame_call_applies(kwargs) and \
            kwargs['network_query_fn'].matches(kwargs['network_fn'], bool(use_viewdirs)):
        net = kwargs['network_fn']
        cam = _camera(H, W, K, c2w, ndc, near, far, use_viewdirs, c2w_staticcam)
        all_ret = _render_frame_fused(net.ctx, cam, 0, int(H) * int(W), chunk, net, kwargs.get('network_fine'),
                                      kwargs['N_samples'], kwargs.get('N_importance', 0),
                                      kwargs.get('lindisp', False), kwargs.get('white_bkgd', False))
        for k in all_ret:
            all_ret[k] = torch.reshape(all_ret[k], [H, W] + list(all_ret[k].shape[1:]))
        k_extract = ['rgb_map', 'disp_map', 'acc_map']
        return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]
    if c2w is not None:
        packed = generate_rays(H, W, K, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, device=model_device)
        sh = (H, W, 3)
    else:
        packed, sh = pack_rays(H, W, K, rays, None, ndc, near, far, use_viewdirs, c2w_staticcam,
                               device=model_device)
    all_ret = batchify_rays(packed, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                calculate_metrics=False, metrics_include_lpips=True, metrics_device='cuda'):
    """Render a sequence of poses (nerf.ipynb:650-758): returns ``(rgbs, disps)`` as numpy arrays, or
    ``(rgbs, disps, avg_metrics)`` with ``calculate_metrics``; optionally writes 8-bit PNGs."""
    import os
    import time
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []
    want_metrics = calculate_metrics and gt_imgs is not None
    all_metrics = {'psnr': [], 'ssim': [], 'mse': []}
    if metrics_include_lpips:
        all_metrics['lpips'] = []
    t = time.time()
    for i, c2w in enumerate(render_poses):
        print(i, time.time() - t)
        t = time.time()
        rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        if i == 0:
            print(rgb.shape, disp.shape)
        if want_metrics and render_factor == 0 and i < len(gt_imgs):
            m = globals()['calculate_metrics'](rgb, gt_imgs[i], include_lpips=metrics_include_lpips,
                                               device=metrics_device)
            for key in all_metrics:
                if m.get(key) is not None:
                    all_metrics[key].append(m[key])
            print(f"Frame {i} - PSNR: {m['psnr']:.2f}, SSIM: {m['ssim']:.4f}")
        if savedir is not None:
            write_png(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
    rgbs, disps = np.stack(rgbs, 0), np.stack(disps, 0)
    if isinstance(render_kwargs.get('network_fn'), NeRF):
        _warn_if_scale_bound_was_loose(render_kwargs['network_fn'].ctx, "render_path")
    if want_metrics:
        avg = {}
        for key, values in all_metrics.items():
            if values:
                avg[f'avg_{key}'] = np.mean(values)
                avg[f'std_{key}'] = np.std(values)
        return rgbs, disps, avg
    return rgbs, disps


def write_png(path, img8):
    """Minimal 8-bit RGB/RGBA/gray PNG writer (stands in for ``imageio.imwrite``, nerf.ipynb:728)."""
    import struct
    import zlib
    img8 = np.ascontiguousarray(img8, dtype=np.uint8)
    if img8.ndim == 2:
        img8 = img8[..., None]
    h, w, c = img8.shape
    color = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + img8[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ----------------------------------------------------------------------------------------------
# model construction / checkpoints (nerf.ipynb:876-960)
# ----------------------------------------------------------------------------------------------

def load_checkpoint(path):
    """``torch.load`` of a reference checkpoint ``{global_step, network_fn_state_dict,
    network_fine_state_dict, optimizer_state_dict}`` (written at nerf.ipynb:1290-1299)."""
    return torch.load(path, map_location="cpu", weights_only=False)


def create_nerf(args, device=None):
    """``create_nerf`` (nerf.ipynb:876-960): embedders, coarse/fine models, the query function, the optimizer,
    checkpoint reload (weights, optimizer state, ``global_step``) and the train/test render kwargs. ``grad_vars`` is the
    list of models (their parameters live on the device)."""
    import os
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views, embeddirs_fn = 0, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, device=device)
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, device=device)
    network_query_fn = make_network_query_fn(embed_fn, embeddirs_fn, args.netchunk)
    grad_vars = [model] + ([model_fine] if model_fine is not None else [])
    optimizer = Adam(grad_vars, lr=getattr(args, "lrate", 5e-4), betas=(0.9, 0.999))     # nerf.ipynb:905
    start = 0
    ckpt_dir = os.path.join(args.basedir, args.expname, "checkpoints")
    ft_path = getattr(args, "ft_path", None)
    if ft_path is not None and ft_path != 'None':
        ckpts = [ft_path]
    else:
        ckpts = [os.path.join(ckpt_dir, f) for f in sorted(os.listdir(ckpt_dir)) if 'tar' in f] \
            if os.path.isdir(ckpt_dir) else []
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not getattr(args, "no_reload", False):
        print('Reloading from', ckpts[-1])
        ckpt = load_checkpoint(ckpts[-1])
        start = ckpt['global_step']
        model.load_state_dict(ckpt['network_fn_state_dict'])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])
        if ckpt.get('optimizer_state_dict'):
            optimizer.load_state_dict(ckpt['optimizer_state_dict'])                     # nerf.ipynb:925
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
        'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


# ----------------------------------------------------------------------------------------------
# training iteration (nerf.ipynb:905, 1258-1282)
# ----------------------------------------------------------------------------------------------

class Adam:
    """Stand-in for ``torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))``
    (nerf.ipynb:905) over this package's models: the moments live next to the master weights on the
    device and the update runs inside :func:`train_on_batch`. ``param_groups[0]['lr']`` is writable so
    the reference's decay loop (nerf.ipynb:1278-1282) works unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.models = [m for m in params if isinstance(m, NeRF)]
        if not self.models:
            raise ValueError("Adam expects the NeRF models themselves (their parameters live on the device)")
        self.param_groups = [{'lr': float(lr), 'betas': tuple(betas), 'eps': float(eps)}]
        self.steps = 0

    def zero_grad(self):
        pass                      # gradients are overwritten by every backward pass

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout over ``list(model.parameters()) + list(model_fine.parameters())``
        (nerf.ipynb:903-905), so the file written at nerf.ipynb:1290-1299 and this one are interchangeable: ``state[i]``
        = ``{'step', 'exp_avg', 'exp_avg_sq'}`` per parameter, in state-dict order model after model."""
        g = self.param_groups[0]
        state, idx = {}, 0
        for m in self.models:
            if self.steps > 0:
                em, ev = m.adam_state()
                for k in m.state_dict_keys():
                    state[idx] = {'step': torch.tensor(float(self.steps)), 'exp_avg': torch.from_numpy(em[k]),
                                  'exp_avg_sq': torch.from_numpy(ev[k])}
                    idx += 1
            else:
                idx += len(m.state_dict_keys())
        group = {'lr': g['lr'], 'betas': tuple(g['betas']), 'eps': g['eps'], 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(idx))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Accepts the layout above (i.e. the reference's ``ckpt['optimizer_state_dict']``, nerf.ipynb:925-932)."""
        if 'param_groups' in sd and sd['param_groups']:
            g = sd['param_groups'][0]
            self.param_groups[0].update({k: g[k] for k in ('lr', 'betas', 'eps') if k in g})
            self.param_groups[0]['betas'] = tuple(self.param_groups[0]['betas'])
        state = sd.get('state', {})
        if not state:
            self.steps = int(sd.get('steps', 0))
            return
        idx, steps = 0, 0
        for m in self.models:
            keys = m.state_dict_keys()
            entries = [state.get(idx + j) for j in range(len(keys))]
            idx += len(keys)
            # torch.optim.Adam keeps state only for parameters that ever received a gradient: with use_viewdirs=False
            # the reference's views_linears.0.* never do (nerf/nerf.py:43 registers them regardless), so its checkpoints
            # have no entry for them. A missing entry is a parameter whose moments are still zero.
            zeros = [np.zeros(m._expected_shape(k), np.float32) for k in keys]
            m.load_adam_state([z if e is None else e['exp_avg'].detach().cpu().numpy() for e, z in zip(entries, zeros)],
                              [z if e is None else e['exp_avg_sq'].detach().cpu().numpy() for e, z in zip(entries, zeros)])
            present = [e for e in entries if e is not None]
            if present:
                steps = max(steps, int(float(present[0]['step'])))
        if idx < len(state) or any(k >= idx for k in state):
            raise ValueError(f"optimizer state has entries for {max(state) + 1} parameters, the models have {idx}")
        self.steps = steps


def save_checkpoint(path, global_step, network_fn, network_fine, optimizer):
    """The file the reference writes every ``i_weights`` iterations (nerf.ipynb:1290-1299) and reloads in
    ``create_nerf`` (:925-935)."""
    _warn_if_scale_bound_was_loose(network_fn.ctx, "save_checkpoint")
    torch.save({'global_step': int(global_step),
                'network_fn_state_dict': network_fn.state_dict(),
                'network_fine_state_dict': network_fine.state_dict() if network_fine is not None else None,
                'optimizer_state_dict': optimizer.state_dict()}, path)


def train_on_batch(H, W, K, batch_rays, target_s, optimizer, chunk=1024 * 32, ndc=True, near=0., far=1.,
                   use_viewdirs=False, network_fn=None, network_query_fn=None, N_samples=64, N_importance=0,
                   network_fine=None, perturb=0., raw_noise_std=0., white_bkgd=False, lindisp=False, pytest=False,
                   apply_update=True, _packed_rays=None, _z_vals_fine=None, **unused):
    """One iteration of the reference's training loop body (nerf.ipynb:1258-1282) on the GPU:
    ``render(H, W, K, rays=batch_rays, retraw=True, **render_kwargs_train)``, ``img_loss =
    img2mse(rgb, target_s)`` (``+ img2mse(rgb0, target_s)``), ``loss.backward()``, ``optimizer.step()``.
    Pass the same ``render_kwargs_train`` (plus ``near``/``far``) as keyword arguments. Returns
    ``{'loss','img_loss','psnr'[, 'img_loss0','psnr0'], 'rgb'[, 'rgb0']}`` (tensors on the device)."""
    if not isinstance(network_fn, NeRF) or (network_fine is not None and not isinstance(network_fine, NeRF)):
        raise TypeError("train_on_batch needs this package's NeRF models")
    ctx = network_fn.ctx
    if _packed_rays is not None:      # (tests: a ray record exactly as the reference's render() packed it, e.g. NDC-warped)
        packed = _dev(_packed_rays, ctx).contiguous()
    else:
        packed = _pack_batch(ctx, H, W, K, batch_rays, ndc, near, far, use_viewdirs)
    target = _dev(target_s, ctx).reshape(-1, 3)
    N, stride = packed.shape
    if target.shape[0] != N:
        raise RuntimeError(f"target_s has {target.shape[0]} rows for {N} rays")
    Sc, Si = int(N_samples), int(N_importance)
    o = dict(device=ctx.device, dtype=torch.float32)
    a = TrainArgs()
    a.rays, a.target, a.n_rays, a.ray_stride = packed.data_ptr(), target.data_ptr(), N, stride
    a.N_samples, a.N_importance = Sc, Si
    # network_fine=None with N_importance > 0: both passes through network_fn (nerf.ipynb:471)
    a.slot_coarse, a.slot_fine = network_fn.slot, (network_fine.slot if Si > 0 and network_fine is not None else -1)
    a.lindisp, a.white_bkgd = int(bool(lindisp)), int(bool(white_bkgd))
    keep = [packed, target]
    perturb = float(perturb)
    try:
        noise_std = float(raw_noise_std)      # YAML may hand over the string '1e0'
    except (TypeError, ValueError):
        noise_std = 0.0
    n_c, n_i, n_f = N * Sc, (N * Si if Si > 0 else 0), (N * (Sc + Si) if Si > 0 else 0)
    if pytest:                                # the reference's RNG order, incl. its re-seeding before every draw
        def draw(shape):
            np.random.seed(0)
            return _dev(np.random.rand(*shape), ctx)
        t_rand = draw((N, Sc)) if perturb > 0. else None
        n0 = _noise((N, Sc), raw_noise_std, pytest, ctx)
        u = draw((N, Si)) if perturb > 0. and Si > 0 else None
        n1 = _noise((N, Sc + Si), raw_noise_std, pytest, ctx) if Si > 0 else None
    else:
        # every uniform of the iteration in ONE draw and every normal in another (t_rand | u and noise0 | noise as the two
        # halves of a flat buffer: the same distributions as the reference's four draws, two launches instead of four)
        t_rand = u = n0 = n1 = None
        if perturb > 0.:
            uni = torch.rand(n_c + n_i, **o)
            t_rand, u = uni[:n_c].view(N, Sc), (uni[n_c:].view(N, Si) if Si > 0 else None)
        if noise_std > 0.0:
            nrm = torch.randn(n_c + n_f, **o)
            if noise_std != 1.0:              # (every YAML of the reference has raw_noise_std 1e0: x * 1 = x)
                nrm = nrm * noise_std
            n0, n1 = nrm[:n_c].view(N, Sc), (nrm[n_c:].view(N, Sc + Si) if Si > 0 else None)
    if perturb > 0.:
        a.perturb = 1
    for name, t in (("t_rand", t_rand), ("noise0", n0), ("u_rand", u), ("noise", n1)):
        if t is not None:
            keep.append(t)
            setattr(a, name, t.data_ptr())
    g = optimizer.param_groups[0]
    a.lr, (a.beta1, a.beta2), a.eps = g['lr'], g['betas'], g['eps']
    a.apply_update = int(bool(apply_update))
    if apply_update:
        optimizer.steps += 1
    a.step = max(optimizer.steps, 1)
    # img_loss, img_loss0, loss, psnr, psnr0 (nerf.ipynb:1262-1272) come back as five numbers of one device array: no tensor
    # operation per number on this side (mse2psnr's log / multiply / divide and the sum of the losses were four launches)
    stats = torch.empty(5, **o)
    rgb = torch.empty((N, 3), **o)
    rgb0 = torch.empty((N, 3), **o) if Si > 0 else None
    a.stats, a.rgb_map, a.rgb0 = stats.data_ptr(), rgb.data_ptr(), (rgb0.data_ptr() if Si > 0 else None)
    if _z_vals_fine is not None and Si > 0:      # (parity tests: the fine pass at the reference's fine depths)
        zin = _dev(_z_vals_fine, ctx).reshape(N, Sc + Si)
        keep.append(zin)
        a.z_vals_fine_in = zin.data_ptr()
    a.stream = ctx.stream().value
    check(ctx.lib.nerf_train_step(ctx.handle, C.byref(a)))
    out = {'img_loss': stats[0], 'psnr': stats[3], 'rgb': rgb, 'loss': stats[2]}
    if Si > 0:
        out.update(img_loss0=stats[1], psnr0=stats[4], rgb0=rgb0)
    return out


def _pack_batch(ctx, H, W, K, rays, ndc, near, far, use_viewdirs):
    """``render(H, W, K, rays=batch_rays, ...)``'s packing of a ray batch (nerf.ipynb:596-629) in ONE kernel
    (``nerf_pack_rays``): unit viewing directions before the NDC warp, ``ndc_rays(H, W, K[0][0], 1., ...)``, near / far
    columns. ``rays`` is the ``(rays_o, rays_d)`` pair or the stacked ``[2, N, 3]`` tensor; rows may be strided views."""
    rays_o, rays_d = rays
    sh = tuple(rays_d.shape)

    def rows(t):
        if not torch.is_tensor(t):
            t = torch.as_tensor(np.asarray(t))
        t = t.detach().to(device=ctx.device, dtype=torch.float32).reshape(-1, 3) if t.dim() != 2 else \
            t.detach().to(device=ctx.device, dtype=torch.float32)
        if t.stride(1) != 1 or t.stride(0) < 3:
            t = t.contiguous()
        return t

    ro, rd = rows(rays_o), rows(rays_d)
    if ro.shape != rd.shape or ro.shape[-1] != 3:
        raise RuntimeError(f"rays_o {tuple(ro.shape)} and rays_d {tuple(rd.shape)} must both be [N, 3]")
    cam = Camera()
    cam.H, cam.W = int(H), int(W)
    cam.ndc = int(bool(ndc))
    cam.ndc_focal = float(K[0][0]) if ndc else 0.0
    cam.near, cam.far, cam.use_viewdirs = float(near), float(far), int(bool(use_viewdirs))
    n = ro.shape[0]
    out = torch.empty((n, 11 if use_viewdirs else 8), device=ctx.device, dtype=torch.float32)
    check(ctx.lib.nerf_pack_rays(ctx.handle, C.byref(cam), _ptr(ro), ro.stride(0), _ptr(rd), rd.stride(0), n, _ptr(out),
                                 ctx.stream()))
    return out


# ----------------------------------------------------------------------------------------------
# image metrics (nerf_helpers.py:8-214)
# ----------------------------------------------------------------------------------------------

def _image_metrics(img1, img2, max_val=1.0):
    ctx = get_context(img1.device if torch.is_tensor(img1) and img1.is_cuda else
                      (img2.device if torch.is_tensor(img2) and img2.is_cuda else None))
    a, b = _dev(img1, ctx), _dev(img2, ctx)
    if a.dim() != 3 or a.shape != b.shape or a.shape[-1] != 3:
        raise ValueError(f"Expected 3D tensor [H, W, C], got {tuple(a.shape)} and {tuple(b.shape)}")
    out = torch.empty(2, device=ctx.device, dtype=torch.float32)
    check(ctx.lib.nerf_image_metrics(ctx.handle, _ptr(a), _ptr(b), a.shape[0], a.shape[1], float(max_val),
                                     _ptr(out), ctx.stream()))
    return out


def calculate_ssim(img1, img2, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    """Mean SSIM of two ``[H,W,3]`` images (nerf_helpers.py:21-111). The kernel implements the
    reference's defaults (11-tap, sigma 1.5, k1 0.01, k2 0.03); other settings raise."""
    if (filter_size, filter_sigma, k1, k2) != (11, 1.5, 0.01, 0.03) or return_map:
        raise NotImplementedError("calculate_ssim: only the reference defaults are implemented on the GPU")
    return float(_image_metrics(img1, img2, max_val)[0].item())


def calculate_lpips(img1, img2, net='vgg', device='cuda', normalize=True):
    """nerf_helpers.py:114-146 needs the ``lpips`` package and its VGG weights (a remote fetch);
    neither exists offline, so this raises exactly as the reference does when ``lpips`` is missing."""
    import lpips  # noqa: F401  (ImportError, as in the reference)
    raise RuntimeError("LPIPS is outside this build's scope (SURVEY.md section 2 row 7)")


def calculate_metrics(img1, img2, include_lpips=True, lpips_net='vgg', device='cuda'):
    """``{'mse','psnr','ssim'[, 'lpips']}`` (nerf_helpers.py:148-214); SSIM and MSE come from one GPU pass."""
    out = _image_metrics(img1, img2, 1.0).cpu()
    mse = float(out[1])
    metrics = {'mse': mse, 'psnr': float(mse2psnr(torch.tensor(mse))), 'ssim': float(out[0])}
    if include_lpips:
        try:
            metrics['lpips'] = calculate_lpips(img1, img2, net=lpips_net, device=device)
        except Exception as e:      # same soft failure as the reference (nerf_helpers.py:207-212)
            print(f"Warning: Could not calculate LPIPS: {e}")
            metrics['lpips'] = None
    return metrics


# ----------------------------------------------------------------------------------------------
# metrics used by the PSNR report (nerf_helpers.py:8-18)
# ----------------------------------------------------------------------------------------------

def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10.0 * torch.log(x) / math.log(10.0)


def to8b(x):
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)
