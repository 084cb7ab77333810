"""Random-ray batching of the training loop (reference: train(), nerf/nerf.ipynb cell 19, section 6 and the head
of the main loop; raw-JSON lines 1186-1276 of the cell source are cited below as ``train:NNN``).

Two modes, as in the reference:

* ``use_batching`` (``no_batching: False``, train:190-203, 230-242): the rays of every training image are generated
  once, concatenated with the pixel colours into ``rays_rgb [n_train*H*W, 3, 3]``, shuffled, and consumed in
  consecutive windows of ``N_rand``; reshuffled with ``torch.randperm`` after an epoch.
* per-image (``no_batching: True``, train:243-276): a random training image per iteration, optional centre crop
  during the first ``precrop_iters`` iterations, ``N_rand`` distinct pixels of it.

The random draws are the reference's, in the reference's order (``np.random.shuffle`` / ``np.random.choice`` on the
numpy stream, ``torch.randperm`` on torch's), so seeding both streams reproduces its batches; a private
``numpy.random.RandomState`` may be passed instead of the global numpy stream. Everything else stays on the
tensors' device: rays come from ``get_rays`` (nerf_helpers.py:222-296) and are cached per image.
"""
import numpy as np
import torch

from .host import get_rays


class RayBatcher:
    def __init__(self, images, poses, H, W, K, i_train, N_rand, use_batching=True, precrop_iters=0,
                 precrop_frac=0.5, device=None, rng=None):
        self.H, self.W, self.K = int(H), int(W), K
        self.i_train = np.asarray(i_train)
        self.N_rand = N_rand
        self.use_batching = bool(use_batching)
        self.precrop_iters, self.precrop_frac = precrop_iters, precrop_frac
        self.rng = np.random if rng is None else rng
        if device is None:
            device = images.device if torch.is_tensor(images) else "cpu"
        self.device = torch.device(device)
        as_t = lambda x: (x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))).float().to(self.device)
        self.images = as_t(images)[..., :3]
        self.poses = as_t(poses)
        self._rays = {}
        self.i_batch = 0
        self.rays_rgb = None
        if self.use_batching:
            # train:193-201 - [n_train, H, W, ro+rd+rgb, 3] flattened to [n_train*H*W, 3, 3], then np.random.shuffle.
            # (shuffling an index vector draws the same numbers as shuffling the rows themselves)
            per_image = []
            for i in self.i_train:
                ro, rd = self._image_rays(int(i))
                per_image.append(torch.stack([ro, rd, self.images[int(i)]], dim=2))   # [H, W, 3, 3]
            rays_rgb = torch.stack(per_image, 0).reshape(-1, 3, 3)
            perm = np.arange(rays_rgb.shape[0])
            self.rng.shuffle(perm)
            self.rays_rgb = rays_rgb[torch.from_numpy(perm).to(self.device)]
            self._rays.clear()

    def _image_rays(self, img_i):
        if img_i not in self._rays:
            self._rays[img_i] = get_rays(self.H, self.W, self.K, self.poses[img_i, :3, :4])
        return self._rays[img_i]

    def next(self, i, start=0):
        """Batch of iteration ``i`` -> ``(batch_rays [2, N_rand, 3], target_s [N_rand, 3])`` (train:229-276)."""
        if self.use_batching:
            batch = self.rays_rgb[self.i_batch:self.i_batch + self.N_rand]
            batch = torch.transpose(batch, 0, 1)
            batch_rays, target_s = batch[:2], batch[2]
            self.i_batch += self.N_rand
            if self.i_batch >= self.rays_rgb.shape[0]:
                rand_idx = torch.randperm(self.rays_rgb.shape[0])            # CPU generator, as the reference
                self.rays_rgb = self.rays_rgb[rand_idx.to(self.device)]
                self.i_batch = 0
            return batch_rays, target_s
        img_i = int(self.rng.choice(self.i_train))
        target = self.images[img_i]
        rays_o, rays_d = self._image_rays(img_i)
        if self.N_rand is None:
            return torch.stack([rays_o, rays_d], 0), target
        H, W = self.H, self.W
        if i < self.precrop_iters:
            dH = int(H // 2 * self.precrop_frac)
            dW = int(W // 2 * self.precrop_frac)
            rows = torch.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH)
            cols = torch.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW)
        else:
            rows, cols = torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W)
        coords = torch.stack(torch.meshgrid(rows, cols, indexing="ij"), -1).reshape(-1, 2)
        select_inds = self.rng.choice(coords.shape[0], size=[self.N_rand], replace=False)
        select_coords = coords[select_inds].long().to(self.device)
        r, c = select_coords[:, 0], select_coords[:, 1]
        return torch.stack([rays_o[r, c], rays_d[r, c]], 0), target[r, c]
