"""Dataset ingestion for the render path: ``load_blender_data``
(nerf/load_blender.py:37-89) and ``load_llff_data`` (nerf/load_llff.py:242-315) with the reference's
return tuples, so real lego / fern frames can be rendered and scored when the data exists.

The reference reads images with ``imageio`` and resizes with ``cv2`` / ImageMagick; neither is in this
image, so PNG files are decoded by a small pure-Python reader (8-bit, non-interlaced - what Blender and
``mogrify -format png`` write). JPEG is not decodable here: LLFF scenes must already contain their
``images_<factor>`` PNG folder (the reference creates it with ``mogrify``, which is attempted as well).
Camera / pose math is numpy, restated function by function.
"""
import json
import os
import struct
import zlib

import numpy as np

from .synthetic import pose_spherical

__all__ = ["read_png", "load_blender_data", "load_llff_data", "recenter_poses", "spherify_poses",
           "render_path_spiral", "poses_avg", "viewmatrix", "normalize"]


# ----------------------------------------------------------------------------------------------
# PNG
# ----------------------------------------------------------------------------------------------

def read_png(path):
    """uint8 array [H,W] / [H,W,2|3|4] of an 8-bit non-interlaced PNG (stands in for ``imageio.imread``)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file (JPEG and other formats cannot be decoded in this image)")
    pos, idat, ihdr, plte = 8, [], None, None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IEND":
            break
        pos += 12 + n
    w, h, depth, color, _, _, interlace = ihdr
    if depth != 8 or interlace != 0:
        raise ValueError(f"{path}: only 8-bit non-interlaced PNGs are supported (depth {depth}, interlace {interlace})")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    stride = w * ch
    raw = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for r in range(h):
        ft = int(raw[r, 0])
        line = raw[r, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):
            cur = np.zeros(stride, np.int32)
            for i in range(stride):             # serial dependency on the pixel to the left
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        else:
            raise ValueError(f"{path}: bad PNG filter type {ft}")
        out[r] = cur
        prev = cur
    img = out.reshape(h, w, ch)
    if color == 3:
        img = plte[img[..., 0]]
    return img[..., 0] if img.shape[-1] == 1 else img


def _area_half(img):
    """``cv2.resize(img, (W//2, H//2), interpolation=cv2.INTER_AREA)`` for an exact 2x reduction: the mean
    of each 2x2 block (load_blender.py:83-84)."""
    h, w = img.shape[0] // 2 * 2, img.shape[1] // 2 * 2
    x = img[:h, :w]
    return 0.25 * (x[0::2, 0::2] + x[1::2, 0::2] + x[0::2, 1::2] + x[1::2, 1::2])


# ----------------------------------------------------------------------------------------------
# Blender (nerf/load_blender.py:37-89)
# ----------------------------------------------------------------------------------------------

def _blender_split(basedir, split, stride):
    """Images (RGBA, float32 in [0,1]) and camera-to-world matrices of one split, every `stride`-th frame."""
    with open(os.path.join(basedir, f'transforms_{split}.json')) as fp:
        meta = json.load(fp)
    frames = meta['frames'][::stride]
    imgs = np.stack([read_png(os.path.join(basedir, fr['file_path'] + '.png')) for fr in frames])
    poses = np.stack([np.asarray(fr['transform_matrix']) for fr in frames])
    return (imgs / 255.).astype(np.float32), poses.astype(np.float32), float(meta['camera_angle_x'])


def load_blender_data(basedir, half_res=False, testskip=1):
    """``(imgs [N,H,W,4] in [0,1], poses [N,4,4], render_poses [40,4,4], [H, W, focal], i_split)`` for a
    NeRF-synthetic scene: train frames in full, val/test every `testskip`-th (0 = all), the 40-view
    spherical render path, optional 2x area down-sampling (nerf/load_blender.py:37-89)."""
    per_split, offsets = [], [0]
    for split in ('train', 'val', 'test'):
        stride = 1 if (split == 'train' or testskip == 0) else testskip
        imgs, poses, angle = _blender_split(basedir, split, stride)
        per_split.append((imgs, poses))
        offsets.append(offsets[-1] + len(imgs))
    i_split = [np.arange(offsets[k], offsets[k + 1]) for k in range(3)]
    imgs = np.concatenate([p[0] for p in per_split], 0)
    poses = np.concatenate([p[1] for p in per_split], 0)
    H, W = imgs.shape[1:3]
    focal = .5 * W / np.tan(.5 * angle)                     # camera_angle_x of the last split read (:71-73)
    render_poses = np.stack([pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 41)[:-1]], 0)
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        imgs = np.stack([_area_half(im.astype(np.float64)) for im in imgs], 0)   # float64 like np.zeros at :81
    return imgs, poses, render_poses, [H, W, focal], i_split


# ----------------------------------------------------------------------------------------------
# LLFF (nerf/load_llff.py)
# ----------------------------------------------------------------------------------------------

def _image_files(d):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(('JPG', 'jpg', 'png'))]


def _minify(basedir, factor):
    """load_llff.py:8-62: the reference shells out to ImageMagick; try the same, else explain."""
    imgdir = os.path.join(basedir, 'images_{}'.format(factor))
    if os.path.exists(imgdir):
        return
    from shutil import which
    if which("mogrify") is None:
        raise RuntimeError(f"{imgdir} does not exist and ImageMagick's `mogrify` (which the reference uses to "
                           "create it) is not installed; provide the down-sampled PNG folder")
    from subprocess import check_output
    src = os.path.join(basedir, 'images')
    os.makedirs(imgdir)
    check_output('cp {}/* {}'.format(src, imgdir), shell=True)
    ext = _image_files(src)[0].split('.')[-1]
    check_output(' '.join(['mogrify', '-resize', '{}%'.format(100. / factor), '-format', 'png', '*.{}'.format(ext)]),
                 shell=True, cwd=imgdir)
    if ext != 'png':
        check_output('rm {}/*.{}'.format(imgdir, ext), shell=True)


def _load_data(basedir, factor=None, load_imgs=True):
    """Raw LLFF arrays: ``poses [3,5,N]`` (rotation | translation | hwf), ``bds [2,N]`` and, optionally,
    ``imgs [H,W,3,N]`` in [0,1] from ``images_<factor>`` (load_llff.py:65-139, the ``factor`` form)."""
    table = np.load(os.path.join(basedir, 'poses_bounds.npy'))            # [N, 17] = 3x5 pose + near/far
    poses = np.moveaxis(table[:, :15].reshape(-1, 3, 5), 0, -1).copy()
    bds = table[:, 15:].T.copy()
    if factor is not None:
        _minify(basedir, factor)
    imgdir = os.path.join(basedir, 'images' if factor is None else f'images_{factor}')
    if not os.path.exists(imgdir):
        raise RuntimeError(f"{imgdir} does not exist")
    files = _image_files(imgdir)
    if len(files) != poses.shape[-1]:
        raise RuntimeError('Mismatch between imgs {} and poses {} !!!!'.format(len(files), poses.shape[-1]))
    first = read_png(files[0])
    poses[0, 4, :], poses[1, 4, :] = first.shape[0], first.shape[1]       # h, w of the images actually used
    poses[2, 4, :] /= (1 if factor is None else factor)                   # focal shrinks with the images
    if not load_imgs:
        return poses, bds
    imgs = np.stack([read_png(f)[..., :3] / 255. for f in files], -1)
    return poses, bds, imgs


def normalize(x):
    """Unit vector (load_llff.py `normalize`)."""
    x = np.asarray(x)
    return x / np.sqrt(np.sum(x * x))


def viewmatrix(z, up, pos):
    """Right-handed camera frame [x | y | z | origin] as a [3,4] matrix whose z axis is `z` and whose y axis is
    `up` made orthogonal to it (load_llff.py `viewmatrix`)."""
    fwd = normalize(z)
    right = normalize(np.cross(up, fwd))
    true_up = normalize(np.cross(fwd, right))
    return np.column_stack([right, true_up, fwd, pos])


def poses_avg(poses):
    """'Average' camera of a set of [N,3,5] poses: mean position, summed viewing and up directions, and
    the first pose's hwf column (load_llff.py `poses_avg`)."""
    frame = viewmatrix(poses[:, :3, 2].sum(0), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([frame, poses[0, :3, -1:]], 1)


def _homogeneous(m34):
    """[...,3,4] -> [...,4,4] with a (0,0,0,1) bottom row."""
    m34 = np.asarray(m34)
    bottom = np.broadcast_to(np.array([0., 0., 0., 1.]), m34.shape[:-2] + (1, 4))
    return np.concatenate([m34, bottom], -2)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """N cameras on a spiral around the average pose, all looking at the point `focal` in front of it
    (load_llff.py `render_path_spiral`; `zdelta` is unused there as well)."""
    scale = np.append(np.asarray(rads, dtype=np.float64), 1.0)
    frame, hwf = c2w[:3, :4], c2w[:, 4:5]
    target = frame @ np.array([0., 0., -focal, 1.])
    out = []
    for theta in np.linspace(0., 2. * np.pi * rots, N + 1)[:-1]:
        eye = frame @ (np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * scale)
        out.append(np.concatenate([viewmatrix(eye - target, up, eye), hwf], 1))
    return out


def recenter_poses(poses):
    """Express all poses in the frame of their average camera (load_llff.py `recenter_poses`)."""
    out = poses.copy()
    to_avg = np.linalg.inv(_homogeneous(poses_avg(poses)[:3, :4]))
    out[:, :3, :4] = (to_avg @ _homogeneous(poses[:, :3, :4]))[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """360-degree captures: recentre on the point closest to all optical axes, scale the camera shell to unit
    radius (``bds`` is rescaled in place, as in the reference) and emit a 120-view circular path
    (load_llff.py `spherify_poses`)."""
    axes = poses[:, :3, 2:3]                       # [N,3,1] viewing directions
    origins = poses[:, :3, 3:4]
    # least-squares point nearest to the lines origin + t*axis
    proj = np.eye(3) - axes * np.swapaxes(axes, 1, 2)
    lhs = (np.swapaxes(proj, 1, 2) @ proj).mean(0)
    rhs = (-proj @ origins).mean(0)
    centre = np.squeeze(-np.linalg.inv(lhs) @ rhs)

    z_axis = normalize((poses[:, :3, 3] - centre).mean(0))
    x_axis = normalize(np.cross([.1, .2, .3], z_axis))
    y_axis = normalize(np.cross(z_axis, x_axis))
    world = np.column_stack([x_axis, y_axis, z_axis, centre])
    reset = np.linalg.inv(_homogeneous(world[None])) @ _homogeneous(poses[:, :3, :4])

    radius = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    shrink = 1. / radius
    reset[:, :3, 3] *= shrink
    bds *= shrink
    radius *= shrink
    height = np.mean(reset[:, :3, 3], 0)[2]
    ring = np.sqrt(radius ** 2 - height ** 2)
    circle = []
    for th in np.linspace(0., 2. * np.pi, 120):
        eye = np.array([ring * np.cos(th), ring * np.sin(th), height])
        fwd = normalize(eye)
        right = normalize(np.cross(fwd, np.array([0, 0, -1.])))
        circle.append(np.column_stack([right, normalize(np.cross(fwd, right)), fwd, eye]))
    circle = np.stack(circle, 0)
    hwf = poses[0, :3, -1:]
    circle = np.concatenate([circle, np.broadcast_to(hwf, circle[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, circle, bds


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """``(images [N,H,W,3], poses [N,3,5], bds [N,2], render_poses, i_test)`` (load_llff.py:242-315):
    axis convention fix-up, depth-bound rescale by ``bd_factor``, recentring, then either the spherified
    circle or the forward-facing spiral as render path, and the hold-out view nearest the average pose."""
    poses, bds, imgs = _load_data(basedir, factor=factor)
    # LLFF stores [down, right, back]; NeRF wants [right, up, back]
    poses = np.concatenate([poses[:, 1:2], -poses[:, 0:1], poses[:, 2:]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    scale = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= scale
    bds *= scale
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        centre = poses_avg(poses)
        up = normalize(poses[:, :3, 1].sum(0))
        near_d, far_d = bds.min() * .9, bds.max() * 5.
        focus = 1. / (.25 / near_d + .75 / far_d)                 # dt = .75 blend of the disparities
        radii = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        views, turns = 120, 2
        if path_zflat:
            centre[:3, 3] += -near_d * .1 * centre[:3, 2]
            radii[2] = 0.
            views, turns = 60, 1
        render_poses = render_path_spiral(centre, up, radii, focus, near_d * .2, zrate=.5, rots=turns, N=views)
    render_poses = np.array(render_poses).astype(np.float32)
    mean_pos = poses_avg(poses)[:3, 3]
    i_test = np.argmin(np.sum(np.square(mean_pos - poses[:, :3, 3]), -1))
    return images, poses.astype(np.float32), bds, render_poses, i_test
