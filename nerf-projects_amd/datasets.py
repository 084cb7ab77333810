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

def load_blender_data(basedir, half_res=False, testskip=1):
    """``(imgs [N,H,W,4] float32 in [0,1], poses [N,4,4], render_poses [40,4,4], [H, W, focal], i_split)``."""
    splits = ['train', 'val', 'test']
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, 'transforms_{}.json'.format(s)), 'r') as fp:
            metas[s] = json.load(fp)
    all_imgs, all_poses, counts = [], [], [0]
    for s in splits:
        meta = metas[s]
        skip = 1 if (s == 'train' or testskip == 0) else testskip
        imgs, poses = [], []
        for frame in meta['frames'][::skip]:
            imgs.append(read_png(os.path.join(basedir, frame['file_path'] + '.png')))
            poses.append(np.array(frame['transform_matrix']))
        imgs = (np.array(imgs) / 255.).astype(np.float32)          # keep all 4 channels (RGBA)
        poses = np.array(poses).astype(np.float32)
        counts.append(counts[-1] + imgs.shape[0])
        all_imgs.append(imgs)
        all_poses.append(poses)
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    H, W = imgs[0].shape[:2]
    camera_angle_x = float(meta['camera_angle_x'])
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    render_poses = np.stack([pose_spherical(angle, -30.0, 4.0) for angle in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        imgs = np.stack([_area_half(img.astype(np.float64)) for img in imgs], 0)   # float64, like np.zeros at :81
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
    """load_llff.py:65-139 for the ``factor`` form used by ``load_llff_data``."""
    poses_arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    poses = poses_arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = poses_arr[:, -2:].transpose([1, 0])
    sfx = ''
    if factor is not None:
        sfx = '_{}'.format(factor)
        _minify(basedir, factor)
    else:
        factor = 1
    imgdir = os.path.join(basedir, 'images' + sfx)
    if not os.path.exists(imgdir):
        raise RuntimeError(f"{imgdir} does not exist")
    imgfiles = _image_files(imgdir)
    if poses.shape[-1] != len(imgfiles):
        raise RuntimeError('Mismatch between imgs {} and poses {} !!!!'.format(len(imgfiles), poses.shape[-1]))
    sh = read_png(imgfiles[0]).shape
    poses[:2, 4, :] = np.array(sh[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    if not load_imgs:
        return poses, bds
    imgs = np.stack([read_png(f)[..., :3] / 255. for f in imgfiles], -1)
    return poses, bds, imgs


def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    vec2 = normalize(z)
    vec0 = normalize(np.cross(up, vec2))
    vec1 = normalize(np.cross(vec2, vec0))
    return np.stack([vec0, vec1, vec2, pos], 1)


def poses_avg(poses):
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    vec2 = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(vec2, up, center), hwf], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    render_poses = []
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    for theta in np.linspace(0., 2. * np.pi * rots, N + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        z = normalize(c - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.])))
        render_poses.append(np.concatenate([viewmatrix(z, up, c), hwf], 1))
    return render_poses


def recenter_poses(poses):
    poses_ = poses + 0
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    c2w = poses_avg(poses)
    c2w = np.concatenate([c2w[:3, :4], bottom], -2)
    bottom = np.tile(np.reshape(bottom, [1, 1, 4]), [poses.shape[0], 1, 1])
    poses = np.concatenate([poses[:, :3, :4], bottom], -2)
    poses = np.linalg.inv(c2w) @ poses
    poses_[:, :3, :4] = poses[:, :3, :4]
    return poses_


def spherify_poses(poses, bds):
    def p34_to_44(p):
        return np.concatenate([p, np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [p.shape[0], 1, 1])], 1)
    rays_d = poses[:, :3, 2:3]
    rays_o = poses[:, :3, 3:4]

    def min_line_dist(rays_o, rays_d):
        A_i = np.eye(3) - rays_d * np.transpose(rays_d, [0, 2, 1])
        b_i = -A_i @ rays_o
        return np.squeeze(-np.linalg.inv((np.transpose(A_i, [0, 2, 1]) @ A_i).mean(0)) @ (b_i).mean(0))
    center = min_line_dist(rays_o, rays_d)
    up = (poses[:, :3, 3] - center).mean(0)
    vec0 = normalize(up)
    vec1 = normalize(np.cross([.1, .2, .3], vec0))
    vec2 = normalize(np.cross(vec0, vec1))
    c2w = np.stack([vec1, vec2, vec0, center], 1)
    poses_reset = np.linalg.inv(p34_to_44(c2w[None])) @ p34_to_44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(poses_reset[:, :3, 3]), -1)))
    sc = 1. / rad
    poses_reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    centroid = np.mean(poses_reset[:, :3, 3], 0)
    zh = centroid[2]
    radcircle = np.sqrt(rad ** 2 - zh ** 2)
    new_poses = []
    for th in np.linspace(0., 2. * np.pi, 120):
        camorigin = np.array([radcircle * np.cos(th), radcircle * np.sin(th), zh])
        up = np.array([0, 0, -1.])
        vec2 = normalize(camorigin)
        vec0 = normalize(np.cross(vec2, up))
        vec1 = normalize(np.cross(vec2, vec0))
        new_poses.append(np.stack([vec0, vec1, vec2, camorigin], 1))
    new_poses = np.stack(new_poses, 0)
    new_poses = np.concatenate([new_poses, np.broadcast_to(poses[0, :3, -1:], new_poses[:, :3, -1:].shape)], -1)
    poses_reset = np.concatenate([poses_reset[:, :3, :4],
                                  np.broadcast_to(poses[0, :3, -1:], poses_reset[:, :3, -1:].shape)], -1)
    return poses_reset, new_poses, bds


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """``(images [N,H,W,3], poses [N,3,5], bds [N,2], render_poses, i_test)`` (load_llff.py:242-315)."""
    poses, bds, imgs = _load_data(basedir, factor=factor)
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        c2w = poses_avg(poses)
        up = normalize(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / (((1. - dt) / close_depth + dt / inf_depth))
        zdelta = close_depth * .2
        tt = poses[:, :3, 3]
        rads = np.percentile(np.abs(tt), 90, 0)
        c2w_path = c2w
        N_views, N_rots = 120, 2
        if path_zflat:
            zloc = -close_depth * .1
            c2w_path[:3, 3] = c2w_path[:3, 3] + zloc * c2w_path[:3, 2]
            rads[2] = 0.
            N_rots = 1
            N_views //= 2
        render_poses = render_path_spiral(c2w_path, up, rads, focal, zdelta, zrate=.5, rots=N_rots, N=N_views)
    render_poses = np.array(render_poses).astype(np.float32)
    c2w = poses_avg(poses)
    dists = np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)
    i_test = np.argmin(dists)
    return images.astype(np.float32), poses.astype(np.float32), bds, render_poses, i_test
