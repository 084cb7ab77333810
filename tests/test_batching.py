"""Random-ray batching of the training loop (nerf/nerf.ipynb cell 19, section 6 + loop head) against the oracle's
numpy restatement, with the same numpy random stream: identical pixels, identical rays. CPU only."""
import numpy as np
import torch

from conftest import _oracle


def _scene(n_img=5, H=12, W=16, seed=0):
    from nerf_projects_amd import synthetic
    rs = np.random.RandomState(seed)
    images = rs.rand(n_img, H, W, 4).astype(np.float32)[..., :3]
    poses = np.stack([np.asarray(synthetic.pose_spherical(40.0 * k - 90, -30.0, 4.0), np.float32) for k in range(n_img)])
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], np.float32)
    return images, poses, H, W, K


def test_global_batching_matches_reference_logic():
    import nerf_projects_amd as N
    O = _oracle()
    images, poses, H, W, K = _scene()
    i_train = np.array([0, 2, 3])
    want = O.global_ray_batches(images, poses, H, W, K, i_train, 64, 4, np.random.RandomState(7))
    b = N.RayBatcher(images, poses, H, W, K, i_train, 64, use_batching=True, rng=np.random.RandomState(7))
    assert b.rays_rgb.shape == (3 * H * W, 3, 3)
    for k in range(4):
        rays, target = b.next(k)
        assert rays.shape == (2, 64, 3) and target.shape == (64, 3)
        np.testing.assert_allclose(rays.numpy(), want[k][0], rtol=0, atol=2e-6)
        np.testing.assert_array_equal(target.numpy(), want[k][1])


def test_global_batching_reshuffles_after_an_epoch():
    import nerf_projects_amd as N
    images, poses, H, W, K = _scene(n_img=2, H=4, W=4)
    b = N.RayBatcher(images, poses, H, W, K, [0, 1], 16, use_batching=True, rng=np.random.RandomState(1))
    seen = [b.next(k)[1].numpy() for k in range(2)]                  # one epoch = 32 rays
    assert b.i_batch == 0
    first = np.concatenate(seen).reshape(-1, 3)
    torch.manual_seed(3)
    again = np.concatenate([b.next(k)[1].numpy() for k in range(2)]).reshape(-1, 3)
    key = lambda a: sorted(map(tuple, np.round(a, 6)))
    assert key(first) == key(again) == key(images[[0, 1]].reshape(-1, 3))       # a permutation of every pixel


def test_per_image_batching_matches_reference_logic():
    import nerf_projects_amd as N
    O = _oracle()
    images, poses, H, W, K = _scene()
    i_train = np.array([1, 2, 4])
    rs_a, rs_b = np.random.RandomState(11), np.random.RandomState(11)
    b = N.RayBatcher(images, poses, H, W, K, i_train, 24, use_batching=False, precrop_iters=2, precrop_frac=0.5, rng=rs_a)
    for i in range(4):                      # iterations 0, 1 are centre-cropped
        rays, target = b.next(i)
        w_rays, w_target, img_i, sel = O.per_image_ray_batch(images, poses, H, W, K, i_train, 24, i, 2, 0.5, rs_b)
        np.testing.assert_allclose(rays.numpy(), w_rays, rtol=0, atol=2e-6)
        np.testing.assert_array_equal(target.numpy(), w_target)
        assert len({tuple(s) for s in sel}) == 24                     # distinct pixels
        if i < 2:
            assert sel[:, 0].min() >= H // 2 - H // 4 and sel[:, 0].max() <= H // 2 + H // 4 - 1
            assert sel[:, 1].min() >= W // 2 - W // 4 and sel[:, 1].max() <= W // 2 + W // 4 - 1


def test_per_image_batching_uses_the_global_numpy_stream_by_default():
    import nerf_projects_amd as N
    O = _oracle()
    images, poses, H, W, K = _scene()
    i_train = np.array([0, 1, 2, 3])
    np.random.seed(5)
    rays, target = N.RayBatcher(images, poses, H, W, K, i_train, 8, use_batching=False).next(10)
    np.random.seed(5)
    w_rays, w_target, *_ = O.per_image_ray_batch(images, poses, H, W, K, i_train, 8, 10, 0, 0.5, np.random)
    np.testing.assert_allclose(rays.numpy(), w_rays, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(target.numpy(), w_target)
