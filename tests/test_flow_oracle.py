"""CPU: oracle/tvl1_oracle.py -- the published TV-L1 algorithm restated (PARITY UNPINNED: the reference's flow comes from
a third-party binary, build_wof_clips.py:70-73, and the tree holds neither frames nor flow images).  What can be pinned
here are known answers of the algorithm itself: a translated texture yields that translation, identical frames yield zero
flow, the building blocks are mutually adjoint, the 8-bit quantisation follows the -b 20 rule."""
import numpy as np
import pytest

import tvl1_oracle as tv


def _texture(h, w, seed, margin=24):
    """Smooth random texture (repeated box blurs of white noise), float 0..255."""
    rng = np.random.default_rng(seed)
    t = rng.random((h + 2 * margin, w + 2 * margin))
    for _ in range(6):
        t = (t + np.roll(t, 1, 0) + np.roll(t, -1, 0) + np.roll(t, 1, 1) + np.roll(t, -1, 1)) / 5.0
    return (t - t.min()) / (t.max() - t.min()) * 255.0


def _shifted_pair(h, w, dx, dy, seed=0, margin=24):
    """frame1(x, y) = frame0(x - dx, y - dy): the content moves by (dx, dy), so the flow from frame0 to frame1 is (dx, dy)."""
    t = _texture(h, w, seed, margin)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)

    def sample(ox, oy):
        x, y = xs + margin - ox, ys + margin - oy
        x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
        fx, fy = x - x0, y - y0
        return (t[y0, x0] * (1 - fx) * (1 - fy) + t[y0, x0 + 1] * fx * (1 - fy) + t[y0 + 1, x0] * (1 - fx) * fy + t[y0 + 1, x0 + 1] * fx * fy)
    return np.rint(sample(0, 0)).astype(np.uint8), np.rint(sample(dx, dy)).astype(np.uint8)


@pytest.mark.parametrize("dx,dy", [(3.0, -1.5), (-0.75, 2.25), (0.0, 0.0)])
def test_translation_is_recovered(dx, dy):
    f0, f1 = _shifted_pair(96, 128, dx, dy)
    u1, u2, counts = tv.tvl1_flow(f0, f1)
    inner = (slice(16, -16), slice(16, -16))
    assert abs(np.median(u1[inner]) - dx) < 0.1 and abs(np.median(u2[inner]) - dy) < 0.1
    assert np.abs(u1[inner] - dx).mean() < 0.2 and np.abs(u2[inner] - dy).mean() < 0.2
    assert len(counts) == len(tv.pyramid_sizes(96, 128)) and all(1 <= n <= tv.ITERATIONS for lvl in counts for n in lvl)


def test_identical_frames_give_exactly_zero_flow():
    f0, _ = _shifted_pair(64, 80, 0, 0, seed=3)
    u1, u2, counts = tv.tvl1_flow(f0, f0)
    assert (u1 == 0).all() and (u2 == 0).all()
    assert all(n == 1 for lvl in counts for n in lvl)              # the first update is zero: every warp stops at once


def test_divergence_is_minus_the_adjoint_of_the_forward_gradient():
    rng = np.random.default_rng(1)
    u, p1, p2 = (rng.standard_normal((17, 23)).astype(np.float32) for _ in range(3))
    ux, uy = tv.forward_gradient(u)
    lhs = float((ux.astype(np.float64) * p1 + uy.astype(np.float64) * p2).sum())
    rhs = -float((u.astype(np.float64) * tv.divergence(p1 * (np.arange(23) < 22), p2 * (np.arange(17) < 16)[:, None])).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_pyramid_and_quantisation_rules():
    assert tv.pyramid_sizes(256, 340) == [(256, 340), (205, 272), (164, 218), (131, 174), (105, 139)]
    assert tv.pyramid_sizes(20, 20) == [(20, 20), (16, 16)]
    q = tv.flow_to_image(np.array([-25.0, -20.0, -10.0, 0.0, 0.0784, 10.0, 20.0, 99.0], dtype=np.float32))
    assert q.tolist() == [0, 0, 64, 128, 128, 191, 255, 255]          # [-20, 20] -> [0, 255], rounded, clamped
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert (tv.resize_bilinear(img, 3, 4) == img).all() and (tv.warp_bilinear(img, np.zeros_like(img), np.zeros_like(img)) == img).all()
    ident = tv.warp_homography(img, np.eye(3))
    assert (ident == img).all()
    shifted = tv.warp_homography(img, [[1, 0, 1], [0, 1, 0], [0, 0, 1]])        # content moves one pixel to the right
    assert (shifted[:, 1:] == img[:, :-1]).all()
