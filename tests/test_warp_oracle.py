"""CPU: oracle/warp_oracle.py -- camera-motion estimation of the reference's warped flow, flow-match branch (PARITY UNPINNED:
extract_warp_gpu, build_wof_clips.py:70-73, is a third-party binary absent from the tree, and no frame or flow image ships).
Pinned here: known answers of the restated OpenCV rules themselves."""
import numpy as np
import pytest

import warp_oracle as wo


def _checkerboard(h=48, w=64, cell=(12, 16), seed=0):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.int64)
    for by in range(0, h, cell[0]):
        for bx in range(0, w, cell[1]):
            if ((by // cell[0]) + (bx // cell[1])) % 2 == 0:
                img[by:by + cell[0], bx:bx + cell[1]] = 200
    return (img + rng.integers(0, 8, img.shape)).clip(0, 255).astype(np.uint8)


def analytic_pair(h, w, H, seed=0, blob=None):
    """A smooth texture f evaluated analytically: frame0(x) = f(x), frame1(x) = f(H^-1 x), i.e. content at x moves to H x.
    blob = (x0, y0, size, dx, dy): a square of another texture that moves by (dx, dy) instead (foreground)."""
    rng = np.random.default_rng(seed)
    k = rng.uniform(0.05, 0.45, (24, 2)) * rng.choice([-1, 1], (24, 2))
    ph = rng.uniform(0, 2 * np.pi, 24)
    am = rng.uniform(0.5, 1.0, 24)

    def f(x, y, flip=1.0):
        v = sum(a * np.sin(flip * (kx * x + ky * y) + p) for (kx, ky), p, a in zip(k, ph, am))
        return 127.5 + 110.0 * v / np.abs(am).sum() * 2.0

    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    Hi = np.linalg.inv(H)
    den = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    bx, by = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / den, (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / den
    f0, f1 = f(xs, ys), f(bx, by)
    if blob is not None:
        x0, y0, size, dx, dy = blob
        g0 = f(1.2 * (xs - x0) + 40.0, 0.9 * (ys - y0) - 17.0, -1.0)
        m0 = (xs >= x0) & (xs < x0 + size) & (ys >= y0) & (ys < y0 + size)
        m1 = (xs - dx >= x0) & (xs - dx < x0 + size) & (ys - dy >= y0) & (ys - dy < y0 + size)
        f0 = np.where(m0, g0, f0)
        f1 = np.where(m1, f(1.2 * (xs - dx - x0) + 40.0, 0.9 * (ys - dy - y0) - 17.0, -1.0), f1)
    return np.rint(f0.clip(0, 255)).astype(np.uint8), np.rint(f1.clip(0, 255)).astype(np.uint8)


def synthetic_matches(H, n_in, n_out, seed, w=340, h=256, noise=0.0):
    rng = np.random.default_rng(seed)
    src = np.stack([rng.uniform(0, w, n_in + n_out), rng.uniform(0, h, n_in + n_out)], 1)
    p = np.c_[src, np.ones(len(src))] @ H.T
    dst = p[:, :2] / p[:, 2:]
    dst[:n_in] += rng.normal(0, noise, (n_in, 2))
    dst[n_in:] += rng.uniform(5, 40, (n_out, 2)) * rng.choice([-1, 1], (n_out, 2))
    order = rng.permutation(len(src))
    return src[order].astype(np.float32), dst[order].astype(np.float32), (order < n_in)


def test_checkerboard_junctions_are_the_corners():
    img = _checkerboard()
    c = wo.good_features(img, 50, 0.01, 3.0)
    junctions = {(x, y) for x in (16, 32, 48) for y in (12, 24, 36)}
    assert len(c) == 9
    for x, y in c:
        assert min(abs(x - jx) + abs(y - jy) for jx, jy in junctions) <= 2
    s = wo.corner_strength(img)
    vals = [s[int(y), int(x)] for x, y in c]
    assert vals == sorted(vals, reverse=True)                                    # strongest first
    assert np.abs(wo.corner_strength(img.T).T - s).max() <= 1e-5 * s.max()       # x and y play the same role
    # a flat image has no corners at all; the quality floor is relative to the strongest response
    assert len(wo.good_features(np.full((32, 40), 90, np.uint8))) == 0


def test_min_distance_and_cap():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    for md in (0.0, 3.0, 6.0):
        c = wo.good_features(img, 1000, 0.001, md)
        d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)) + np.eye(len(c)) * 1e9
        assert len(c) > 10 and (md < 1 or d.min() >= md)
    many = wo.good_features(img, 1000, 0.001, 3.0)
    assert (wo.good_features(img, 7, 0.001, 3.0) == many[:7]).all()              # the cap cuts the same ordered list


def test_sample_draws_are_distinct_and_a_pure_function_of_their_counters():
    seen = set()
    for j in range(200):
        idx = wo.draw_sample(7, 3, j, 9)
        assert len(set(idx)) == 4 and all(0 <= i < 9 for i in idx) and idx == wo.draw_sample(7, 3, j, 9)
        seen.add(tuple(idx))
    assert len(seen) > 150 and wo.draw_sample(7, 3, 0, 9) != wo.draw_sample(7, 4, 0, 9)
    assert wo.mix32(0) == 0 and wo.mix32(1) != 1


def test_known_homography_is_recovered_among_outliers():
    H = np.array([[1.01, 0.02, 3.0], [-0.015, 0.99, -2.0], [2e-5, -1e-5, 1.0]])
    src, dst, inl = synthetic_matches(H, 220, 90, seed=5)
    G, count, winner, mask = wo.ransac_homography(src, dst, 1.0, 128, seed=11, pair=2)
    assert winner >= 0 and count >= 220 and (mask[inl] == 1).all() and mask[~inl].sum() <= count - 220
    corners = np.array([[0, 0, 1], [340, 0, 1], [0, 256, 1], [340, 256, 1.0]])
    a, b = corners @ G.T, corners @ H.T
    assert np.abs(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:]).max() < 1e-3       # float32 matches: ~1e-5 px each
    # the four-point solver is exact on exact data; fewer than four matches give the identity
    assert np.abs(wo.homography_4pt(src[inl][:4].astype(np.float64), dst[inl][:4].astype(np.float64)) - H).max() < 1e-3
    E, c0, w0, _ = wo.ransac_homography(src[:3], dst[:3])
    assert (E == np.eye(3)).all() and c0 == 0 and w0 == -1


def test_camera_motion_guards():
    """dense_flow keeps the identity unless there are > 50 matches and > 25 inliers."""
    img = _checkerboard()                                     # 9 corners only
    H, matches, inliers = wo.camera_motion(img, np.zeros(img.shape, np.float32), np.zeros(img.shape, np.float32))
    assert (H == np.eye(3)).all() and matches <= 50 and inliers == 0
    f0, _ = analytic_pair(64, 80, np.eye(3), seed=2)
    u = np.full(f0.shape, 1.5, np.float32)
    H, matches, inliers = wo.camera_motion(f0, u, -u, hypotheses=64)
    assert matches > 50 and inliers == matches                # a pure translation field: every match agrees
    assert np.abs(H - np.array([[1, 0, 1.5], [0, 1, -1.5], [0, 0, 1]])).max() < 1e-6
