"""GPU: camera-motion estimation of the warped flow (csrc/vq_flow.hip: corner kernels, RANSAC kernel, host selection / refit
in the library) against oracle/warp_oracle.py, and the whole warped-flow pipeline on synthetic camera motion.

PARITY UNPINNED with respect to the reference (third-party extract_warp_gpu, build_wof_clips.py:70-73, absent; no frames, no
flow images).  Bars: corners -- the same list, bit for bit (integer pixel positions, fp32 strengths computed in the same
order); RANSAC -- the same winning hypothesis, inlier count and mask (same counter-hash samples, fp64 scoring), matrix to
1e-9 relative (Gaussian elimination here, LAPACK there)."""
import numpy as np
import pytest

import tvl1_oracle as tv
import warp_oracle as wo
from test_warp_oracle import _checkerboard, analytic_pair, synthetic_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def flow_mod(gpu):
    from video_query_algorithms_amd.tsn import flow
    return flow


def test_corners_equal_the_oracle_bit_for_bit(flow_mod):
    rng = np.random.default_rng(8)
    frames = np.stack([_checkerboard(48, 64), rng.integers(0, 256, (48, 64), dtype=np.uint8), analytic_pair(48, 64, np.eye(3), seed=4)[0],
                       np.full((48, 64), 77, np.uint8)])
    m = flow_mod.Tvl1Flow(4, 48, 64)
    for cap, q, md in ((1000, 0.001, 3.0), (25, 0.01, 5.0), (1000, 0.05, 0.0)):
        corners, counts = m.good_features(frames, cap, q, md)
        for i in range(len(frames)):
            want = wo.good_features(frames[i], cap, q, md)
            assert counts[i] == len(want), (i, cap, q, md, counts[i], len(want))
            assert (corners[i, :counts[i]] == want).all(), (i, cap, q, md)
    assert counts[3] == 0
    # a frame's corners do not depend on its batch
    solo, c1 = m.good_features(frames[1:2])
    both, c2 = m.good_features(frames)
    assert c1[0] == c2[1] and (solo[0] == both[1]).all()
    m.close()


def test_ransac_winner_inliers_and_matrix_equal_the_oracle(flow_mod):
    H = [np.array([[1.01, 0.02, 3.0], [-0.015, 0.99, -2.0], [2e-5, -1e-5, 1.0]]), np.array([[0.98, -0.03, -4.0], [0.03, 0.98, 6.0], [0, 0, 1.0]]),
         np.eye(3)]
    sets = [synthetic_matches(H[0], 220, 90, seed=5), synthetic_matches(H[1], 120, 80, seed=6, noise=0.1), synthetic_matches(H[2], 3, 0, seed=7)]
    mp = 320
    src, dst = np.zeros((3, mp, 2), np.float32), np.zeros((3, mp, 2), np.float32)
    counts = np.array([len(s[0]) for s in sets], np.int32)
    for i, (s, d, _) in enumerate(sets):
        src[i, :len(s)], dst[i, :len(d)] = s, d
    m = flow_mod.Tvl1Flow(4, 32, 32)
    for refit in (False, True):
        r = m.ransac_homography(src, dst, counts, 1.0, 96, seed=11, refit=refit)
        for i in range(3):
            G, cnt, winner, mask = wo.ransac_homography(sets[i][0], sets[i][1], 1.0, 96, seed=11, pair=i, refit=refit)
            assert (r["winner"][i], r["inliers"][i]) == (winner, cnt), (i, refit)
            assert (r["mask"][i, :counts[i]] == mask).all() and r["mask"][i, counts[i]:].sum() == 0
            assert np.abs(r["H"][i] - G).max() <= 1e-9 * np.abs(G).max(), (i, refit, np.abs(r["H"][i] - G).max())
    assert r["winner"][2] == -1 and (r["H"][2] == np.eye(3)).all()              # three matches: nothing to draw
    assert r["inliers"][0] >= 220 and r["inliers"][1] >= 100
    # more hypotheses than threads, and another seed: still the oracle's answer
    r2 = m.ransac_homography(src[:2], dst[:2], counts[:2], 1.0, 700, seed=99)
    for i in range(2):
        G, cnt, winner, _ = wo.ransac_homography(sets[i][0], sets[i][1], 1.0, 700, seed=99, pair=i)
        assert (r2["winner"][i], r2["inliers"][i]) == (winner, cnt)
    m.close()


def _corner_error(G, H, w, h):
    c = np.array([[0, 0, 1], [w, 0, 1], [0, h, 1], [w, h, 1.0]])
    a, b = c @ G.T, c @ H.T
    return np.abs(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:]).max()


def test_warped_flow_removes_the_camera_motion_and_keeps_the_foreground(flow_mod):
    """A textured background under a known homography (pan + slight rotation and zoom) with a foreground square that moves
    on its own: the estimated camera motion is the background's, the second-pass flow is ~0 on the background and the
    square's motion relative to the camera on the square."""
    h, w = 128, 160
    th = 0.01
    H = np.array([[1.01 * np.cos(th), -np.sin(th), 2.5], [np.sin(th), 1.01 * np.cos(th), -1.5], [0, 0, 1.0]])
    f0, f1 = analytic_pair(h, w, H, seed=12, blob=(60, 50, 28, 6.0, 4.0))
    e0, e1 = analytic_pair(h, w, np.eye(3), seed=13)
    m = flow_mod.Tvl1Flow(2, h, w)
    r = m.warped(np.stack([f0, e0]), np.stack([f1, e1]), seed=3, fields=True)
    assert r["matches"][0] > 50 and r["inliers"][0] > 0.6 * r["matches"][0]
    assert _corner_error(r["H"][0], H, w, h) < 0.5
    assert _corner_error(r["H"][1], np.eye(3), w, h) < 0.1                      # a static camera stays static
    plain = m.flow(np.stack([f0, e0]), np.stack([f1, e1]), images=False)
    bg = np.ones((h, w), bool)
    bg[40:95, 50:105] = False
    bg[:12], bg[-12:], bg[:, :12], bg[:, -12:] = False, False, False, False
    assert np.median(np.hypot(plain["u1"][0], plain["u2"][0])[bg]) > 1.5        # the camera motion is in the plain flow ...
    assert np.median(np.hypot(r["u1"][0], r["u2"][0])[bg]) < 0.15               # ... and gone from the warped flow
    sq = (slice(58, 74), slice(68, 84))                                         # inside the square in both frames
    cam = H @ np.array([74.0, 64.0, 1.0])
    rel = np.array([74.0 + 6.0, 64.0 + 4.0]) - cam[:2] / cam[2]                 # the square's motion seen from the moved camera
    assert abs(np.median(r["u1"][0][sq]) - rel[0]) < 0.6 and abs(np.median(r["u2"][0][sq]) - rel[1]) < 0.6
    # the 8-bit images are the -b 20 quantisation of those fields
    assert (r["flow_x"][0] == tv.flow_to_image(r["u1"][0])).all() and (r["flow_y"][1] == tv.flow_to_image(r["u2"][1])).all()
    # the oracle's estimate from the device's first-pass flow: same corners, same samples -> same matrix
    G, matches, inliers = wo.camera_motion(f0, plain["u1"][0], plain["u2"][0], seed=3, pair=0)
    assert matches == r["matches"][0] and inliers == r["inliers"][0]
    assert np.abs(G - r["H"][0]).max() <= 1e-8 * np.abs(G).max()
    m.close()


def test_warped_consecutive_frames_of_a_panning_clip(flow_mod):
    """extract_warp_gpu -s 1 on a short clip: n + 1 frames -> n image pairs; a steady pan leaves ~mid-grey warped flow."""
    h, w = 96, 128
    frames = []
    for t in range(5):
        H = np.array([[1, 0, 2.0 * t], [0, 1, -1.0 * t], [0, 0, 1.0]])
        frames.append(analytic_pair(h, w, H, seed=21)[1])
    frames = np.stack(frames)
    m = flow_mod.Tvl1Flow(3, h, w)                                               # 4 pairs through batches of 3
    fx, fy = m.warped_consecutive(frames, seed=1)
    px, py = m.consecutive(frames)
    assert fx.shape == fy.shape == (4, h, w) and fx.dtype == np.uint8
    inner = (slice(None), slice(16, -16), slice(16, -16))
    assert abs(np.median(px[inner]) - tv.flow_to_image(np.float32([2.0]))[0]) <= 1 and abs(np.median(py[inner]) - tv.flow_to_image(np.float32([-1.0]))[0]) <= 1
    assert abs(int(np.median(fx[inner])) - 128) <= 1 and abs(int(np.median(fy[inner])) - 128) <= 1
    m.close()


def test_one_call_warped_flow_equals_the_steps(flow_mod):
    """vq_flow_warped (frames uploaded once, first-pass fields kept on the device, corners moved by a kernel) against the same
    sequence made of the public calls (Tvl1Flow.warped_steps: flow -> good_features -> numpy gather -> ransac_homography -> flow):
    identical corner / inlier counts, homographies equal to rounding (the inverse handed to the second pass is computed by cofactors
    here and by LAPACK there), flow images equal up to one grey level on a handful of pixels, fields to 1e-3 px."""
    pairs = [analytic_pair(96, 128, np.array([[1.002, 0.004, 2.5], [-0.003, 0.999, -1.5], [0, 0, 1.0]]), seed=7 + k) for k in range(3)]
    f0, f1 = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    m = flow_mod.Tvl1Flow(4, 96, 128)
    a = m.warped(f0, f1, seed=3, images=True, fields=True)
    b = m.warped_steps(f0, f1, seed=3, images=True, fields=True)
    assert (a["matches"] == b["matches"]).all() and (a["inliers"] == b["inliers"]).all() and a["matches"].min() > 50
    assert np.abs(a["H"] - b["H"]).max() <= 1e-9 * np.abs(b["H"]).max()
    assert np.abs(a["u1"] - b["u1"]).max() <= 1e-3 and np.abs(a["u2"] - b["u2"]).max() <= 1e-3
    d = np.abs(a["flow_x"].astype(int) - b["flow_x"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.001
    only = m.warped(f0[:1], f1[:1], seed=3, images=True, fields=False)            # a pair's result does not depend on its batch
    assert (only["flow_x"][0] == a["flow_x"][0]).all() and "u1" not in only
    m.close()
