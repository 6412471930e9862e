"""Seeded synthetic RGB-D + high/low-touch streams for parity tests and bench.py.

Analytic depth (float32 metres, z in the camera frame) of three scenes seen by a pin-hole camera:
  wall    fronto-parallel plane z = 2 m, camera translating slowly in x
  room    4 x 2.5 x 3 m box seen from inside, camera on a 0.5 m circle, 1 deg/frame yaw
  sphere  radius 1.5 m seen from inside, 5 mm/frame translation
Camera presets reuse the reference's config values (configs/scannet_scene0.yaml:12-15,
configs/TUM_RGBD_rgbd_1.yaml:11-14, 2x configs/zed_native_l515.yaml:30-33).
"""
import math

import numpy as np

from .pose import pose_from_matrix

CAMERAS = {
    # name: (width, height, fx, fy, cx, cy)
    "scannet": (640, 480, 571.623718, 571.623718, 319.5, 239.5),
    "tum": (640, 480, 517.306408, 516.469215, 318.643040, 255.313989),
    "l515_720p": (1280, 720, 913.7234, 913.54254, 644.2084, 375.5897),
}


def camera(name, scale=1.0):
    """(W, H, (fx, fy, cx, cy)); scale < 1 shrinks the image for quick parity cases."""
    w, h, fx, fy, cx, cy = CAMERAS[name]
    if scale != 1.0:
        w, h = int(round(w * scale)), int(round(h * scale))
        fx, fy = fx * scale, fy * scale
        cx, cy = (cx + 0.5) * scale - 0.5, (cy + 0.5) * scale - 0.5
    return w, h, (fx, fy, cx, cy)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def camera_pose(scene, frame):
    """world_T_cam as (R 3x3, c 3) in float64 for frame index `frame`."""
    if scene == "wall":
        return np.eye(3), np.array([0.002 * frame, 0.0, 0.0])
    if scene == "room":
        a = math.radians(1.0 * frame)
        # on a 0.5 m circle around the room centre, looking outward along the radius with a slight
        # pitch so walls and floor/ceiling are both seen
        c = 0.5 * np.array([math.sin(a), 0.0, math.cos(a)])
        return _rot_y(a) @ _rot_x(math.radians(8.0)), c
    if scene == "sphere":
        return _rot_y(math.radians(0.25 * frame)), np.array([0.005 * frame, 0.0, 0.0])
    raise ValueError(scene)


def _depth(scene, R, c, w, h, intr):
    fx, fy, cx, cy = intr
    xs = (np.arange(w, dtype=np.float64) - cx) / fx
    ys = (np.arange(h, dtype=np.float64) - cy) / fy
    dc = np.stack(np.broadcast_arrays(xs[None, :], ys[:, None], np.ones((h, w))), axis=-1)
    d = dc @ R.T  # world-frame ray directions; ray = c + t * d, camera-frame depth == t
    if scene == "wall":
        t = (2.0 - c[2]) / d[..., 2]
    elif scene == "room":
        lo = np.array([-2.0, -1.25, -1.5])
        hi = np.array([2.0, 1.25, 1.5])
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (lo - c) / d
            t2 = (hi - c) / d
        t = np.where(d > 0, t2, np.where(d < 0, t1, np.inf)).min(axis=-1)
    elif scene == "sphere":
        rad = 1.5
        a = (d * d).sum(-1)
        b = 2.0 * (d @ c)
        cc = float(c @ c) - rad * rad
        t = (-b + np.sqrt(b * b - 4 * a * cc)) / (2 * a)
    else:
        raise ValueError(scene)
    t = np.where(np.isfinite(t) & (t > 0), t, 0.0)
    return t.astype(np.float32)


def semantics(w, h):
    """ht = clip(0.5 + 0.4 sin(x/37) cos(y/29), 0.01, 0.99), lt = 1 - ht (float32)."""
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    ht = np.clip(0.5 + 0.4 * np.sin(x / 37.0) * np.cos(y / 29.0), 0.01, 0.99).astype(np.float32)
    lt = (np.float32(1.0) - ht).astype(np.float32)
    return ht, lt


def frame(scene, index, cam="scannet", scale=1.0, noise=False, holes=False, semantic=True):
    """One synthetic frame: dict(rgb, depth, ht, lt, pose, intrinsics, width, height)."""
    w, h, intr = camera(cam, scale)
    R, c = camera_pose(scene, index)
    depth = _depth(scene, R, c, w, h, intr)
    if noise:
        rng = np.random.default_rng(20240607 + index)
        depth = (depth + rng.normal(0.0, 0.001, size=depth.shape)).astype(np.float32)
    if holes:
        rng = np.random.default_rng(7 + index)
        depth = np.where(rng.random(depth.shape) < 0.01, np.float32(0), depth).astype(np.float32)
    rgb = np.random.default_rng(1 + index).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    ht, lt = semantics(w, h) if semantic else (None, None)
    m = np.eye(4)
    m[:3, :3] = R.T
    m[:3, 3] = -R.T @ c
    pose = pose_from_matrix(m.astype(np.float32))  # cam_T_world
    return dict(rgb=rgb, depth=depth, ht=ht, lt=lt, pose=pose,
                intrinsics=tuple(float(np.float32(v)) for v in intr), width=w, height=h)


def stream(scene, n, **kw):
    return [frame(scene, i, **kw) for i in range(n)]
