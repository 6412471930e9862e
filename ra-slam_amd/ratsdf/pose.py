"""Host-side SE3 helpers in float32, mirroring ``SE3<float>`` (utils/cuda/lie_group.cuh:8-45).

The reference builds the quaternion from a rotation matrix on the host with Eigen
(lie_group.cuh:15-19) and composes extrinsics with ``SE3::operator*`` (lie_group.cuh:38-40,
used at modules/tsdf_module.cc:28,33).  Eigen is not vendored in the reference tree; the formulas
below restate Eigen 3.3.7's ``Quaternion(Matrix3)``, quaternion product, ``_transformVector`` and
``inverse`` in single precision.  A pose is the 7-tuple (qx, qy, qz, qw, tx, ty, tz).
"""
import numpy as np

f32 = np.float32


def identity_pose():
    return (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0)


def pose_from_matrix(m):
    """Eigen quaternionbase_assign_impl<Matrix3f> (trace method) on the top-left 3x3 + top-right 3x1."""
    m = np.asarray(m, dtype=np.float32)
    r = m[:3, :3]
    t = m[:3, 3] if m.shape[1] > 3 else np.zeros(3, dtype=np.float32)
    q = np.zeros(4, dtype=np.float32)  # x, y, z, w
    tr = f32(r[0, 0] + r[1, 1]) + r[2, 2]
    if tr > f32(0):
        s = np.sqrt(f32(tr + f32(1.0)))
        q[3] = f32(0.5) * s
        s = f32(0.5) / s
        q[0] = f32(r[2, 1] - r[1, 2]) * s
        q[1] = f32(r[0, 2] - r[2, 0]) * s
        q[2] = f32(r[1, 0] - r[0, 1]) * s
    else:
        i = 0
        if r[1, 1] > r[0, 0]:
            i = 1
        if r[2, 2] > r[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        s = np.sqrt(f32(f32(f32(r[i, i] - r[j, j]) - r[k, k]) + f32(1.0)))
        q[i] = f32(0.5) * s
        s = f32(0.5) / s
        q[3] = f32(r[k, j] - r[j, k]) * s
        q[j] = f32(r[j, i] + r[i, j]) * s
        q[k] = f32(r[k, i] + r[i, k]) * s
    return tuple(float(v) for v in (q[0], q[1], q[2], q[3], t[0], t[1], t[2]))


def _cross(a, b):
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]),
                     f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=np.float32)


def _qrot(q, v):
    qv = q[:3]
    uv = _cross(qv, v)
    uv = uv + uv
    c = _cross(qv, uv)
    return np.array([f32(f32(v[i] + f32(q[3] * uv[i])) + c[i]) for i in range(3)],
                    dtype=np.float32)


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        f32(f32(f32(aw * bx) + f32(ax * bw)) + f32(ay * bz)) - f32(az * by),
        f32(f32(f32(aw * by) + f32(ay * bw)) + f32(az * bx)) - f32(ax * bz),
        f32(f32(f32(aw * bz) + f32(az * bw)) + f32(ax * by)) - f32(ay * bx),
        f32(f32(f32(aw * bw) - f32(ax * bx)) - f32(ay * by)) - f32(az * bz),
    ], dtype=np.float32)


def _split(p):
    a = np.asarray(p, dtype=np.float32)
    return a[:4].copy(), a[4:].copy()


def compose(a, b):
    """SE3::operator*: (Ra*Rb, Ra*tb + ta)."""
    qa, ta = _split(a)
    qb, tb = _split(b)
    q = _qmul(qa, qb)
    t = _qrot(qa, tb) + ta
    return tuple(float(v) for v in (*q, *t))


def invert(p):
    """SE3::Inverse: (R^-1, R^-1 * (-t)), Eigen inverse = conjugate / squaredNorm."""
    q, t = _split(p)
    n2 = f32(f32(q[0] * q[0]) + f32(q[1] * q[1])) + f32(f32(q[2] * q[2]) + f32(q[3] * q[3]))
    qi = np.array([-q[0] / n2, -q[1] / n2, -q[2] / n2, q[3] / n2], dtype=np.float32)
    ti = _qrot(qi, -t)
    return tuple(float(v) for v in (*qi, *ti))


def apply(p, v):
    """SE3::Apply: R*v + t."""
    q, t = _split(p)
    return _qrot(q, np.asarray(v, dtype=np.float32)) + t
