"""CPU restatement (numpy) of the image handling around the segmentation network
(segmentation/inference.cc:9-99) -- TEST INFRASTRUCTURE, never imported by the product.

  resize_u8_linear   cv::resize(src CV_8UC3, dst, size) with the default INTER_LINEAR as OpenCV
                     evaluates it for 8-bit images (imgproc/resize.cpp, OpenCV 3.x/4.x, pinned by the
                     reference to "OpenCV 4" via find_package): half-pixel sample positions, 11-bit
                     fixed-point coefficients (cvRound(c * 2048)), horizontal pass into int, vertical
                     pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2.
  resize_f32_linear  the same resize for CV_32FC1 (float coefficients, horizontal then vertical)
  infer_one          inference.cc:63-99 around a caller-supplied network function

OpenCV is an un-vendored dependency of the reference and absent here: parity of the two resizes with
the real library is UNPINNED; they follow the published algorithm and pin the product's device
implementation (ra-slam_amd/ratsdf/segmentation.py) against drift.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis_table(src, dst):
    """(index of the first tap, clamped; float weight of the second tap) per output position."""
    scale = np.float64(src) / np.float64(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0
    s[hi] = src - 1
    return s, f


def _fix(c):
    return np.rint(c.astype(np.float32) * np.float32(COEF_SCALE)).astype(np.int64)   # cvRound -> short


def resize_u8_linear(img, out_h, out_w):
    img = np.asarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    sx, fx = _axis_table(w, out_w)
    sy, fy = _axis_table(h, out_h)
    a0, a1 = _fix(np.float32(1) - fx), _fix(fx)
    b0, b1 = _fix(np.float32(1) - fy), _fix(fy)
    src = img.astype(np.int64)
    x1 = np.minimum(sx + 1, w - 1)
    y1 = np.minimum(sy + 1, h - 1)
    shp = (1, out_w) + (1,) * (img.ndim - 2)
    hor = src[:, sx] * a0.reshape(shp) + src[:, x1] * a1.reshape(shp)      # rows of the source, int
    r0, r1 = hor[sy], hor[y1]
    shp = (out_h, 1) + (1,) * (img.ndim - 2)
    out = (((b0.reshape(shp) * (r0 >> 4)) >> 16) + ((b1.reshape(shp) * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_f32_linear(img, out_h, out_w):
    img = np.asarray(img, dtype=np.float32)
    h, w = img.shape
    sx, fx = _axis_table(w, out_w)
    sy, fy = _axis_table(h, out_h)
    x1 = np.minimum(sx + 1, w - 1)
    y1 = np.minimum(sy + 1, h - 1)
    hor = img[:, sx] * (np.float32(1) - fx)[None, :] + img[:, x1] * fx[None, :]
    return (hor[sy] * (np.float32(1) - fy)[:, None] + hor[y1] * fy[:, None]).astype(np.float32)


def infer_one(rgb, width, height, network, ret_uint8=False):
    """network: callable (1 x 3 x H' x W' float32 array in [0, 1]) -> 2 x H' x W' float32 array."""
    whole_w = (width // 32 + 1) * 32            # inference.cc:47
    whole_h = (height // 32 + 1) * 32           # inference.cc:48
    x = resize_u8_linear(rgb, whole_h, whole_w)                     # :74 (8-bit result)
    x = x.astype(np.float32) * np.float32(1.0 / 255.0)              # :13
    y = network(np.transpose(x, (2, 0, 1))[None])                   # :14-18, :83-84
    if ret_uint8:                                                   # :33-41, network resolution
        return [np.clip(y[k] * np.float32(255), 0, 255).astype(np.uint8) for k in range(2)]
    return [resize_f32_linear(y[k], height, width) for k in range(2)]   # :23-31
