// stand-in declarations of the reference's utils/cuda/camera.cuh:13-68 (type check only)
#pragma once
template <typename T>
struct CameraIntrinsics {
  const T fx, fy, cx, cy;
  CameraIntrinsics(const T& fx_, const T& fy_, const T& cx_, const T& cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
  CameraIntrinsics<T> Inverse() const { return CameraIntrinsics<T>(1 / fx, 1 / fy, -cx / fx, -cy / fy); }
};
class CameraParams {
 public:
  CameraIntrinsics<float> intrinsics;
  CameraIntrinsics<float> intrinsics_inv;
  int img_h;
  int img_w;
  CameraParams(const CameraIntrinsics<float>& k, int h, int w) : intrinsics(k), intrinsics_inv(k.Inverse()), img_h(h), img_w(w) {}
};
