// stand-in declarations of the reference's utils/gl/image.h:109-117 (type check only), with the
// host-upload method the binding's default RATSDF_GL_UPLOAD expects a maintainer to add
#pragma once
#include <cstring>
#include <vector>
class GLImage8UC4 {
 public:
  void BindImage(int height, int width, const void* = nullptr) { h_ = height; w_ = width; pixels_.assign((size_t)h_ * w_ * 4, 0); }
  int GetHeight() { return h_; }
  int GetWidth() { return w_; }
  void LoadHost(const void* rgba) { std::memcpy(pixels_.data(), rgba, pixels_.size()); }
  std::vector<unsigned char> pixels_;

 private:
  int h_ = 0, w_ = 0;
};
