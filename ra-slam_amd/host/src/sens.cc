// sens.cc -- ScanNet .sens container + the reference's reader on top of it
// (third_party/scannet/sensorData.hpp:491-540, RGBDFrame.h:286-297, calibrationData.hpp:41-44;
// utils/offline_data_provider/scannet_sens_reader.cc:8-82).
#include <zlib.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "ratsdf/dataset.hpp"

namespace ratsdf {

namespace {
enum { kColorRaw = 0, kColorPng = 1, kColorJpeg = 2 };     // COMPRESSION_TYPE_COLOR, include.hpp:259-264
enum { kDepthRaw = 0, kDepthZlib = 1, kDepthOcci = 2 };    // COMPRESSION_TYPE_DEPTH, include.hpp:265-270

int cv_round(float v) { return (int)std::nearbyintf(v); }  // cvRound: half to even
}  // namespace

// cv::resize, INTER_LINEAR, CV_8UC3 (imgproc/resize.cpp): the same arithmetic as
// oracle/segmentation_oracle.py:resize_u8_linear and ratsdf/segmentation.py:resize_u8_linear
RgbImage resize_rgb_linear(const RgbImage& src, int out_w, int out_h) {
  RgbImage dst;
  dst.width = out_w;
  dst.height = out_h;
  dst.data.resize((size_t)out_w * out_h * 3);
  struct Tap {
    int s0, s1, a0, a1;
  };
  auto table = [](int n_src, int n_dst) {
    std::vector<Tap> t((size_t)n_dst);
    const double scale = (double)n_src / n_dst;
    for (int d = 0; d < n_dst; ++d) {
      float f = (float)((d + 0.5) * scale - 0.5);
      int s = (int)std::floor(f);
      f -= (float)s;
      if (s < 0) {
        f = 0;
        s = 0;
      }
      if (s >= n_src - 1) {
        f = 0;
        s = n_src - 1;
      }
      t[(size_t)d] = Tap{s, std::min(s + 1, n_src - 1), cv_round((1.f - f) * 2048.f), cv_round(f * 2048.f)};
    }
    return t;
  };
  const std::vector<Tap> tx = table(src.width, out_w), ty = table(src.height, out_h);
  std::vector<int> row0((size_t)out_w * 3), row1((size_t)out_w * 3);
  auto hresize = [&](int sy, std::vector<int>& out) {
    const uint8_t* S = &src.data[(size_t)sy * src.width * 3];
    for (int x = 0; x < out_w; ++x)
      for (int c = 0; c < 3; ++c)
        out[(size_t)x * 3 + c] = S[tx[(size_t)x].s0 * 3 + c] * tx[(size_t)x].a0 + S[tx[(size_t)x].s1 * 3 + c] * tx[(size_t)x].a1;
  };
  int have0 = -1, have1 = -1;
  for (int y = 0; y < out_h; ++y) {
    const Tap& t = ty[(size_t)y];
    if (have0 != t.s0) {
      if (have1 == t.s0) std::swap(row0, row1), std::swap(have0, have1);
      else hresize(t.s0, row0), have0 = t.s0;
    }
    if (have1 != t.s1) hresize(t.s1, row1), have1 = t.s1;
    uint8_t* D = &dst.data[(size_t)y * out_w * 3];
    for (int i = 0; i < out_w * 3; ++i) {
      const int v = (((t.a0 * (row0[(size_t)i] >> 4)) >> 16) + ((t.a1 * (row1[(size_t)i] >> 4)) >> 16) + 2) >> 2;
      D[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  return dst;
}

scannet_sens_reader::scannet_sens_reader(const std::string& sens_filepath) : path_(sens_filepath) {
  std::ifstream in(sens_filepath, std::ios::binary | std::ios::ate);
  if (!in) throw std::runtime_error("could not open file " + sens_filepath);  // sensorData.hpp:494-496
  const std::streamoff len = in.tellg();
  in.seekg(0);
  file_.resize((size_t)len);
  in.read(reinterpret_cast<char*>(file_.data()), len);
  if (!in) throw std::runtime_error(sens_filepath + ": read error");
  size_t p = 0;
  auto need = [&](uint64_t n) {  // p <= size always; lengths come from 64-bit fields of the file
    if (n > (uint64_t)(file_.size() - p)) throw std::runtime_error(path_ + ": truncated .sens stream");
  };
  auto rd = [&](void* dst, size_t n) {
    need(n);
    std::memcpy(dst, &file_[p], n);
    p += n;
  };
  uint32_t version = 0;
  rd(&version, 4);
  if (version != 4)  // M_SENSOR_DATA_VERSION, sensorData.hpp:80,125-126
    throw std::runtime_error(path_ + ": invalid file version -- found " + std::to_string(version) + " but expected 4");
  uint64_t name_len = 0;
  rd(&name_len, 8);
  need(name_len);
  p += (size_t)name_len;  // m_sensorName
  rd(color_intr_, 64);
  rd(color_extr_, 64);
  rd(depth_intr_, 64);
  rd(depth_extr_, 64);
  rd(&color_type_, 4);
  rd(&depth_type_, 4);
  rd(&color_w_, 4);
  rd(&color_h_, 4);
  rd(&depth_w_, 4);
  rd(&depth_h_, 4);
  rd(&depth_shift_, 4);
  uint64_t n_frames = 0;
  rd(&n_frames, 8);
  if (n_frames > (1u << 26)) throw std::runtime_error(path_ + ": implausible frame count");
  frames_.resize((size_t)n_frames);
  for (auto& f : frames_) {  // RGBDFrame::loadFromFile, RGBDFrame.h:286-297
    rd(f.cam_to_world, 64);
    uint64_t ts_color, ts_depth, csize, dsize;
    rd(&ts_color, 8);
    rd(&ts_depth, 8);
    rd(&csize, 8);
    rd(&dsize, 8);
    need(csize);
    f.color_off = p;
    f.color_size = (size_t)csize;
    p += (size_t)csize;
    need(dsize);
    f.depth_off = p;
    f.depth_size = (size_t)dsize;
    p += (size_t)dsize;
  }
  // IMU frames follow (sensorData.hpp:529-536); the reader does not use them
}

CameraIntrinsics<float> scannet_sens_reader::get_camera_intrinsics() const {  // scannet_sens_reader.cc:12-18
  return CameraIntrinsics<float>(depth_intr_[0], depth_intr_[5], depth_intr_[2], depth_intr_[6]);
}

SE3<float> scannet_sens_reader::get_camera_extrinsics() const {  // :20-36 (asserts there)
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (depth_extr_[4 * i + j] != (i == j ? 1.f : 0.f))
        throw std::runtime_error(path_ + ": depth extrinsics are not the identity");
  return SE3<float>::Identity();
}

void scannet_sens_reader::get_depth_frame_by_id(PngImage* out, int frame_idx) const {  // :40-53
  const FrameRec& f = frames_.at((size_t)frame_idx);
  if ((int)depth_w_ != get_width() || (int)depth_h_ != get_height())
    throw std::runtime_error(path_ + ": depth frames are not 640 x 480");
  out->width = (int)depth_w_;
  out->height = (int)depth_h_;
  out->channels = 1;
  out->bit_depth = 16;
  out->data.assign((size_t)depth_w_ * depth_h_ * 2, 0);
  if (depth_type_ == kDepthRaw) {  // RGBDFrame.cc: raw ushort
    if (f.depth_size != out->data.size()) throw std::runtime_error(path_ + ": bad raw depth size");
    std::memcpy(out->data.data(), &file_[f.depth_off], out->data.size());
  } else if (depth_type_ == kDepthZlib) {  // zlib stream of the ushort image
    uLongf n = (uLongf)out->data.size();
    if (uncompress(out->data.data(), &n, &file_[f.depth_off], (uLong)f.depth_size) != Z_OK || n != out->data.size())
      throw std::runtime_error(path_ + ": depth frame " + std::to_string(frame_idx) + " does not inflate");
  } else {
    throw std::runtime_error(path_ + ": unsupported depth compression type " + std::to_string(depth_type_));
  }
}

RgbImage scannet_sens_reader::decode_color_full(int frame_idx) const {  // :56 decompressColorAlloc
  const FrameRec& f = frames_.at((size_t)frame_idx);
  RgbImage full;
  if (color_type_ == kColorJpeg) {
    full = decode_jpeg(&file_[f.color_off], f.color_size, path_ + " frame " + std::to_string(frame_idx));
    if (full.width != (int)color_w_ || full.height != (int)color_h_)
      throw std::runtime_error(path_ + ": colour frame size differs from the header");
  } else if (color_type_ == kColorPng) {  // RGBDFrame.cc:56-63: stbi_load_from_memory(..., 3) takes PNG too
    const PngImage rgb = to_rgb8(decode_png(&file_[f.color_off], f.color_size, path_ + " frame " + std::to_string(frame_idx)));
    if (rgb.width != (int)color_w_ || rgb.height != (int)color_h_)
      throw std::runtime_error(path_ + ": colour frame size differs from the header");
    full.width = rgb.width;
    full.height = rgb.height;
    full.data = rgb.data;
  } else if (color_type_ == kColorRaw) {
    if (f.color_size != (size_t)color_w_ * color_h_ * 3) throw std::runtime_error(path_ + ": bad raw colour size");
    full.width = (int)color_w_;
    full.height = (int)color_h_;
    full.data.assign(&file_[f.color_off], &file_[f.color_off] + f.color_size);
  } else {
    throw std::runtime_error(path_ + ": unsupported colour compression type " + std::to_string(color_type_));
  }
  return full;
}

void scannet_sens_reader::get_color_frame_by_id(PngImage* out, int frame_idx) const {  // :55-66
  const RgbImage full = decode_color_full(frame_idx);
  const RgbImage small = (full.width == get_width() && full.height == get_height())
                             ? full
                             : resize_rgb_linear(full, get_width(), get_height());  // :62
  out->width = small.width;
  out->height = small.height;
  out->channels = 3;
  out->bit_depth = 8;
  out->data = small.data;
}

SE3<float> scannet_sens_reader::get_camera_pose_by_id(int frame_idx) const {  // :68-75
  // the 16 floats are a row-major camera-to-world matrix; the pose handed on is its inverse
  return SE3<float>(frames_.at((size_t)frame_idx).cam_to_world, 4).Inverse();
}

}  // namespace ratsdf
