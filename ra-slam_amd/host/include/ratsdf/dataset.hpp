// dataset.hpp -- offline data providers with the reference's interface
// (utils/offline_data_provider/offline_data_provider.h:15-94, folder_reader.h / folder_reader.cc),
// free of OpenCV / yaml-cpp / Eigen:
//
//   folder layout (folder_reader.h:38-52)
//     <folder>/camera_config.yaml      Camera.fx/fy/cx/cy, depthmap_factor, optional Extrinsics (16
//                                      floats, row-major 4x4)
//     <folder>/trajectory.txt          one line per frame: id + the top 3 rows of cam_T_world
//                                      (12 floats, row-major); the reader pre-multiplies the
//                                      extrinsics into every pose (folder_reader.cc:88,101)
//     <folder>/<id>_rgb.png            8-bit colour, returned in RGB order (folder_reader.cc:68-70)
//     <folder>/<id>_depth.png          16-bit grey, returned unchanged (cv::IMREAD_UNCHANGED, :60)
//
// PNG decoding is a small zlib-based reader (dataset.cc) restricted to what cv::imread produces for such
// files: non-interlaced, colour types 0 / 2 / 3 / 4 / 6, bit depth 8 or 16.  Colour reads follow
// cv::imread's default flag (8-bit 3 channels: grey is replicated, alpha dropped, 16-bit samples
// keep their high byte), depth reads keep 16-bit samples as they are.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "types.hpp"

namespace ratsdf {

struct PngImage {
  int width = 0, height = 0, channels = 0, bit_depth = 0;  // as stored (after palette expansion)
  std::vector<uint8_t> data;  // row-major, samples of 16-bit images in host byte order
};
// throws std::runtime_error with a message naming the file
PngImage read_png(const std::string& path);
PngImage decode_png(const uint8_t* data, size_t size, const std::string& name);  // the same from memory
PngImage to_rgb8(const PngImage& in);  // 8-bit RGB: grey replicated, alpha dropped, 16-bit samples >> 8

// "key: value" files as written for ORB-SLAM / OpenCV FileStorage (configs/*.yaml): scalars and
// one-level flow sequences, comments, a leading %YAML line
class YamlLite {
 public:
  explicit YamlLite(const std::string& path);
  bool has(const std::string& key) const { return values_.count(key) != 0; }
  float as_float(const std::string& key) const;                 // throws if missing
  std::vector<float> as_floats(const std::string& key) const;   // empty if missing

 private:
  std::map<std::string, std::string> values_;
};

// 8-bit RGB image, row-major
struct RgbImage {
  int width = 0, height = 0;
  std::vector<uint8_t> data;
};
// Baseline / extended-sequential Huffman JPEG (jpeg.cc): output identical, byte for byte, to what the
// reference's loader produces (third_party/scannet/stb_image/stb_image.h via RGBDFrame.cc:56-63: its
// IDCT, its chroma filters, its fixed-point colour conversion).  Throws std::runtime_error naming `name`.
RgbImage decode_jpeg(const uint8_t* data, size_t size, const std::string& name);
// cv::resize(src CV_8UC3, dst, cv::Size(out_w, out_h)) with its default INTER_LINEAR, as OpenCV
// evaluates it for 8-bit images (11-bit fixed-point coefficients; see oracle/segmentation_oracle.py)
RgbImage resize_rgb_linear(const RgbImage& src, int out_w, int out_h);

struct LogEntry {  // folder_reader.h:31-34
  int id;
  SE3<float> cam_T_world;
};

// depth image in metres as offline_eval.cc:74 produces it:
// cv::Mat::convertTo(CV_32FC1, 1. / factor) = (float)sample * (float)(1.0 / factor)
std::vector<float> depth_to_metres(const PngImage& depth, float depthmap_factor);

// utils/offline_data_provider/offline_data_provider.h:21-94 (same member names; const, and PngImage
// in place of cv::Mat)
class offline_data_provider {
 public:
  virtual ~offline_data_provider() = default;
  virtual CameraIntrinsics<float> get_camera_intrinsics() const = 0;
  virtual SE3<float> get_camera_extrinsics() const = 0;
  virtual float get_depth_map_factor() const = 0;
  virtual void get_depth_frame_by_id(PngImage* depth_img, int frame_idx) const = 0;  // 16-bit, as stored
  virtual void get_color_frame_by_id(PngImage* rgb_img, int frame_idx) const = 0;    // 8-bit RGB
  virtual SE3<float> get_camera_pose_by_id(int frame_idx) const = 0;
  virtual int get_size() const = 0;
  virtual int get_width() const = 0;
  virtual int get_height() const = 0;
};

class folder_reader : public offline_data_provider {  // same member names as the reference class
 public:
  explicit folder_reader(const std::string& folder_path);
  CameraIntrinsics<float> get_camera_intrinsics() const override;
  SE3<float> get_camera_extrinsics() const override;
  float get_depth_map_factor() const override { return depth_factor_; }
  void get_depth_frame_by_id(PngImage* depth_img, int frame_idx) const override;  // as stored
  void get_color_frame_by_id(PngImage* rgb_img, int frame_idx) const override;    // 8-bit RGB
  SE3<float> get_camera_pose_by_id(int frame_idx) const override;
  int get_size() const override { return size_; }
  int get_width() const override { return width_; }
  int get_height() const override { return height_; }

 private:
  std::vector<LogEntry> parse_log_entries() const;
  std::string logdir_;
  YamlLite camera_config_;
  std::vector<LogEntry> log_entries_;
  int size_ = 0, width_ = 0, height_ = 0;
  float depth_factor_ = 1.f;
};

// ScanNet .sens stream (third_party/scannet/sensorData.hpp: version 4 container, JPEG or raw colour,
// zlib or raw 16-bit depth) behind the reference's reader interface
// (utils/offline_data_provider/scannet_sens_reader.{h,cc}:8-82): intrinsics of the DEPTH camera,
// identity extrinsics (asserted by the reference, checked here), depth as stored (16-bit, to be
// divided by get_depth_map_factor() = m_depthShift), colour decoded and resized to 640 x 480 like
// the depth (scannet_sens_reader.cc:61-62), pose = inverse of the stored camera-to-world matrix.
class scannet_sens_reader : public offline_data_provider {
 public:
  explicit scannet_sens_reader(const std::string& sens_filepath);
  CameraIntrinsics<float> get_camera_intrinsics() const override;
  SE3<float> get_camera_extrinsics() const override;
  float get_depth_map_factor() const override { return depth_shift_; }
  void get_depth_frame_by_id(PngImage* depth_img, int frame_idx) const override;  // 16-bit, 1 channel
  void get_color_frame_by_id(PngImage* rgb_img, int frame_idx) const override;    // 8-bit RGB, 640 x 480
  SE3<float> get_camera_pose_by_id(int frame_idx) const override;
  int get_size() const override { return (int)frames_.size(); }
  int get_width() const override { return 640; }    // scannet_sens_reader.cc:78-81
  int get_height() const override { return 480; }
  // beyond the reference: what the header says
  int color_width() const { return (int)color_w_; }
  int color_height() const { return (int)color_h_; }
  int depth_width() const { return (int)depth_w_; }
  int depth_height() const { return (int)depth_h_; }
  // the colour frame as decompressColorAlloc returns it (sensorData.hpp:170-176), before cv::resize
  RgbImage decode_color_full(int frame_idx) const;

 private:
  struct FrameRec {
    float cam_to_world[16];
    size_t color_off, color_size, depth_off, depth_size;
  };
  std::string path_;
  std::vector<uint8_t> file_;  // the whole stream (ScanNet scenes: a few hundred MB to a few GB)
  float color_intr_[16], color_extr_[16], depth_intr_[16], depth_extr_[16];
  int32_t color_type_ = -1, depth_type_ = -1;
  uint32_t color_w_ = 0, color_h_ = 0, depth_w_ = 0, depth_h_ = 0;
  float depth_shift_ = 1000.f;
  std::vector<FrameRec> frames_;
};

// Frames of a folder in order, decoded ahead of the consumer by a few threads (a 640x480 PNG pair
// takes ~5 ms to inflate and unfilter; the engine integrates a frame in ~40 us).  Not in the
// reference, whose harness decodes on the integration thread (offline_eval.cc:66-75).
struct Frame {
  PngImage rgb;              // 8-bit RGB
  std::vector<float> depth;  // metres
  SE3<float> pose;           // cam_T_world, extrinsics included (folder_reader.cc:101)
};

class FramePrefetcher {
 public:
  FramePrefetcher(const offline_data_provider& reader, int n_frames, int threads);
  ~FramePrefetcher();
  bool next(Frame* out);  // false after the last frame; rethrows a decoding error of that frame

 private:
  struct Slot {
    Frame frame;
    std::string error;
    int ready_for = -1;
  };
  void work();
  const offline_data_provider& reader_;
  const int n_;
  std::vector<Slot> slots_;
  std::vector<std::thread> workers_;
  std::mutex mtx_;
  std::condition_variable cv_;
  int next_decode_ = 0, consumed_ = 0;
  bool stop_ = false;
};

}  // namespace ratsdf
