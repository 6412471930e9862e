// dataset.cc -- folder_reader and its helpers (see dataset.hpp).  Follows
// utils/offline_data_provider/folder_reader.cc:9-105 member by member.
#include "ratsdf/dataset.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <stdexcept>

namespace ratsdf {

// ---- PNG --------------------------------------------------------------------------------------
namespace {

uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

// Scratch buffers are per thread and only ever grow: fresh megabyte-sized vectors per image mean
// mmap / page-fault traffic that serialises the decoder threads on the process's memory map.
void read_file(const std::string& path, std::vector<uint8_t>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) {
    fclose(f);
    throw std::runtime_error("cannot read " + path);
  }
  if (out->size() < (size_t)n) out->resize((size_t)n);
  const size_t got = fread(out->data(), 1, (size_t)n, f);
  fclose(f);
  if (got != (size_t)n) throw std::runtime_error("cannot read " + path);
  out->resize((size_t)n);
}

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

}  // namespace

PngImage read_png(const std::string& path) {
  static thread_local std::vector<uint8_t> bytes;
  read_file(path, &bytes);
  return decode_png(bytes.data(), bytes.size(), path);
}

PngImage decode_png(const uint8_t* file_data, size_t file_size, const std::string& path) {
  static thread_local std::vector<uint8_t> idat, raw, pix;
  struct View {  // (the parser below indexes `file` like the vector it used to be)
    const uint8_t* p;
    size_t n;
    size_t size() const { return n; }
    const uint8_t* data() const { return p; }
    const uint8_t& operator[](size_t i) const { return p[i]; }
  } file{file_data, file_size};
  idat.clear();
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) throw std::runtime_error(path + ": not a PNG");
  uint32_t width = 0, height = 0;
  int bit_depth = 0, color_type = -1, interlace = 0;
  std::vector<uint8_t> palette;
  size_t at = 8;
  bool end = false;
  while (!end && at + 12 <= file.size()) {
    const uint32_t len = be32(&file[at]);
    const char* type = reinterpret_cast<const char*>(&file[at + 4]);
    if (at + 12 + (size_t)len > file.size()) throw std::runtime_error(path + ": truncated chunk");
    const uint8_t* data = &file[at + 8];
    const uint32_t crc = be32(&file[at + 8 + len]);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), &file[at + 4], len + 4) != crc)
      throw std::runtime_error(path + ": chunk CRC mismatch");
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) throw std::runtime_error(path + ": bad IHDR");
      width = be32(data);
      height = be32(data + 4);
      bit_depth = data[8];
      color_type = data[9];
      interlace = data[12];
    } else if (!memcmp(type, "PLTE", 4)) {
      palette.assign(data, data + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      end = true;
    }
    at += 12 + (size_t)len;
  }
  int channels;
  switch (color_type) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error(path + ": unsupported colour type");
  }
  if (interlace) throw std::runtime_error(path + ": interlaced PNG not supported");
  if (!(bit_depth == 8 || (bit_depth == 16 && color_type != 3)))
    throw std::runtime_error(path + ": unsupported bit depth");
  if (width == 0 || height == 0 || width > (1u << 15) || height > (1u << 15))
    throw std::runtime_error(path + ": bad size");
  const size_t bpp = (size_t)channels * bit_depth / 8;  // bytes per complete pixel
  const size_t stride = bpp * width;
  raw.resize((stride + 1) * height);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    throw std::runtime_error(path + ": inflate failed");
  // undo the scanline filters in place (PNG specification, section 9)
  pix.resize(stride * height);
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t* in = &raw[(stride + 1) * y + 1];
    uint8_t* out = &pix[stride * y];
    const uint8_t* up = y ? &pix[stride * (y - 1)] : nullptr;
    // one tight loop per filter type (the first bpp bytes of a row have no left neighbour)
    switch (ft) {
      case 0:
        memcpy(out, in, stride);
        break;
      case 1:
        for (size_t i = 0; i < bpp; ++i) out[i] = in[i];
        for (size_t i = bpp; i < stride; ++i) out[i] = (uint8_t)(in[i] + out[i - bpp]);
        break;
      case 2:
        if (up) {
          for (size_t i = 0; i < stride; ++i) out[i] = (uint8_t)(in[i] + up[i]);
        } else {
          memcpy(out, in, stride);
        }
        break;
      case 3:
        for (size_t i = 0; i < bpp; ++i) out[i] = (uint8_t)(in[i] + ((up ? up[i] : 0) >> 1));
        for (size_t i = bpp; i < stride; ++i)
          out[i] = (uint8_t)(in[i] + ((out[i - bpp] + (up ? up[i] : 0)) >> 1));
        break;
      case 4:
        for (size_t i = 0; i < bpp; ++i) out[i] = (uint8_t)(in[i] + paeth(0, up ? up[i] : 0, 0));
        for (size_t i = bpp; i < stride; ++i)
          out[i] = (uint8_t)(in[i] + paeth(out[i - bpp], up ? up[i] : 0, up ? up[i - bpp] : 0));
        break;
      default:
        throw std::runtime_error(path + ": bad filter type");
    }
  }
  PngImage img;
  img.width = (int)width;
  img.height = (int)height;
  img.bit_depth = bit_depth;
  if (color_type == 3) {  // palette -> RGB
    img.channels = 3;
    img.data.resize((size_t)width * height * 3);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
      const size_t p = (size_t)pix[i] * 3;
      if (p + 3 > palette.size()) throw std::runtime_error(path + ": palette index out of range");
      memcpy(&img.data[i * 3], &palette[p], 3);
    }
    return img;
  }
  img.channels = channels;
  if (bit_depth == 16) {  // big-endian samples -> host order
    img.data.resize(pix.size());
    uint16_t* d = reinterpret_cast<uint16_t*>(img.data.data());
    for (size_t i = 0; i < pix.size() / 2; ++i) d[i] = (uint16_t)((pix[2 * i] << 8) | pix[2 * i + 1]);
  } else {
    img.data.assign(pix.begin(), pix.end());
  }
  return img;
}

// what cv::imread(path) (IMREAD_COLOR) followed by cv::COLOR_BGR2RGB leaves: 8-bit RGB -- and what
// stbi_load_from_memory(..., 3) leaves for a PNG (grey replicated, alpha dropped, 16-bit samples >> 8)
PngImage to_rgb8(const PngImage& in) {
  PngImage out;
  out.width = in.width;
  out.height = in.height;
  out.channels = 3;
  out.bit_depth = 8;
  const size_t n = (size_t)in.width * in.height;
  out.data.resize(n * 3);
  auto sample = [&](size_t pixel, int ch) -> uint8_t {
    const size_t i = pixel * in.channels + ch;
    if (in.bit_depth == 16) return (uint8_t)(reinterpret_cast<const uint16_t*>(in.data.data())[i] >> 8);
    return in.data[i];
  };
  for (size_t p = 0; p < n; ++p) {
    if (in.channels <= 2) {  // grey (+ alpha): replicated
      const uint8_t g = sample(p, 0);
      out.data[p * 3] = out.data[p * 3 + 1] = out.data[p * 3 + 2] = g;
    } else {  // RGB (+ alpha): alpha dropped
      for (int c = 0; c < 3; ++c) out.data[p * 3 + c] = sample(p, c);
    }
  }
  return out;
}

// ---- YAML -------------------------------------------------------------------------------------
YamlLite::YamlLite(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string line, key, pending;
  auto strip = [](std::string s) {
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return std::string();
    const size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
  };
  while (std::getline(f, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    line = strip(line);
    if (line.empty() || line[0] == '%' || line.rfind("---", 0) == 0) continue;
    if (!pending.empty()) {  // inside a flow sequence that spans lines
      pending += " " + line;
      if (line.find(']') != std::string::npos) {
        values_[key] = pending;
        pending.clear();
      }
      continue;
    }
    const size_t colon = line.find(':');
    if (colon == std::string::npos) continue;
    key = strip(line.substr(0, colon));
    std::string val = strip(line.substr(colon + 1));
    if (!val.empty() && val[0] == '[' && val.find(']') == std::string::npos) {
      pending = val;
      continue;
    }
    if (val.size() >= 2 && (val.front() == '"' || val.front() == '\'') && val.back() == val.front())
      val = val.substr(1, val.size() - 2);
    values_[key] = val;
  }
}

float YamlLite::as_float(const std::string& key) const {
  const auto it = values_.find(key);
  if (it == values_.end()) throw std::runtime_error("camera_config.yaml: missing key " + key);
  return strtof(it->second.c_str(), nullptr);
}

std::vector<float> YamlLite::as_floats(const std::string& key) const {
  std::vector<float> out;
  const auto it = values_.find(key);
  if (it == values_.end()) return out;
  std::string s = it->second;
  for (char& c : s)
    if (c == '[' || c == ']' || c == ',') c = ' ';
  std::istringstream is(s);
  std::string tok;
  while (is >> tok) out.push_back(strtof(tok.c_str(), nullptr));
  return out;
}

// ---- depth scaling ----------------------------------------------------------------------------
std::vector<float> depth_to_metres(const PngImage& depth, float depthmap_factor) {
  if (depth.channels != 1) throw std::runtime_error("depth image must have one channel");
  // offline_eval.cc:74: img_depth.convertTo(img_depth, CV_32FC1, 1. / factor): alpha is computed in
  // double, and OpenCV's 16U/8U -> 32F conversion multiplies in float by (float)alpha
  const float alpha = (float)(1.0 / (double)depthmap_factor);
  const size_t n = (size_t)depth.width * depth.height;
  std::vector<float> out(n);
  if (depth.bit_depth == 16) {
    const uint16_t* d = reinterpret_cast<const uint16_t*>(depth.data.data());
    for (size_t i = 0; i < n; ++i) out[i] = (float)d[i] * alpha;
  } else {
    for (size_t i = 0; i < n; ++i) out[i] = (float)depth.data[i] * alpha;
  }
  return out;
}

// ---- folder_reader ----------------------------------------------------------------------------
folder_reader::folder_reader(const std::string& folder_path)
    : logdir_(folder_path), camera_config_(folder_path + "/camera_config.yaml") {  // :9-13
  log_entries_ = parse_log_entries();                                             // :16
  size_ = (int)log_entries_.size();
  if (size_ == 0) throw std::runtime_error(folder_path + "/trajectory.txt: no entries");
  PngImage test_rgb, test_depth;                                                   // :20-27
  get_depth_frame_by_id(&test_depth, 0);
  get_color_frame_by_id(&test_rgb, size_ - 1);
  if (test_depth.width != test_rgb.width || test_depth.height != test_rgb.height)
    throw std::runtime_error(folder_path + ": depth and colour sizes differ");
  width_ = test_depth.width;
  height_ = test_depth.height;
  depth_factor_ = camera_config_.as_float("depthmap_factor");
}

CameraIntrinsics<float> folder_reader::get_camera_intrinsics() const {  // :30-36
  return CameraIntrinsics<float>(camera_config_.as_float("Camera.fx"), camera_config_.as_float("Camera.fy"),
                                 camera_config_.as_float("Camera.cx"), camera_config_.as_float("Camera.cy"));
}

SE3<float> folder_reader::get_camera_extrinsics() const {  // :38-50
  const std::vector<float> e = camera_config_.as_floats("Extrinsics");
  if (e.empty()) return SE3<float>::Identity();
  if (e.size() != 16) throw std::runtime_error("camera_config.yaml: Extrinsics needs 16 values");
  return SE3<float>(e.data(), 4);
}

void folder_reader::get_depth_frame_by_id(PngImage* depth_img, int frame_idx) const {  // :54-61
  if (frame_idx < 0 || frame_idx >= size_) throw std::runtime_error("invalid frame index");
  const int id = log_entries_[frame_idx].id;
  *depth_img = read_png(logdir_ + "/" + std::to_string(id) + "_depth.png");
}

void folder_reader::get_color_frame_by_id(PngImage* rgb_img, int frame_idx) const {  // :63-71
  if (frame_idx < 0 || frame_idx >= size_) throw std::runtime_error("invalid frame index");
  const int id = log_entries_[frame_idx].id;
  *rgb_img = to_rgb8(read_png(logdir_ + "/" + std::to_string(id) + "_rgb.png"));
}

SE3<float> folder_reader::get_camera_pose_by_id(int frame_idx) const {  // :73-78
  if (frame_idx < 0 || frame_idx >= size_) throw std::runtime_error("invalid frame index");
  return log_entries_[frame_idx].cam_T_world;
}

std::vector<LogEntry> folder_reader::parse_log_entries() const {  // :86-105
  const SE3<float> extrinsics = get_camera_extrinsics();
  std::vector<LogEntry> entries;
  std::ifstream fin(logdir_ + "/trajectory.txt");
  if (!fin) throw std::runtime_error("cannot open " + logdir_ + "/trajectory.txt");
  int id;
  float b[12];
  // the last row (0 0 0 1) is not saved
  while (fin >> id >> b[0] >> b[1] >> b[2] >> b[3] >> b[4] >> b[5] >> b[6] >> b[7] >> b[8] >> b[9] >>
         b[10] >> b[11]) {
    entries.push_back({id, extrinsics * SE3<float>(b, 4)});  // 3x4 row-major, row stride 4
  }
  return entries;
}

// ---- prefetching frame source -----------------------------------------------------------------
FramePrefetcher::FramePrefetcher(const offline_data_provider& reader, int n_frames, int threads)
    : reader_(reader), n_(n_frames), slots_((size_t)std::max(2, 2 * std::max(1, threads))) {
  for (int t = 0; t < std::max(1, threads); ++t) workers_.emplace_back([this] { work(); });
}

FramePrefetcher::~FramePrefetcher() {
  {
    std::lock_guard<std::mutex> lock(mtx_);
    stop_ = true;
  }
  cv_.notify_all();
  for (auto& w : workers_) w.join();
}

void FramePrefetcher::work() {
  for (;;) {
    int idx;
    {
      std::unique_lock<std::mutex> lock(mtx_);
      // take the next frame index, but never run more than the ring ahead of the consumer
      cv_.wait(lock, [this] { return stop_ || (next_decode_ < n_ && next_decode_ < consumed_ + (int)slots_.size()); });
      if (stop_ || next_decode_ >= n_) return;
      idx = next_decode_++;
    }
    Slot& s = slots_[(size_t)idx % slots_.size()];
    try {
      s.frame.pose = reader_.get_camera_pose_by_id(idx);
      reader_.get_color_frame_by_id(&s.frame.rgb, idx);
      PngImage d;
      reader_.get_depth_frame_by_id(&d, idx);
      s.frame.depth = depth_to_metres(d, reader_.get_depth_map_factor());
      s.error.clear();
    } catch (const std::exception& e) {
      s.error = e.what();
    }
    {
      std::lock_guard<std::mutex> lock(mtx_);
      s.ready_for = idx;
    }
    cv_.notify_all();
  }
}

bool FramePrefetcher::next(Frame* out) {
  if (consumed_ >= n_) return false;
  Slot& s = slots_[(size_t)consumed_ % slots_.size()];
  {
    std::unique_lock<std::mutex> lock(mtx_);
    cv_.wait(lock, [&] { return s.ready_for == consumed_; });
  }
  if (!s.error.empty()) throw std::runtime_error(s.error);
  std::swap(*out, s.frame);
  {
    std::lock_guard<std::mutex> lock(mtx_);
    s.ready_for = -1;
    ++consumed_;
  }
  cv_.notify_all();
  return true;
}

}  // namespace ratsdf
