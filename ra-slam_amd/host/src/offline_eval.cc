// offline_eval.cc -- the reference's offline harness (main/offline_eval.cc:37-99) on the host layer:
// a folder dataset is read frame by frame and integrated through TSDFSystem; the map can then be
// written in the reference's formats (DownloadAll records, DownloadAllMesh triple).
//
// Differences that are deliberate:
//   * no segmentation network here: Integrate gets empty ht / lt, i.e. the all-ones
//     images of modules/tsdf_module.cc:27-31;
//   * the queue is drained (Flush) before the downloads; the reference calls terminate() right after
//     the last Integrate, which drops whatever is still queued (SURVEY 8b quirk i);
//   * kept as is: the reader's extrinsics are passed to TSDFSystem although folder_reader has already
//     multiplied them into every pose (SURVEY 8b quirk iii, offline_eval.cc:57);
//   * <folder> may be a ScanNet .sens stream (scannet_sens_reader, as examples/scannet_evaluation/
//     eval_one.cc:41-87 uses it).
//
// usage: ratsdf_offline_eval <folder> [--lib libratsdf.so] [--voxel 0.01]
//          [--max-depth 6] [--device 0] [--frames N] [--download-all FILE] [--download-mesh PREFIX]
//          [--reader-only] [--dump-frames DIR] [--dump-raw-color (.sens: also DIR/<i>.color, the colour
//          frame before the resize, and DIR/raw_meta.txt = its width and height)]
//          [--threads N (decoder threads, default 4)]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>

#include "ratsdf/dataset.hpp"
#include "ratsdf/tsdf_system.hpp"

using namespace ratsdf;

namespace {
void write_file(const std::string& path, const void* data, size_t bytes) {
  std::ofstream f(path, std::ios::binary);
  f.write(static_cast<const char*>(data), (std::streamsize)bytes);
  if (!f) {
    fprintf(stderr, "cannot write %s\n", path.c_str());
    exit(2);
  }
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <folder> [options]\n", argv[0]);
    return 2;
  }
  const std::string data_path = argv[1];
  const char* lib = nullptr;
  float voxel_size = 0.01f, max_depth = 6.f;  // offline_eval.cc:49-53
  int device = 0, max_frames = -1, threads = 4;
  std::string download_all, download_mesh, dump_dir;
  bool reader_only = false, dump_raw_color = false;
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) {
        fprintf(stderr, "missing value after %s\n", a.c_str());
        exit(2);
      }
      return argv[++i];
    };
    if (a == "--lib") lib = next();
    else if (a == "--voxel") voxel_size = strtof(next(), nullptr);
    else if (a == "--max-depth") max_depth = strtof(next(), nullptr);
    else if (a == "--device") device = atoi(next());
    else if (a == "--frames") max_frames = atoi(next());
    else if (a == "--threads") threads = atoi(next());
    else if (a == "--download-all") download_all = next();
    else if (a == "--download-mesh") download_mesh = next();
    else if (a == "--dump-frames") dump_dir = next();
    else if (a == "--reader-only") reader_only = true;
    else if (a == "--dump-raw-color") dump_raw_color = true;
    else {
      fprintf(stderr, "unknown option %s\n", a.c_str());
      return 2;
    }
  }
  const bool is_sens = data_path.size() > 5 && data_path.substr(data_path.size() - 5) == ".sens";
  try {
    std::unique_ptr<offline_data_provider> provider;
    if (is_sens) provider = std::make_unique<scannet_sens_reader>(data_path);
    else provider = std::make_unique<folder_reader>(data_path);
    const offline_data_provider& reader = *provider;
    int n = reader.get_size();
    if (max_frames >= 0 && max_frames < n) n = max_frames;
    const CameraIntrinsics<float> K = reader.get_camera_intrinsics();
    const SE3<float> ext = reader.get_camera_extrinsics();
    if (!dump_dir.empty()) {
      std::ofstream meta(dump_dir + "/meta.txt");
      meta.precision(9);
      const ratsdf_pose e = ext.abi();
      meta << reader.get_width() << " " << reader.get_height() << " " << n << " " << K.fx << " " << K.fy << " "
           << K.cx << " " << K.cy << " " << reader.get_depth_map_factor() << "\n"
           << e.qx << " " << e.qy << " " << e.qz << " " << e.qw << " " << e.tx << " " << e.ty << " " << e.tz
           << "\n";
    }
    if (dump_raw_color && is_sens && !dump_dir.empty()) {
      const auto& sr = static_cast<const scannet_sens_reader&>(reader);
      std::ofstream(dump_dir + "/raw_meta.txt") << sr.color_width() << " " << sr.color_height() << "\n";
      for (int i = 0; i < n; ++i) {
        const RgbImage full = sr.decode_color_full(i);
        write_file(dump_dir + "/" + std::to_string(i) + ".color", full.data.data(), full.data.size());
      }
    }
    std::unique_ptr<TSDFSystem> tsdf;
    if (!reader_only)
      tsdf = std::make_unique<TSDFSystem>(voxel_size, voxel_size * 6, max_depth, K, ext, device,
                                          &Api::Load(lib));
    fprintf(stderr, "[offline_eval] stream size %d (%dx%d)\n", n, reader.get_width(), reader.get_height());
    std::vector<float> poses;
    double t_read = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    FramePrefetcher source(reader, n, threads);  // decodes ahead; frames still arrive in order
    Frame fr;
    for (int frame_idx = 0; frame_idx < n; ++frame_idx) {  // offline_eval.cc:66-85
      if (tsdf && tsdf->is_terminated()) break;
      const auto t0 = std::chrono::steady_clock::now();
      if (!source.next(&fr)) break;  // pose, colour frame, depth frame in metres (:69-74)
      t_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const SE3<float>& cam_T_world = fr.pose;
      const PngImage& rgb = fr.rgb;
      const std::vector<float>& depth = fr.depth;
      if (!dump_dir.empty()) {
        const std::string stem = dump_dir + "/" + std::to_string(frame_idx);
        write_file(stem + ".rgb", rgb.data.data(), rgb.data.size());
        write_file(stem + ".depth", depth.data(), depth.size() * 4);
        const ratsdf_pose p = cam_T_world.abi();
        const float v[7] = {p.qx, p.qy, p.qz, p.qw, p.tx, p.ty, p.tz};
        poses.insert(poses.end(), v, v + 7);
      }
      if (tsdf) {
        const Image img_rgb{rgb.data.data(), rgb.height, rgb.width, kU8C3};
        const Image img_depth{depth.data(), rgb.height, rgb.width, kF32C1};
        tsdf->Integrate(cam_T_world, img_rgb, img_depth);  // no ht / lt: ones (tsdf_module.cc:27-31)
      }
    }
    if (!dump_dir.empty()) write_file(dump_dir + "/poses.bin", poses.data(), poses.size() * 4);
    if (tsdf) {
      tsdf->Flush();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      fprintf(stderr, "[offline_eval] %d frames in %.3f s (%.1f frames/s; waited %.3f s for the decoders)\n", n, dt,
              n / dt, t_read);
      if (!download_all.empty()) tsdf->DownloadAll(download_all);
      if (!download_mesh.empty())  // offline_eval.cc:95-98
        tsdf->DownloadAllMesh(download_mesh + "_vertices.bin", download_mesh + "_indices.bin",
                              download_mesh + "_vertices_prob.bin");
      tsdf->terminate();
    } else {
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      fprintf(stderr, "[offline_eval] read %d frames in %.3f s (%.1f frames/s, %d decoder threads)\n", n, dt,
              n / dt, threads);
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "[offline_eval] %s\n", e.what());
    return 1;
  }
  printf("OK\n");
  return 0;
}
