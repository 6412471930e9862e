// ref_sens_dump -- TEST INFRASTRUCTURE, build container only.
//
// A ten-line driver around the REFERENCE's own .sens loader: it is compiled against
// /root/reference/third_party/scannet (sensorData.hpp + RGBDFrame.cc, which instantiates the vendored
// stb_image.h) by oracle/Makefile's `ref` target, with the binary going to oracle/_ref/ (git-ignored;
// no reference source is copied into this repository).  It decodes a .sens stream exactly as
// utils/offline_data_provider/scannet_sens_reader.cc:44-75 does up to -- not including -- the OpenCV /
// Eigen steps (cv::resize, SE3::Inverse), i.e. decompressColorAlloc (stb JPEG), decompressDepthAlloc
// (stb zlib), the stored camera-to-world matrix and the depth calibration, and dumps them raw.
// tests/golden/make_sens_ref_golden.py turns the dump into the committed fixture that pins
// ra-slam_amd/host/src/{sens,jpeg}.cc byte for byte.
//
//   ref_sens_dump <in.sens> <outdir>
//     outdir/meta.txt   colorW colorH depthW depthH nframes depthShift fx fy cx cy   (depth calibration)
//     outdir/<i>.color  colorW*colorH*3 bytes (RGB, before any resize)
//     outdir/<i>.depth  depthW*depthH uint16
//     outdir/poses.bin  nframes * 16 float32 (camera-to-world, row major, as stored)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "sensorData.hpp"

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s in.sens outdir\n", argv[0]);
    return 2;
  }
  try {
    ml::SensorData sd(argv[1]);
    const std::string out = argv[2];
    const auto& K = sd.m_calibrationDepth.m_intrinsic.matrix2;
    {
      std::ofstream m(out + "/meta.txt");
      m.precision(9);
      m << sd.m_colorWidth << ' ' << sd.m_colorHeight << ' ' << sd.m_depthWidth << ' ' << sd.m_depthHeight << ' '
        << sd.m_frames.size() << ' ' << sd.m_depthShift << ' ' << K[0][0] << ' ' << K[1][1] << ' ' << K[0][2]
        << ' ' << K[1][2] << '\n';
    }
    std::ofstream poses(out + "/poses.bin", std::ios::binary);
    for (size_t i = 0; i < sd.m_frames.size(); ++i) {
      ml::vec3uc* c = sd.decompressColorAlloc(i);
      unsigned short* d = sd.decompressDepthAlloc(i);
      if (!c || !d) {
        std::fprintf(stderr, "frame %zu: decode failed\n", i);
        return 1;
      }
      std::ofstream(out + "/" + std::to_string(i) + ".color", std::ios::binary)
          .write(reinterpret_cast<const char*>(c), (std::streamsize)sd.m_colorWidth * sd.m_colorHeight * 3);
      std::ofstream(out + "/" + std::to_string(i) + ".depth", std::ios::binary)
          .write(reinterpret_cast<const char*>(d), (std::streamsize)sd.m_depthWidth * sd.m_depthHeight * 2);
      poses.write(reinterpret_cast<const char*>(sd.m_frames[i].getCameraToWorld().matrix), 16 * sizeof(float));
      std::free(c);
      std::free(d);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
