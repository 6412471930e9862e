// jpeg.cc -- baseline / extended-sequential Huffman JPEG decoder for the colour frames of ScanNet
// .sens streams (third_party/scannet/sensorData.hpp:170-176 -> RGBDFrame.cc:56-63: the reference
// decodes them with its vendored third_party/scannet/stb_image/stb_image.h, v2.08, three channels
// requested).
//
// Colour bytes feed voxel rgb, and byte work has to be bit-exact, so this decoder restates the
// arithmetic of THAT decoder -- not libjpeg's, which differs from it in every stage after the entropy
// decoder (up to 3 LSB on ~8 % of the bytes of a 4:2:0 picture):
//   dequantisation      coefficient * table entry, kept in 16 bits                 stb_image.h:1735,1765
//   inverse DCT         jidctint's structure with 12-bit constants (x * 4096 + 0.5, truncated), column
//                       pass first (+512 >> 10, DC-only shortcut), row pass without a shortcut
//                       (+65536 + (128 << 17) >> 17), clamp to 0..255                     :1930-2028
//   chroma upsampling   "jfif-centered" filters: 2:1 horizontal (3a + b + 2) >> 2 on BOTH sides,
//                       2:1 vertical (3 near + far + 2) >> 2, 2x2 = vertical 3:1 sums then
//                       (3 t0 + t1 + 8) >> 4, everything else nearest; which rows are "near" and
//                       "far" follows the decoder's line-stepping state machine          :2864-3067,3345-3385
//   YCbCr -> RGB        20-bit fixed point with the coefficients rounded to 12 bits and the Cb term
//                       of green masked to its upper 16 bits                              :3094-3120
// stb's SSE2 kernels (what an x86-64 build of the reference runs) are written to produce the same bits
// as the scalar code restated here; tests/test_sens_reader.py holds fixtures decoded by BOTH builds of
// the reference's loader (oracle/_ref, build container only) and they are identical.
// Pinned by: tests/golden/sens_ref_*.npz (made by tests/golden/make_sens_ref_golden.py from the
// reference's own loader).
// Supported like stb: 8-bit precision, 1 or 3 components, sampling factors 1..4, interleaved and
// non-interleaved scans, restart intervals, 8-bit quantisation tables.  Not supported: progressive
// (stb handles it; ScanNet's recorder never writes it), arithmetic coding, 12-bit, CMYK.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ratsdf/dataset.hpp"

namespace ratsdf {
namespace {

const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  void build(const uint8_t* counts, const uint8_t* symbols, int n) {
    std::memcpy(vals, symbols, (size_t)n);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      code += counts[len - 1];
      k += counts[len - 1];
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    present = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int stride = 0, rows = 0;   // allocated plane: whole interleaved MCUs (img_comp.w2 / h2)
  int width = 0, height = 0;  // samples that exist: ceil(image * samp / max_samp) (img_comp.x / y)
  int pred = 0;
  std::vector<uint8_t> plane;
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int nbits = 0;
  bool hit_marker = false;
  void fill() {
    while (nbits <= 24) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;
          } else {  // a marker: feed zeros until the caller deals with it
            hit_marker = true;
            b = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= (uint32_t)b << (24 - nbits);
      nbits += 8;
    }
  }
  int get(int n) {
    if (n == 0) return 0;
    if (nbits < n) fill();
    const int v = (int)(acc >> (32 - n));
    acc <<= n;
    nbits -= n;
    return v;
  }
  void reset() {
    acc = 0;
    nbits = 0;
    hit_marker = false;
  }
};

inline int decode_symbol(BitReader& br, const Huff& h, const std::string& name) {
  int code = br.get(1);
  int len = 1;
  while (len <= 16 && (h.maxcode[len] < 0 || code > h.maxcode[len])) {
    code = (code << 1) | br.get(1);
    ++len;
  }
  if (len > 16) throw std::runtime_error(name + ": corrupt JPEG (bad Huffman code)");
  return h.vals[h.valptr[len] + code - h.mincode[len]];
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// ---- inverse DCT (stb_image.h:1930-2028) ----------------------------------------------------------
// Fixed-point constants exactly as the reference forms them: the float literal times 4096, plus 0.5 in
// double, truncated toward zero (so the negative ones are NOT rounded to nearest).
constexpr int fx12(float x) { return (int)(x * 4096 + 0.5); }
constexpr int K_0_5411961 = fx12(0.5411961f), K_M1_847759065 = fx12(-1.847759065f),
              K_0_765366865 = fx12(0.765366865f), K_1_175875602 = fx12(1.175875602f),
              K_0_298631336 = fx12(0.298631336f), K_2_053119869 = fx12(2.053119869f),
              K_3_072711026 = fx12(3.072711026f), K_1_501321110 = fx12(1.501321110f),
              K_M0_899976223 = fx12(-0.899976223f), K_M2_562915447 = fx12(-2.562915447f),
              K_M1_961570560 = fx12(-1.961570560f), K_M0_390180644 = fx12(-0.390180644f);
static_assert(K_0_5411961 == 2217 && K_M1_847759065 == -7567 && K_0_765366865 == 3135 &&
                  K_1_175875602 == 4816 && K_0_298631336 == 1223 && K_2_053119869 == 8410 &&
                  K_3_072711026 == 12586 && K_1_501321110 == 6149 && K_M0_899976223 == -3685 &&
                  K_M2_562915447 == -10497 && K_M1_961570560 == -8034 && K_M0_390180644 == -1597,
              "12-bit IDCT constants");

// One 8-point pass: even part -> e0..e3 (still to be biased by the caller), odd part -> o0..o3.
// out[k] = e[k] + o[3 - k]  and  out[7 - k] = e[k] - o[3 - k]  (k = 0..3) after the caller's bias.
struct Pass {
  int e0, e1, e2, e3, o0, o1, o2, o3;
};
inline Pass idct8(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
  Pass r;
  const int z = (s2 + s6) * K_0_5411961;
  const int t2 = z + s6 * K_M1_847759065;
  const int t3 = z + s2 * K_0_765366865;
  const int t0 = (s0 + s4) << 12, t1 = (s0 - s4) << 12;
  r.e0 = t0 + t3;
  r.e3 = t0 - t3;
  r.e1 = t1 + t2;
  r.e2 = t1 - t2;
  const int a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
  const int m = (a + b) * K_1_175875602;
  const int pc = m + c * K_M0_899976223;
  const int pd = m + d * K_M2_562915447;
  const int pa = a * K_M1_961570560;
  const int pb = b * K_M0_390180644;
  r.o3 = s1 * K_1_501321110 + (pc + pb);
  r.o2 = s3 * K_3_072711026 + (pd + pa);
  r.o1 = s5 * K_2_053119869 + (pd + pb);
  r.o0 = s7 * K_0_298631336 + (pc + pa);
  return r;
}
inline uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void idct_block(const int16_t* d, uint8_t* out, int stride) {
  int ws[64];
  for (int c = 0; c < 8; ++c) {  // columns: 2 extra bits of precision kept
    const int16_t* in = d + c;
    int* w = ws + c;
    if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
      const int dc = in[0] << 2;
      for (int r = 0; r < 8; ++r) w[8 * r] = dc;
      continue;
    }
    Pass p = idct8(in[0], in[8], in[16], in[24], in[32], in[40], in[48], in[56]);
    p.e0 += 512; p.e1 += 512; p.e2 += 512; p.e3 += 512;
    w[0] = (p.e0 + p.o3) >> 10;
    w[56] = (p.e0 - p.o3) >> 10;
    w[8] = (p.e1 + p.o2) >> 10;
    w[48] = (p.e1 - p.o2) >> 10;
    w[16] = (p.e2 + p.o1) >> 10;
    w[40] = (p.e2 - p.o1) >> 10;
    w[24] = (p.e3 + p.o0) >> 10;
    w[32] = (p.e3 - p.o0) >> 10;
  }
  for (int r = 0; r < 8; ++r) {  // rows: remove 1 << 17, round, re-centre on 128
    const int* w = ws + 8 * r;
    uint8_t* o = out + (size_t)r * stride;
    Pass p = idct8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    const int bias = 65536 + (128 << 17);
    p.e0 += bias; p.e1 += bias; p.e2 += bias; p.e3 += bias;
    o[0] = clamp255((p.e0 + p.o3) >> 17);
    o[7] = clamp255((p.e0 - p.o3) >> 17);
    o[1] = clamp255((p.e1 + p.o2) >> 17);
    o[6] = clamp255((p.e1 - p.o2) >> 17);
    o[2] = clamp255((p.e2 + p.o1) >> 17);
    o[5] = clamp255((p.e2 - p.o1) >> 17);
    o[3] = clamp255((p.e3 + p.o0) >> 17);
    o[4] = clamp255((p.e3 - p.o0) >> 17);
  }
}

// ---- chroma upsampling (stb_image.h:2864-3067): one output row from the nearer and the farther
// input row of `w` samples; returns the row to read from (the input itself when nothing is to do) ---
const uint8_t* up_v2(uint8_t* out, const uint8_t* near, const uint8_t* far, int w) {
  for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
  return out;
}
const uint8_t* up_h2(uint8_t* out, const uint8_t* in, int w) {
  if (w == 1) {
    out[0] = out[1] = in[0];
    return out;
  }
  out[0] = in[0];
  out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
  for (int i = 1; i < w - 1; ++i) {
    const int n = 3 * in[i] + 2;
    out[2 * i] = (uint8_t)((n + in[i - 1]) >> 2);
    out[2 * i + 1] = (uint8_t)((n + in[i + 1]) >> 2);
  }
  out[2 * w - 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
  out[2 * w - 1] = in[w - 1];
  return out;
}
const uint8_t* up_h2v2(uint8_t* out, const uint8_t* near, const uint8_t* far, int w) {
  if (w == 1) {
    out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
    return out;
  }
  int t1 = 3 * near[0] + far[0];
  out[0] = (uint8_t)((t1 + 2) >> 2);
  for (int i = 1; i < w; ++i) {
    const int t0 = t1;
    t1 = 3 * near[i] + far[i];
    out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
    out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
  }
  out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
  return out;
}
const uint8_t* up_nearest(uint8_t* out, const uint8_t* near, int w, int hs) {
  for (int i = 0; i < w; ++i)
    for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
  return out;
}

// ---- YCbCr -> RGB (stb_image.h:3094-3120) ---------------------------------------------------------
constexpr int fx20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }
constexpr int kCrR = fx20(1.40200f), kCrG = fx20(0.71414f), kCbG = fx20(0.34414f), kCbB = fx20(1.77200f);
static_assert(kCrR == 5743 << 8 && kCrG == 2925 << 8 && kCbG == 1410 << 8 && kCbB == 7258 << 8,
              "colour constants");

}  // namespace

RgbImage decode_jpeg(const uint8_t* data, size_t size, const std::string& name) {
  auto fail = [&](const char* what) -> void { throw std::runtime_error(name + ": " + what); };
  if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) fail("not a JPEG stream");
  uint8_t quant[4][64] = {};
  bool have_q[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<Component> comp;
  int width = 0, height = 0, hmax = 1, vmax = 1, restart = 0, mcus_x = 0, mcus_y = 0;
  int scans = 0;
  bool done = false;
  size_t pos = 2;
  auto u16 = [&](size_t p) { return (int)data[p] << 8 | data[p + 1]; };

  // one entropy-coded block -> dequantised coefficients in natural order -> 8x8 samples
  int16_t coef[64];
  auto decode_block = [&](BitReader& br, Component& c, uint8_t* out) {
    std::memset(coef, 0, sizeof(coef));
    const uint8_t* q = quant[c.tq];
    const int t = decode_symbol(br, dc[c.td], name);
    if (t > 15) fail("corrupt JPEG (DC size)");
    c.pred += t ? extend(br.get(t), t) : 0;
    coef[0] = (int16_t)(c.pred * q[0]);
    for (int k = 1; k < 64;) {
      const int rs = decode_symbol(br, ac[c.ta], name);
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (rs != 0xF0) break;  // EOB
        k += 16;                // ZRL
        continue;
      }
      k += r;
      if (k > 63) fail("corrupt JPEG (AC run)");
      const int z = kZigZag[k];
      coef[z] = (int16_t)(extend(br.get(s), s) * q[z]);
      ++k;
    }
    idct_block(coef, out, c.stride);
  };

  while (!done) {
    if (pos + 2 > size) fail("truncated JPEG");
    if (data[pos] != 0xFF) fail("corrupt JPEG (marker expected)");
    while (pos < size && data[pos] == 0xFF) ++pos;  // fill bytes
    if (pos >= size) fail("truncated JPEG");
    const int m = data[pos++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) break;
    if (pos + 2 > size) fail("truncated JPEG");
    const int len = u16(pos);
    if (len < 2 || pos + (size_t)len > size) fail("truncated JPEG");
    const uint8_t* seg = data + pos + 2;
    const int n = len - 2;
    if (m == 0xDB) {  // DQT
      int i = 0;
      while (i < n) {
        const int pq = seg[i] >> 4, tq = seg[i] & 15;
        ++i;
        if (pq != 0) fail("unsupported JPEG (16-bit quantisation table)");  // as stb: "bad DQT type"
        if (tq > 3 || i + 64 > n) fail("corrupt JPEG (DQT)");
        for (int k = 0; k < 64; ++k) quant[tq][kZigZag[k]] = seg[i++];
        have_q[tq] = true;
      }
    } else if (m == 0xC4) {  // DHT
      int i = 0;
      while (i + 17 <= n) {
        const int tc = seg[i] >> 4, th = seg[i] & 15;
        int total = 0;
        for (int k = 0; k < 16; ++k) total += seg[i + 1 + k];
        if (th > 3 || tc > 1 || total > 256 || i + 17 + total > n) fail("corrupt JPEG (DHT)");
        (tc ? ac : dc)[th].build(seg + i + 1, seg + i + 17, total);
        i += 17 + total;
      }
    } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1
      if (!comp.empty()) fail("corrupt JPEG (second frame header)");
      if (n < 6 || seg[0] != 8) fail("unsupported JPEG (sample precision)");
      height = u16(pos + 3);
      width = u16(pos + 5);
      const int nc = seg[5];
      if ((nc != 1 && nc != 3) || n != 6 + 3 * nc || width <= 0 || height <= 0)
        fail("unsupported JPEG (components)");
      if ((1 << 30) / width / nc < height) fail("unsupported JPEG (too large)");
      comp.resize((size_t)nc);
      for (int c = 0; c < nc; ++c) {
        Component& k = comp[(size_t)c];
        k.id = seg[6 + 3 * c];
        k.h = seg[7 + 3 * c] >> 4;
        k.v = seg[7 + 3 * c] & 15;
        k.tq = seg[8 + 3 * c];
        if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4 || k.tq > 3) fail("unsupported JPEG (sampling factors)");
        hmax = std::max(hmax, k.h);
        vmax = std::max(vmax, k.v);
      }
      // (hs = hmax / h below must be exact: with h = 3, hmax = 4 the upsampled row would be shorter than the
      // image and the filters read past the plane -- stb_image 2.08 accepts such streams and over-reads too;
      // no encoder writes them)
      for (const auto& k : comp)
        if (hmax % k.h != 0 || vmax % k.v != 0) fail("unsupported JPEG (sampling factors)");
      mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
      mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
      for (auto& c : comp) {
        c.stride = mcus_x * c.h * 8;
        c.rows = mcus_y * c.v * 8;
        c.width = (width * c.h + hmax - 1) / hmax;
        c.height = (height * c.v + vmax - 1) / vmax;
        c.plane.assign((size_t)c.stride * c.rows, 0);
      }
    } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      fail("unsupported JPEG (progressive / lossless / arithmetic)");
    } else if (m == 0xDD) {
      if (n != 2) fail("corrupt JPEG (DRI)");
      restart = u16(pos + 2);
    } else if (m == 0xDA) {  // SOS: one scan, interleaved (all components) or of a single component
      if (comp.empty()) fail("corrupt JPEG (SOS before SOF)");
      if (n < 1) fail("corrupt JPEG (SOS)");
      const int ns = seg[0];
      if (ns < 1 || ns > (int)comp.size() || n != 1 + 2 * ns + 3) fail("corrupt JPEG (SOS)");
      std::vector<Component*> order;
      for (int s = 0; s < ns; ++s) {
        Component* c = nullptr;
        for (auto& k : comp)
          if (k.id == seg[1 + 2 * s]) {
            c = &k;
            break;
          }
        if (!c) fail("corrupt JPEG (SOS component)");
        c->td = seg[2 + 2 * s] >> 4;
        c->ta = seg[2 + 2 * s] & 15;
        if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present || !have_q[c->tq])
          fail("corrupt JPEG (missing table)");
        order.push_back(c);
      }
      if (seg[1 + 2 * ns] != 0 || seg[3 + 2 * ns] != 0) fail("corrupt JPEG (SOS spectral selection)");
      BitReader br{data + pos + (size_t)len, data + size};
      for (auto& c : comp) c.pred = 0;
      int until_restart = restart ? restart : 0x7FFFFFFF;
      bool stop = false;
      // after every MCU: count the restart interval down; at its end the next marker must be RSTn
      // (otherwise the rest of the scan is left as it is, as the reference does: stb_image.h:2474-2479)
      auto mcu_done = [&]() {
        if (--until_restart > 0) return;
        br.reset();
        const uint8_t* p = br.p;
        while (p + 1 < br.end && !(p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF)) ++p;
        if (p + 1 >= br.end || p[1] < 0xD0 || p[1] > 0xD7) {
          stop = true;
          return;
        }
        br.p = p + 2;
        for (auto& c : comp) c.pred = 0;
        until_restart = restart ? restart : 0x7FFFFFFF;
      };
      if (ns == 1) {  // non-interleaved: the component's own blocks in raster order, one block = one MCU
        Component& c = *order[0];
        const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
        for (int by = 0; by < bh && !stop; ++by)
          for (int bx = 0; bx < bw && !stop; ++bx) {
            decode_block(br, c, c.plane.data() + (size_t)by * 8 * c.stride + (size_t)bx * 8);
            mcu_done();
          }
      } else {
        for (int my = 0; my < mcus_y && !stop; ++my)
          for (int mx = 0; mx < mcus_x && !stop; ++mx) {
            for (Component* c : order)
              for (int by = 0; by < c->v; ++by)
                for (int bx = 0; bx < c->h; ++bx)
                  decode_block(br, *c, c->plane.data() + (size_t)(my * c->v + by) * 8 * c->stride +
                                           (size_t)(mx * c->h + bx) * 8);
            mcu_done();
          }
      }
      ++scans;
      // continue behind the entropy-coded segment: the next marker that is not a restart marker
      const uint8_t* p = br.p;
      while (p + 1 < br.end && !(p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7)))
        ++p;
      if (p + 1 >= br.end) {
        done = true;  // no EOI: take what has been decoded
      } else {
        pos = (size_t)(p - data);
      }
      continue;
    }
    pos += (size_t)len;
  }
  if (!scans) fail("no image data in JPEG stream");

  RgbImage img;
  img.width = width;
  img.height = height;
  img.data.resize((size_t)width * height * 3);
  // Row by row: every component is brought to full width by the filter its sampling ratios select
  // (stb_image.h:3345-3361), walking its rows with the reference's line-stepping state machine
  // (:3370-3385): line0 / line1 = the two input rows around the output row, ystep = phase.
  struct Resample {
    int hs, vs, ystep, w_lores, ypos;
    const uint8_t *line0, *line1;
    std::vector<uint8_t> buf;
  };
  std::vector<Resample> rs(comp.size());
  for (size_t k = 0; k < comp.size(); ++k) {
    Resample& r = rs[k];
    r.hs = hmax / comp[k].h;
    r.vs = vmax / comp[k].v;
    r.ystep = r.vs >> 1;
    r.w_lores = (width + r.hs - 1) / r.hs;
    r.ypos = 0;
    r.line0 = r.line1 = comp[k].plane.data();
    r.buf.assign((size_t)width + 8 + (size_t)r.w_lores * r.hs, 0);
  }
  std::vector<const uint8_t*> row(comp.size());
  for (int y = 0; y < height; ++y) {
    for (size_t k = 0; k < comp.size(); ++k) {
      Resample& r = rs[k];
      const bool bot = r.ystep >= (r.vs >> 1);
      const uint8_t* near = bot ? r.line1 : r.line0;
      const uint8_t* far = bot ? r.line0 : r.line1;
      if (r.hs == 1 && r.vs == 1) row[k] = near;
      else if (r.hs == 1 && r.vs == 2) row[k] = up_v2(r.buf.data(), near, far, r.w_lores);
      else if (r.hs == 2 && r.vs == 1) row[k] = up_h2(r.buf.data(), near, r.w_lores);
      else if (r.hs == 2 && r.vs == 2) row[k] = up_h2v2(r.buf.data(), near, far, r.w_lores);
      else row[k] = up_nearest(r.buf.data(), near, r.w_lores, r.hs);
      if (++r.ystep >= r.vs) {
        r.ystep = 0;
        r.line0 = r.line1;
        if (++r.ypos < comp[k].height) r.line1 += comp[k].stride;
      }
    }
    uint8_t* o = &img.data[(size_t)y * width * 3];
    if (comp.size() == 1) {
      for (int x = 0; x < width; ++x, o += 3) o[0] = o[1] = o[2] = row[0][x];
      continue;
    }
    for (int x = 0; x < width; ++x, o += 3) {
      const int yf = (row[0][x] << 20) + (1 << 19);
      const int cb = row[1][x] - 128, cr = row[2][x] - 128;
      const int r = (yf + cr * kCrR) >> 20;
      const int g = (int)(yf + cr * -kCrG + (int)((uint32_t)(cb * -kCbG) & 0xFFFF0000u)) >> 20;
      const int b = (yf + cb * kCbB) >> 20;
      o[0] = clamp255(r);
      o[1] = clamp255(g);
      o[2] = clamp255(b);
    }
  }
  return img;
}

}  // namespace ratsdf
