// jpeg.cc -- baseline / extended-sequential Huffman JPEG decoder for the colour frames of ScanNet
// .sens streams (third_party/scannet/sensorData.hpp: TYPE_JPEG; the reference decodes them with the
// vendored stb_image).
//
// The decoder restates the DEFAULT decoding path of the IJG library (libjpeg 6b and libjpeg-turbo,
// README / jidctint.c "accurate integer" inverse DCT with 13-bit constants, jdsample.c "fancy"
// triangle-filter chroma upsampling for 2:1 horizontal and 2:1 x 2:1 subsampling, jdcolor.c fixed-
// point YCbCr -> RGB), so its output is bit-identical to that library's -- which is what the tests
// pin it against (tests/test_sens_reader.py, via PIL).  stb_image uses another IDCT and upsampler:
// colours can differ from the reference's by a few LSB there; that part of the parity is unpinned
// (no ScanNet data and no stb_image here).
// Supported: 8-bit precision, 1 or 3 components, sampling factors 1 or 2, restart intervals.
// Not supported (never produced by ScanNet's recorder): progressive, arithmetic coding, 12-bit, CMYK.
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
  int blocks_w = 0, blocks_h = 0;   // allocated blocks (whole MCUs)
  int width = 0, height = 0;        // downsampled size = ceil(image * samp / max_samp)
  int pred = 0;
  std::vector<uint8_t> plane;       // blocks_w * 8 wide
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

// jidctint.c, jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2
constexpr int CB = 13, P1 = 2;
constexpr long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270,
               F_0_899976223 = 7373, F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137,
               F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
inline long descale(long x, int n) { return (x + (1L << (n - 1))) >> n; }
inline uint8_t clamp255(long x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void idct_islow(const int* coef, const uint16_t* quant, uint8_t* out, int stride) {
  long ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* in = coef + c;
    const uint16_t* q = quant + c;
    long* w = ws + c;
    if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
      const long dc = ((long)in[0] * q[0]) << P1;
      for (int r = 0; r < 8; ++r) w[8 * r] = dc;
      continue;
    }
    long z2 = (long)in[16] * q[16], z3 = (long)in[48] * q[48];
    long z1 = (z2 + z3) * F_0_541196100;
    long tmp2 = z1 + z3 * (-F_1_847759065);
    long tmp3 = z1 + z2 * F_0_765366865;
    z2 = (long)in[0] * q[0];
    z3 = (long)in[32] * q[32];
    long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = (long)in[56] * q[56];
    tmp1 = (long)in[40] * q[40];
    tmp2 = (long)in[24] * q[24];
    tmp3 = (long)in[8] * q[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336;
    tmp1 *= F_2_053119869;
    tmp2 *= F_3_072711026;
    tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    w[0] = descale(tmp10 + tmp3, CB - P1);
    w[56] = descale(tmp10 - tmp3, CB - P1);
    w[8] = descale(tmp11 + tmp2, CB - P1);
    w[48] = descale(tmp11 - tmp2, CB - P1);
    w[16] = descale(tmp12 + tmp1, CB - P1);
    w[40] = descale(tmp12 - tmp1, CB - P1);
    w[24] = descale(tmp13 + tmp0, CB - P1);
    w[32] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; ++r) {
    const long* w = ws + 8 * r;
    uint8_t* o = out + (size_t)r * stride;
    if (!w[1] && !w[2] && !w[3] && !w[4] && !w[5] && !w[6] && !w[7]) {
      const uint8_t dc = clamp255(descale(w[0], P1 + 3) + 128);
      for (int c = 0; c < 8; ++c) o[c] = dc;
      continue;
    }
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F_0_541196100;
    long tmp2 = z1 + z3 * (-F_1_847759065);
    long tmp3 = z1 + z2 * F_0_765366865;
    long tmp0 = (w[0] + w[4]) << CB, tmp1 = (w[0] - w[4]) << CB;
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7];
    tmp1 = w[5];
    tmp2 = w[3];
    tmp3 = w[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336;
    tmp1 *= F_2_053119869;
    tmp2 *= F_3_072711026;
    tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int S = CB + P1 + 3;
    o[0] = clamp255(descale(tmp10 + tmp3, S) + 128);
    o[7] = clamp255(descale(tmp10 - tmp3, S) + 128);
    o[1] = clamp255(descale(tmp11 + tmp2, S) + 128);
    o[6] = clamp255(descale(tmp11 - tmp2, S) + 128);
    o[2] = clamp255(descale(tmp12 + tmp1, S) + 128);
    o[5] = clamp255(descale(tmp12 - tmp1, S) + 128);
    o[3] = clamp255(descale(tmp13 + tmp0, S) + 128);
    o[4] = clamp255(descale(tmp13 - tmp0, S) + 128);
  }
}

// jdsample.c: h2v1_fancy_upsample on one row of n input samples -> 2n output samples
void fancy_h2v1(const uint8_t* in, int n, uint8_t* out) {
  if (n == 1) {
    out[0] = out[1] = in[0];
    return;
  }
  out[0] = in[0];
  out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
  for (int i = 1; i < n - 1; ++i) {
    const int v = in[i] * 3;
    out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
    out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
  }
  out[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
  out[2 * n - 1] = in[n - 1];
}
// jdsample.c: h2v2_fancy_upsample, one output row from the nearer (in0) and the farther (in1) input row
void fancy_h2v2_row(const uint8_t* in0, const uint8_t* in1, int n, uint8_t* out) {
  if (n == 1) {
    const int s = in0[0] * 3 + in1[0];
    out[0] = (uint8_t)((s * 4 + 8) >> 4);
    out[1] = (uint8_t)((s * 4 + 7) >> 4);
    return;
  }
  int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
  out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
  out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
  lastcol = thiscol;
  thiscol = nextcol;
  for (int i = 1; i < n - 1; ++i) {
    nextcol = in0[i + 1] * 3 + in1[i + 1];
    out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
    out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol;
    thiscol = nextcol;
  }
  out[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
  out[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
}

}  // namespace

RgbImage decode_jpeg(const uint8_t* data, size_t size, const std::string& name) {
  auto fail = [&](const char* what) -> void { throw std::runtime_error(name + ": " + what); };
  if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) fail("not a JPEG stream");
  uint16_t quant[4][64] = {};
  bool have_q[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<Component> comp;
  int width = 0, height = 0, hmax = 1, vmax = 1, restart = 0;
  bool decoded = false;
  size_t pos = 2;
  auto u16 = [&](size_t p) { return (int)data[p] << 8 | data[p + 1]; };
  while (pos + 4 <= size && !decoded) {
    if (data[pos] != 0xFF) fail("corrupt JPEG (marker expected)");
    while (pos < size && data[pos] == 0xFF) ++pos;
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
        if (tq > 3 || i + (pq ? 128 : 64) > n) fail("corrupt JPEG (DQT)");
        for (int k = 0; k < 64; ++k) {
          quant[tq][kZigZag[k]] = pq ? (uint16_t)(seg[i] << 8 | seg[i + 1]) : seg[i];
          i += pq ? 2 : 1;
        }
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
      if (n < 6 || seg[0] != 8) fail("unsupported JPEG (sample precision)");
      height = u16(pos + 3);
      width = u16(pos + 5);
      const int nc = seg[5];
      if ((nc != 1 && nc != 3) || n < 6 + 3 * nc || width <= 0 || height <= 0)
        fail("unsupported JPEG (components)");
      comp.resize((size_t)nc);
      for (int c = 0; c < nc; ++c) {
        comp[(size_t)c].id = seg[6 + 3 * c];
        comp[(size_t)c].h = seg[7 + 3 * c] >> 4;
        comp[(size_t)c].v = seg[7 + 3 * c] & 15;
        comp[(size_t)c].tq = seg[8 + 3 * c];
        if (comp[(size_t)c].h < 1 || comp[(size_t)c].h > 2 || comp[(size_t)c].v < 1 ||
            comp[(size_t)c].v > 2 || comp[(size_t)c].tq > 3)
          fail("unsupported JPEG (sampling factors)");
        hmax = std::max(hmax, comp[(size_t)c].h);
        vmax = std::max(vmax, comp[(size_t)c].v);
      }
    } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      fail("unsupported JPEG (progressive / lossless / arithmetic)");
    } else if (m == 0xDD) {
      if (n < 2) fail("corrupt JPEG (DRI)");
      restart = u16(pos + 2);
    } else if (m == 0xDA) {  // SOS: the only scan of a sequential file with interleaved components
      if (comp.empty()) fail("corrupt JPEG (SOS before SOF)");
      const int ns = seg[0];
      if (ns != (int)comp.size() || n < 1 + 2 * ns + 3) fail("unsupported JPEG (non-interleaved scans)");
      for (int s = 0; s < ns; ++s) {
        Component* c = nullptr;
        for (auto& k : comp)
          if (k.id == seg[1 + 2 * s]) c = &k;
        if (!c) fail("corrupt JPEG (SOS component)");
        c->td = seg[2 + 2 * s] >> 4;
        c->ta = seg[2 + 2 * s] & 15;
        if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present || !have_q[c->tq])
          fail("corrupt JPEG (missing table)");
      }
      const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
      const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
      for (auto& c : comp) {
        c.blocks_w = mcus_x * c.h;
        c.blocks_h = mcus_y * c.v;
        c.width = (width * c.h + hmax - 1) / hmax;
        c.height = (height * c.v + vmax - 1) / vmax;
        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
        c.pred = 0;
      }
      BitReader br{data + pos + (size_t)len, data + size};
      int coef[64];
      int until_restart = restart;
      for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
          if (restart && until_restart == 0) {  // RSTn: byte-align, skip the marker, reset predictors
            br.reset();
            while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
            if (br.p + 1 >= br.end) fail("corrupt JPEG (restart marker missing)");
            br.p += 2;
            for (auto& c : comp) c.pred = 0;
            until_restart = restart;
          }
          for (auto& c : comp)
            for (int by = 0; by < c.v; ++by)
              for (int bx = 0; bx < c.h; ++bx) {
                std::memset(coef, 0, sizeof(coef));
                const int t = decode_symbol(br, dc[c.td], name);
                if (t > 11) fail("corrupt JPEG (DC size)");
                c.pred += t ? extend(br.get(t), t) : 0;
                coef[0] = c.pred;
                for (int k = 1; k < 64;) {
                  const int rs = decode_symbol(br, ac[c.ta], name);
                  const int r = rs >> 4, s = rs & 15;
                  if (s == 0) {
                    if (r != 15) break;  // EOB
                    k += 16;             // ZRL
                    continue;
                  }
                  k += r;
                  if (k > 63) fail("corrupt JPEG (AC run)");
                  coef[kZigZag[k]] = extend(br.get(s), s);
                  ++k;
                }
                const size_t stride = (size_t)c.blocks_w * 8;
                uint8_t* out = c.plane.data() + ((size_t)(my * c.v + by) * 8) * stride +
                               (size_t)(mx * c.h + bx) * 8;
                idct_islow(coef, quant[c.tq], out, (int)stride);
              }
          if (restart) --until_restart;
        }
      decoded = true;
    }
    pos += (size_t)len;
  }
  if (!decoded) fail("no image data in JPEG stream");

  RgbImage img;
  img.width = width;
  img.height = height;
  img.data.resize((size_t)width * height * 3);
  // full-resolution planes (fancy upsampling of subsampled components)
  std::vector<std::vector<uint8_t>> full(comp.size());
  for (size_t ci = 0; ci < comp.size(); ++ci) {
    Component& c = comp[ci];
    const size_t stride = (size_t)c.blocks_w * 8;
    std::vector<uint8_t>& f = full[ci];
    const int fw = width + 2, fh = height + 2;  // room for the odd last column / row
    f.assign((size_t)fw * fh, 0);
    const int hx = hmax / c.h, vy = vmax / c.v;
    // jdsample.c (jinit_upsampler): the triangle filters are only used for components more than two
    // samples wide; narrower ones are replicated
    const bool fancy = c.width > 2;
    if (hx == 1 && vy == 1) {
      for (int y = 0; y < height; ++y) std::memcpy(&f[(size_t)y * fw], &c.plane[(size_t)y * stride], (size_t)width);
    } else if (!fancy && hx == 2) {
      for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) f[(size_t)y * fw + x] = c.plane[(size_t)(y / vy) * stride + x / 2];
    } else if (hx == 2 && vy == 1) {
      for (int y = 0; y < height; ++y) {
        std::vector<uint8_t> row((size_t)c.width * 2);
        fancy_h2v1(&c.plane[(size_t)y * stride], c.width, row.data());
        std::memcpy(&f[(size_t)y * fw], row.data(), (size_t)std::min(width, c.width * 2));
      }
    } else if (hx == 2 && vy == 2) {
      std::vector<uint8_t> row((size_t)c.width * 2);
      for (int y = 0; y < height; ++y) {
        const int sy = y >> 1;
        const int oy = (y & 1) ? std::min(sy + 1, c.height - 1) : std::max(sy - 1, 0);
        fancy_h2v2_row(&c.plane[(size_t)sy * stride], &c.plane[(size_t)oy * stride], c.width, row.data());
        std::memcpy(&f[(size_t)y * fw], row.data(), (size_t)std::min(width, c.width * 2));
      }
    } else {
      fail("unsupported JPEG (chroma subsampling other than 4:4:4, 4:2:2, 4:2:0)");
    }
  }
  const int fw = width + 2;
  if (comp.size() == 1) {
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        const uint8_t g = full[0][(size_t)y * fw + x];
        uint8_t* o = &img.data[((size_t)y * width + x) * 3];
        o[0] = o[1] = o[2] = g;
      }
    return img;
  }
  // jdcolor.c: build_ycc_rgb_table / ycc_rgb_convert (SCALEBITS 16)
  int cr_r[256], cb_b[256];
  long cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) {
    const long x = i - 128;
    cr_r[i] = (int)((91881L * x + 32768L) >> 16);    // FIX(1.40200)
    cb_b[i] = (int)((116130L * x + 32768L) >> 16);   // FIX(1.77200)
    cr_g[i] = -46802L * x;                           // FIX(0.71414)
    cb_g[i] = -22554L * x + 32768L;                  // FIX(0.34414)
  }
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const int Y = full[0][(size_t)y * fw + x], cb = full[1][(size_t)y * fw + x], cr = full[2][(size_t)y * fw + x];
      uint8_t* o = &img.data[((size_t)y * width + x) * 3];
      o[0] = clamp255(Y + cr_r[cr]);
      o[1] = clamp255(Y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
      o[2] = clamp255(Y + cb_b[cb]);
    }
  return img;
}

}  // namespace ratsdf
