// Baseline JPEG codec, see include/singa/io/jpeg.h.
#include "singa/io/jpeg.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace singa {
namespace jpeg {
namespace {

const int kZigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Annex K tables
const uint8_t kQuantLuma[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQuantChroma[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06,
    0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
    0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
    0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4,
    0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41,
    0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
    0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
    0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4,
    0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// cos((2x + 1) u pi / 16) * c(u) / 2
struct DctBasis {
  float m[8][8];  // [u][x]
  DctBasis() {
    for (int u = 0; u < 8; ++u)
      for (int x = 0; x < 8; ++x)
        m[u][x] = (u == 0 ? std::sqrt(0.125f) : 0.5f) *
                  std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.f);
  }
};
const DctBasis kDct;

void Fdct(const float *in, float *out) {  // in[y*8+x] -> out[v*8+u]
  float tmp[64];
  for (int y = 0; y < 8; ++y)
    for (int u = 0; u < 8; ++u) {
      float s = 0.f;
      for (int x = 0; x < 8; ++x) s += in[y * 8 + x] * kDct.m[u][x];
      tmp[y * 8 + u] = s;
    }
  for (int v = 0; v < 8; ++v)
    for (int u = 0; u < 8; ++u) {
      float s = 0.f;
      for (int y = 0; y < 8; ++y) s += tmp[y * 8 + u] * kDct.m[v][y];
      out[v * 8 + u] = s;
    }
}

void Idct(const float *in, float *out) {  // in[v*8+u] -> out[y*8+x]
  float tmp[64];
  for (int v = 0; v < 8; ++v)
    for (int x = 0; x < 8; ++x) {
      float s = 0.f;
      for (int u = 0; u < 8; ++u) s += in[v * 8 + u] * kDct.m[u][x];
      tmp[v * 8 + x] = s;
    }
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      float s = 0.f;
      for (int v = 0; v < 8; ++v) s += tmp[v * 8 + x] * kDct.m[v][y];
      out[y * 8 + x] = s;
    }
}

inline uint8_t Clamp8(float v) {
  return static_cast<uint8_t>(std::min(255.f, std::max(0.f, std::round(v))));
}

// ---- Huffman code built from a (bits, vals) table specification ------------
struct HuffEnc {
  uint16_t code[256];
  uint8_t len[256];
  HuffEnc(const uint8_t *bits, const uint8_t *vals) {
    std::memset(len, 0, sizeof(len));
    std::memset(code, 0, sizeof(code));
    int k = 0;
    unsigned c = 0;
    for (int l = 1; l <= 16; ++l) {
      for (int i = 0; i < bits[l - 1]; ++i, ++k) {
        code[vals[k]] = static_cast<uint16_t>(c++);
        len[vals[k]] = static_cast<uint8_t>(l);
      }
      c <<= 1;
    }
  }
};

struct HuffDec {
  // canonical decoding: per length the smallest code, its index, and count
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
  void Build(const uint8_t *bits, const uint8_t *v, int n) {
    std::memcpy(vals, v, n);
    int k = 0, c = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = c;
      c += bits[l - 1];
      k += bits[l - 1];
      maxcode[l] = bits[l - 1] ? c - 1 : -1;
      c <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
  }
};

class BitWriter {
 public:
  explicit BitWriter(std::string *out) : out_(out) {}
  void Put(unsigned code, int len) {
    acc_ = (acc_ << len) | (code & ((1u << len) - 1));
    n_ += len;
    while (n_ >= 8) {
      uint8_t b = static_cast<uint8_t>(acc_ >> (n_ - 8));
      out_->push_back(static_cast<char>(b));
      if (b == 0xff) out_->push_back('\0');  // byte stuffing
      n_ -= 8;
    }
  }
  void Flush() {
    if (n_ > 0) Put(0x7f, 8 - n_);  // pad with ones
  }

 private:
  std::string *out_;
  uint64_t acc_ = 0;
  int n_ = 0;
};

inline int BitSize(int v) {
  v = v < 0 ? -v : v;
  int n = 0;
  while (v) { ++n; v >>= 1; }
  return n;
}

void PutMarker(std::string *o, int m) {
  o->push_back(static_cast<char>(0xff));
  o->push_back(static_cast<char>(m));
}
void Put16(std::string *o, int v) {
  o->push_back(static_cast<char>((v >> 8) & 0xff));
  o->push_back(static_cast<char>(v & 0xff));
}

void EncodeBlock(const float *coef, const uint16_t *q, int *pred,
                 const HuffEnc &dc, const HuffEnc &ac, BitWriter *bw) {
  int zz[64];
  for (int i = 0; i < 64; ++i) {
    const int pos = kZigzag[i];
    zz[i] = static_cast<int>(std::lround(coef[pos] / q[pos]));
  }
  int diff = zz[0] - *pred;
  *pred = zz[0];
  int s = BitSize(diff);
  bw->Put(dc.code[s], dc.len[s]);
  if (s) bw->Put(diff < 0 ? diff - 1 + (1 << s) : diff, s);
  int run = 0;
  for (int i = 1; i < 64; ++i) {
    if (zz[i] == 0) { ++run; continue; }
    while (run > 15) { bw->Put(ac.code[0xf0], ac.len[0xf0]); run -= 16; }
    s = BitSize(zz[i]);
    const int sym = (run << 4) | s;
    bw->Put(ac.code[sym], ac.len[sym]);
    bw->Put(zz[i] < 0 ? zz[i] - 1 + (1 << s) : zz[i], s);
    run = 0;
  }
  if (run) bw->Put(ac.code[0x00], ac.len[0x00]);
}

}  // namespace

std::string Encode(const uint8_t *px, int W, int H, int C, int quality,
                   const std::string &comment) {
  std::string o;
  if (W <= 0 || H <= 0 || (C != 1 && C != 3) || W > 65535 || H > 65535)
    return o;
  quality = std::min(100, std::max(1, quality));
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  uint16_t q[2][64];
  for (int i = 0; i < 64; ++i) {
    q[0][i] = static_cast<uint16_t>(
        std::min(255, std::max(1, (kQuantLuma[i] * scale + 50) / 100)));
    q[1][i] = static_cast<uint16_t>(
        std::min(255, std::max(1, (kQuantChroma[i] * scale + 50) / 100)));
  }
  PutMarker(&o, 0xd8);
  // JFIF APP0
  PutMarker(&o, 0xe0);
  Put16(&o, 16);
  o.append("JFIF", 5);
  o.push_back(1); o.push_back(1); o.push_back(0);
  Put16(&o, 1); Put16(&o, 1);
  o.push_back(0); o.push_back(0);
  if (!comment.empty()) {
    PutMarker(&o, 0xfe);
    Put16(&o, static_cast<int>(comment.size()) + 2);
    o += comment;
  }
  for (int t = 0; t < (C == 3 ? 2 : 1); ++t) {
    PutMarker(&o, 0xdb);
    Put16(&o, 67);
    o.push_back(static_cast<char>(t));
    for (int i = 0; i < 64; ++i)
      o.push_back(static_cast<char>(q[t][kZigzag[i]]));
  }
  PutMarker(&o, 0xc0);
  Put16(&o, 8 + 3 * C);
  o.push_back(8);
  Put16(&o, H); Put16(&o, W);
  o.push_back(static_cast<char>(C));
  for (int c = 0; c < C; ++c) {
    o.push_back(static_cast<char>(c + 1));
    o.push_back(0x11);
    o.push_back(static_cast<char>(c == 0 ? 0 : 1));
  }
  auto dht = [&](int cls, int id, const uint8_t *bits, const uint8_t *vals,
                 int n) {
    PutMarker(&o, 0xc4);
    Put16(&o, 19 + n);
    o.push_back(static_cast<char>((cls << 4) | id));
    o.append(reinterpret_cast<const char *>(bits), 16);
    o.append(reinterpret_cast<const char *>(vals), n);
  };
  dht(0, 0, kDcLumaBits, kDcVals, 12);
  dht(1, 0, kAcLumaBits, kAcLumaVals, 162);
  if (C == 3) {
    dht(0, 1, kDcChromaBits, kDcVals, 12);
    dht(1, 1, kAcChromaBits, kAcChromaVals, 162);
  }
  PutMarker(&o, 0xda);
  Put16(&o, 6 + 2 * C);
  o.push_back(static_cast<char>(C));
  for (int c = 0; c < C; ++c) {
    o.push_back(static_cast<char>(c + 1));
    o.push_back(static_cast<char>(c == 0 ? 0x00 : 0x11));
  }
  o.push_back(0); o.push_back(63); o.push_back(0);

  const HuffEnc dcl(kDcLumaBits, kDcVals), acl(kAcLumaBits, kAcLumaVals);
  const HuffEnc dcc(kDcChromaBits, kDcVals), acc(kAcChromaBits, kAcChromaVals);
  BitWriter bw(&o);
  int pred[3] = {0, 0, 0};
  float blk[3][64], coef[64];
  for (int by = 0; by < H; by += 8)
    for (int bx = 0; bx < W; bx += 8) {
      for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
          const int sy = std::min(H - 1, by + y), sx = std::min(W - 1, bx + x);
          const uint8_t *p = px + (static_cast<size_t>(sy) * W + sx) * C;
          if (C == 1) {
            blk[0][y * 8 + x] = p[0] - 128.f;
          } else {
            const float r = p[0], g = p[1], b = p[2];
            blk[0][y * 8 + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.f;
            blk[1][y * 8 + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
            blk[2][y * 8 + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
          }
        }
      for (int c = 0; c < C; ++c) {
        Fdct(blk[c], coef);
        EncodeBlock(coef, q[c == 0 ? 0 : 1], &pred[c], c == 0 ? dcl : dcc,
                    c == 0 ? acl : acc, &bw);
      }
    }
  bw.Flush();
  PutMarker(&o, 0xd9);
  return o;
}

namespace {

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int bw = 0, bh = 0;        // blocks per row / column (padded to MCUs)
  std::vector<uint8_t> plane;  // bw*8 x bh*8
  int pred = 0;
};

class BitReader {
 public:
  BitReader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
  // returns -1 at a marker / end of data
  int Bit() {
    if (cnt_ == 0) {
      if (pos_ >= n_) return -1;
      uint8_t b = p_[pos_];
      if (b == 0xff) {
        if (pos_ + 1 >= n_) return -1;
        if (p_[pos_ + 1] == 0) pos_ += 2;
        else return -1;  // a marker: stop
      } else {
        ++pos_;
      }
      cur_ = b;
      cnt_ = 8;
    }
    --cnt_;
    return (cur_ >> cnt_) & 1;
  }
  int Bits(int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) {
      const int b = Bit();
      if (b < 0) return -1;
      v = (v << 1) | b;
    }
    return v;
  }
  void Align() { cnt_ = 0; }
  size_t pos() const { return pos_; }
  void set_pos(size_t p) { pos_ = p; cnt_ = 0; }

 private:
  const uint8_t *p_;
  size_t n_, pos_ = 0;
  int cnt_ = 0;
  uint8_t cur_ = 0;
};

int DecodeSymbol(BitReader *br, const HuffDec &h) {
  int code = 0;
  for (int l = 1; l <= 16; ++l) {
    const int b = br->Bit();
    if (b < 0) return -1;
    code = (code << 1) | b;
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l])
      return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}

inline int Extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

bool Fail(std::string *error, const char *why) {
  if (error) *error = why;
  return false;
}

}  // namespace

bool Decode(const std::string &data, std::vector<uint8_t> *pixels, int *width,
            int *height, int *channels, std::string *comment,
            std::string *error) {
  const uint8_t *p = reinterpret_cast<const uint8_t *>(data.data());
  const size_t n = data.size();
  if (n < 4 || p[0] != 0xff || p[1] != 0xd8) return Fail(error, "no SOI marker");
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false};
  HuffDec hdc[4], hac[4];
  std::vector<Component> comps;
  int W = 0, H = 0, restart = 0, hmax = 1, vmax = 1;
  size_t pos = 2;
  bool decoded = false;
  while (pos + 4 <= n && !decoded) {
    if (p[pos] != 0xff) return Fail(error, "marker expected");
    int m = p[pos + 1];
    pos += 2;
    if (m == 0xff) { --pos; continue; }  // fill byte
    if (m == 0xd9) break;
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (pos + 2 > n) return Fail(error, "truncated segment");
    const size_t len = (p[pos] << 8) | p[pos + 1];
    if (len < 2 || pos + len > n) return Fail(error, "bad segment length");
    const uint8_t *s = p + pos + 2;
    const size_t sl = len - 2;
    if (m == 0xfe) {
      if (comment) comment->assign(reinterpret_cast<const char *>(s), sl);
    } else if (m == 0xdb) {
      size_t i = 0;
      while (i < sl) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        ++i;
        if (tq > 3 || i + (pq ? 128 : 64) > sl) return Fail(error, "bad DQT");
        for (int k = 0; k < 64; ++k) {
          qt[tq][kZigzag[k]] = pq ? static_cast<uint16_t>((s[i] << 8) | s[i + 1])
                                  : s[i];
          i += pq ? 2 : 1;
        }
        have_qt[tq] = true;
      }
    } else if (m == 0xc4) {
      size_t i = 0;
      while (i + 17 <= sl) {
        const int cls = s[i] >> 4, id = s[i] & 15;
        int cnt = 0;
        for (int k = 0; k < 16; ++k) cnt += s[i + 1 + k];
        if (id > 3 || cls > 1 || cnt > 256 || i + 17 + cnt > sl)
          return Fail(error, "bad DHT");
        (cls ? hac[id] : hdc[id]).Build(s + i + 1, s + i + 17, cnt);
        i += 17 + cnt;
      }
    } else if (m == 0xc0 || m == 0xc1) {
      if (sl < 6 || s[0] != 8) return Fail(error, "only 8-bit samples");
      H = (s[1] << 8) | s[2];
      W = (s[3] << 8) | s[4];
      const int nc = s[5];
      if ((nc != 1 && nc != 3) || sl < static_cast<size_t>(6 + 3 * nc) ||
          W == 0 || H == 0)
        return Fail(error, "unsupported frame header");
      comps.resize(nc);
      for (int c = 0; c < nc; ++c) {
        comps[c].id = s[6 + 3 * c];
        comps[c].h = s[7 + 3 * c] >> 4;
        comps[c].v = s[7 + 3 * c] & 15;
        comps[c].tq = s[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 ||
            comps[c].v > 4 || comps[c].tq > 3)
          return Fail(error, "bad sampling factors");
        hmax = std::max(hmax, comps[c].h);
        vmax = std::max(vmax, comps[c].v);
      }
    } else if (m >= 0xc2 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return Fail(error, "only baseline sequential JPEG is supported");
    } else if (m == 0xdd) {
      if (sl < 2) return Fail(error, "bad DRI");
      restart = (s[0] << 8) | s[1];
    } else if (m == 0xda) {
      if (comps.empty()) return Fail(error, "scan before frame header");
      const int ns = s[0];
      if (ns != static_cast<int>(comps.size()) || sl < static_cast<size_t>(4 + 2 * ns))
        return Fail(error, "non-interleaved scans are not supported");
      for (int k = 0; k < ns; ++k) {
        Component *c = nullptr;
        for (auto &cc : comps)
          if (cc.id == s[1 + 2 * k]) c = &cc;
        if (!c) return Fail(error, "scan names an unknown component");
        c->td = s[2 + 2 * k] >> 4;
        c->ta = s[2 + 2 * k] & 15;
        if (c->td > 3 || c->ta > 3 || !hdc[c->td].present ||
            !hac[c->ta].present || !have_qt[c->tq])
          return Fail(error, "scan refers to a missing table");
      }
      if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }
      const int mcuw = 8 * hmax, mcuh = 8 * vmax;
      const int mx = (W + mcuw - 1) / mcuw, my = (H + mcuh - 1) / mcuh;
      for (auto &c : comps) {
        c.bw = mx * c.h;
        c.bh = my * c.v;
        c.plane.assign(static_cast<size_t>(c.bw) * 8 * c.bh * 8, 0);
        c.pred = 0;
      }
      BitReader br(p + pos + len, n - pos - len);
      int until_restart = restart;
      float coef[64], out[64];
      for (int j = 0; j < my; ++j)
        for (int i = 0; i < mx; ++i) {
          if (restart && until_restart == 0) {
            // expect an RSTn marker at the next byte boundary
            br.Align();
            size_t q = br.pos();
            const uint8_t *d = p + pos + len;
            const size_t dn = n - pos - len;
            while (q + 1 < dn && !(d[q] == 0xff && d[q + 1] >= 0xd0 &&
                                   d[q + 1] <= 0xd7))
              ++q;
            if (q + 1 >= dn) return Fail(error, "missing restart marker");
            br.set_pos(q + 2);
            for (auto &c : comps) c.pred = 0;
            until_restart = restart;
          }
          for (auto &c : comps)
            for (int by = 0; by < c.v; ++by)
              for (int bx = 0; bx < c.h; ++bx) {
                std::memset(coef, 0, sizeof(coef));
                int t = DecodeSymbol(&br, hdc[c.td]);
                if (t < 0 || t > 11) return Fail(error, "bad DC code");
                int diff = 0;
                if (t) {
                  const int bits = br.Bits(t);
                  if (bits < 0) return Fail(error, "truncated scan");
                  diff = Extend(bits, t);
                }
                c.pred += diff;
                coef[0] = static_cast<float>(c.pred) * qt[c.tq][0];
                for (int k = 1; k < 64;) {
                  const int rs = DecodeSymbol(&br, hac[c.ta]);
                  if (rs < 0) return Fail(error, "bad AC code");
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r == 15) { k += 16; continue; }
                    break;  // end of block
                  }
                  k += r;
                  if (k > 63) return Fail(error, "AC run past block end");
                  const int bits = br.Bits(sz);
                  if (bits < 0) return Fail(error, "truncated scan");
                  const int zp = kZigzag[k];
                  coef[zp] = static_cast<float>(Extend(bits, sz)) * qt[c.tq][zp];
                  ++k;
                }
                Idct(coef, out);
                const int px0 = (i * c.h + bx) * 8, py0 = (j * c.v + by) * 8;
                const int stride = c.bw * 8;
                for (int y = 0; y < 8; ++y)
                  for (int x = 0; x < 8; ++x)
                    c.plane[static_cast<size_t>(py0 + y) * stride + px0 + x] =
                        Clamp8(out[y * 8 + x] + 128.f);
              }
          if (restart) --until_restart;
        }
      decoded = true;
    }
    pos += len;
  }
  if (!decoded) return Fail(error, "no scan found");
  const int C = static_cast<int>(comps.size());
  pixels->assign(static_cast<size_t>(W) * H * C, 0);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float v[3] = {0.f, 128.f, 128.f};
      for (int c = 0; c < C; ++c) {
        const Component &cc = comps[c];
        const int stride = cc.bw * 8;
        if (cc.h == hmax && cc.v == vmax) {
          v[c] = cc.plane[static_cast<size_t>(y) * stride + x];
          continue;
        }
        // subsampled component: bilinear interpolation between sample
        // centres (the 3/4 - 1/4 "triangle" filter for 2:1 sampling), edge
        // samples replicated
        const int cw = (W * cc.h + hmax - 1) / hmax;
        const int ch = (H * cc.v + vmax - 1) / vmax;
        const float fx = (x + 0.5f) * cc.h / hmax - 0.5f;
        const float fy = (y + 0.5f) * cc.v / vmax - 0.5f;
        const int x0 = static_cast<int>(std::floor(fx));
        const int y0 = static_cast<int>(std::floor(fy));
        const float wx = fx - x0, wy = fy - y0;
        auto at = [&](int yy, int xx) {
          yy = std::min(ch - 1, std::max(0, yy));
          xx = std::min(cw - 1, std::max(0, xx));
          return static_cast<float>(cc.plane[static_cast<size_t>(yy) * stride + xx]);
        };
        v[c] = (1 - wy) * ((1 - wx) * at(y0, x0) + wx * at(y0, x0 + 1)) +
               wy * ((1 - wx) * at(y0 + 1, x0) + wx * at(y0 + 1, x0 + 1));
      }
      uint8_t *o = pixels->data() + (static_cast<size_t>(y) * W + x) * C;
      if (C == 1) {
        o[0] = Clamp8(v[0]);
      } else {
        const float cb = v[1] - 128.f, cr = v[2] - 128.f;
        o[0] = Clamp8(v[0] + 1.402f * cr);
        o[1] = Clamp8(v[0] - 0.344136f * cb - 0.714136f * cr);
        o[2] = Clamp8(v[0] + 1.772f * cb);
      }
    }
  *width = W;
  *height = H;
  *channels = C;
  return true;
}

}  // namespace jpeg
}  // namespace singa
