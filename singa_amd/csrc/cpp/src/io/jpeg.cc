// Baseline JPEG codec, see include/singa/io/jpeg.h.
// The transforms are written as short fixed-count loops over contiguous rows;
// they only pay off once the compiler vectorises them.
#pragma GCC optimize("O3")
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

// Eight floats as one value: the compiler lowers the arithmetic below to the
// widest vector unit the build targets (two SSE registers by default).
typedef float V8 __attribute__((vector_size(32), aligned(4)));

inline V8 Splat(float c) { return V8{c, c, c, c, c, c, c, c}; }

// Row-column inverse transform on whole rows.  `rows` has bit v set when row
// v of the coefficient block holds a nonzero value: after quantisation most
// rows are empty and drop out of both passes.
void Idct(const float *in, unsigned rows, float *out) {  // in[v*8+u] -> out[y*8+x]
  V8 basis[8];
  std::memcpy(basis, kDct.m, sizeof(basis));
  V8 acc[8];
  for (int y = 0; y < 8; ++y) acc[y] = Splat(0.f);
  for (int v = 0; v < 8; ++v) {
    if (!((rows >> v) & 1)) continue;
    V8 t = Splat(0.f);
    for (int u = 0; u < 8; ++u) {
      const float c = in[v * 8 + u];
      if (c != 0.f) t += Splat(c) * basis[u];
    }
    for (int y = 0; y < 8; ++y) acc[y] += Splat(kDct.m[v][y]) * t;
  }
  std::memcpy(out, acc, sizeof(acc));
}

// out[i] + 128 rounded and clamped to 0..255, a row of eight at a time
inline void StoreRow8(const float *row, uint8_t *dst) {
  typedef int I8 __attribute__((vector_size(32)));
  V8 v;
  std::memcpy(&v, row, sizeof(v));
  v += Splat(128.5f);
  const V8 lo = Splat(0.f), hi = Splat(255.5f);
  v = v < lo ? lo : v;
  v = v > hi ? hi : v;
  const I8 iv = __builtin_convertvector(v, I8);
  for (int x = 0; x < 8; ++x) dst[x] = static_cast<uint8_t>(iv[x]);
}

inline uint8_t Clamp8(float v) {  // round half up on the clamped value
  return static_cast<uint8_t>(std::min(255.f, std::max(0.f, v)) + 0.5f);
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
  // codes of up to kLook bits, indexed by the next kLook bits of the stream:
  // (length << 8) | symbol, 0 = longer code
  static constexpr int kLook = 9;
  uint16_t look[1 << kLook];
  bool present = false;
  void Build(const uint8_t *bits, const uint8_t *v, int n) {
    std::memcpy(vals, v, n);
    std::memset(look, 0, sizeof(look));
    int k = 0, c = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = c;
      if (l <= kLook)
        for (int i = 0; i < bits[l - 1] && k + i < n; ++i) {
          const int first = (c + i) << (kLook - l);
          for (int f = 0; f < (1 << (kLook - l)) && first + f < (1 << kLook); ++f)
            look[first + f] = static_cast<uint16_t>((l << 8) | v[k + i]);
        }
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

// Entropy-coded bytes go through a 64-bit window (stuffed zero bytes after
// 0xff removed); the window is never filled past a marker, so at a restart
// boundary pos() is the marker itself.
class BitReader {
 public:
  BitReader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
  /// the next n <= 16 bits without consuming them, zero-padded at the end
  int Peek(int n) {
    if (cnt_ < n) Fill();
    return static_cast<int>(buf_ >> (64 - n));
  }
  /// false when fewer than n bits are left before a marker / the end
  bool Skip(int n) {
    if (cnt_ < n) {
      Fill();
      if (cnt_ < n) return false;
    }
    buf_ <<= n;
    cnt_ -= n;
    return true;
  }
  /// n <= 16 bits as an unsigned number, -1 at a marker / end of data
  int Bits(int n) {
    if (n == 0) return 0;
    const int v = Peek(n);
    return Skip(n) ? v : -1;
  }
  // Unchecked variants for the coefficient loop: after Ensure() the window
  // holds at least 32 bits (one code of up to 16 bits plus up to 15 value
  // bits) unless the data ends, in which case the missing bits read as zero
  // and Underflow() reports it afterwards.
  void Ensure() { if (cnt_ < 32) Fill(); }
  int PeekFast(int n) const { return static_cast<int>(buf_ >> (64 - n)); }
  void SkipFast(int n) { buf_ <<= n; cnt_ -= n; }
  bool Underflow() const { return cnt_ < 0; }
  void Align() {
    if (cnt_ < 0) cnt_ = 0;
    const int d = cnt_ & 7;
    buf_ <<= d;
    cnt_ -= d;
  }
  size_t pos() const { return pos_; }
  void set_pos(size_t p) { pos_ = p; cnt_ = 0; buf_ = 0; }

 private:
  void Fill() {
    while (cnt_ <= 56 && pos_ < n_) {
      const uint8_t b = p_[pos_];
      if (b == 0xff) {
        if (pos_ + 1 >= n_ || p_[pos_ + 1] != 0) break;  // a marker: stop
        pos_ += 2;
      } else {
        ++pos_;
      }
      buf_ |= static_cast<uint64_t>(b) << (56 - cnt_);
      cnt_ += 8;
    }
  }
  const uint8_t *p_;
  size_t n_, pos_ = 0;
  int cnt_ = 0;
  uint64_t buf_ = 0;
};

// after br->Ensure(); -1 = no such code
inline int DecodeSymbolFast(BitReader *br, const HuffDec &h) {
  const int peek = br->PeekFast(16);
  const int e = h.look[peek >> (16 - HuffDec::kLook)];
  if (e) {
    br->SkipFast(e >> 8);
    return e & 255;
  }
  for (int l = HuffDec::kLook + 1; l <= 16; ++l) {
    const int code = peek >> (16 - l);
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
      br->SkipFast(l);
      return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
  }
  return -1;
}

int DecodeSymbol(BitReader *br, const HuffDec &h) {
  const int peek = br->Peek(16);
  const int e = h.look[peek >> (16 - HuffDec::kLook)];
  if (e) return br->Skip(e >> 8) ? (e & 255) : -1;
  for (int l = HuffDec::kLook + 1; l <= 16; ++l) {
    const int code = peek >> (16 - l);
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l])
      return br->Skip(l) ? h.vals[h.valptr[l] + code - h.mincode[l]] : -1;
  }
  return -1;
}

// value of an s-bit field: fields below 2^(s-1) stand for negative numbers
// (branch-free: the sign of v - 2^(s-1) selects the offset)
inline int Extend(int v, int s) {
  const int m = 1 << (s - 1);
  return v + (((v - m) >> 31) & (1 - 2 * m));
}

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
  HuffDec hdc[4] = {}, hac[4] = {};   // zero tables: nothing is "present"
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
      // a malformed header must not be able to request gigabytes of planes
      if (static_cast<uint64_t>(W) * static_cast<uint64_t>(H) > (1ull << 28))
        return Fail(error, "image larger than 2^28 pixels");
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
      if (sl < 1) return Fail(error, "empty scan header");
      const int ns = s[0];
      if (ns != static_cast<int>(comps.size()) || sl < static_cast<size_t>(4 + 2 * ns))
        return Fail(error, "non-interleaved scans are not supported");
      // every frame component must be selected exactly once (duplicate
      // component ids would leave one with tables nobody checked)
      std::vector<int> selected(comps.size(), 0);
      for (int k = 0; k < ns; ++k) {
        Component *c = nullptr;
        for (size_t ci = 0; ci < comps.size(); ++ci)
          if (comps[ci].id == s[1 + 2 * k] && !selected[ci]) {
            c = &comps[ci];
            selected[ci] = 1;
            break;
          }
        if (!c) return Fail(error, "scan names an unknown component");
        c->td = s[2 + 2 * k] >> 4;
        c->ta = s[2 + 2 * k] & 15;
      }
      for (size_t ci = 0; ci < comps.size(); ++ci) {
        const Component &c = comps[ci];
        if (!selected[ci]) return Fail(error, "scan omits a component");
        if (c.td > 3 || c.ta > 3 || !hdc[c.td].present || !hac[c.ta].present ||
            !have_qt[c.tq])
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
      static const float kZero7[7] = {0, 0, 0, 0, 0, 0, 0};
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
                unsigned rows = 1;  // the DC term sits in row 0
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
                  br.Ensure();
                  const int rs = DecodeSymbolFast(&br, hac[c.ta]);
                  if (rs < 0 || br.Underflow())
                    return Fail(error, "bad AC code");
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r == 15) { k += 16; continue; }
                    break;  // end of block
                  }
                  k += r;
                  if (k > 63) return Fail(error, "AC run past block end");
                  const int bits = br.PeekFast(sz);
                  br.SkipFast(sz);
                  if (br.Underflow()) return Fail(error, "truncated scan");
                  const int zp = kZigzag[k];
                  coef[zp] = static_cast<float>(Extend(bits, sz)) * qt[c.tq][zp];
                  rows |= 1u << (zp >> 3);
                  ++k;
                }
                const int px0 = (i * c.h + bx) * 8, py0 = (j * c.v + by) * 8;
                const size_t stride = static_cast<size_t>(c.bw) * 8;
                uint8_t *dst = c.plane.data() + py0 * stride + px0;
                if (rows == 1 && !std::memcmp(coef + 1, kZero7, sizeof(kZero7))) {
                  // only the DC term: a flat block
                  const uint8_t flat = Clamp8(coef[0] * 0.125f + 128.f);
                  for (int y = 0; y < 8; ++y, dst += stride)
                    std::memset(dst, flat, 8);
                } else {
                  Idct(coef, rows, out);
                  for (int y = 0; y < 8; ++y, dst += stride)
                    StoreRow8(out + y * 8, dst);
                }
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
  // Subsampled components: bilinear interpolation between sample centres (the
  // 3/4 - 1/4 "triangle" filter for 2:1 sampling), edge samples replicated.
  // The taps of a column depend on x only and are computed once.
  struct Tap { int i0, i1; float w; };
  auto taps = [](int n_out, int n_in, int num, int den) {
    std::vector<Tap> t(n_out);
    for (int o = 0; o < n_out; ++o) {
      const float f = (o + 0.5f) * num / den - 0.5f;
      const int i = static_cast<int>(std::floor(f));
      t[o].w = f - i;
      t[o].i0 = std::min(n_in - 1, std::max(0, i));
      t[o].i1 = std::min(n_in - 1, std::max(0, i + 1));
    }
    return t;
  };
  std::vector<std::vector<Tap>> xt(C), yt(C);
  std::vector<bool> full(C);
  for (int c = 0; c < C; ++c) {
    const Component &cc = comps[c];
    full[c] = cc.h == hmax && cc.v == vmax;
    if (full[c]) continue;
    xt[c] = taps(W, (W * cc.h + hmax - 1) / hmax, cc.h, hmax);
    yt[c] = taps(H, (H * cc.v + vmax - 1) / vmax, cc.v, vmax);
  }
  std::vector<float> row(static_cast<size_t>(W) * 3);
  for (int y = 0; y < H; ++y) {
    for (int c = 0; c < C; ++c) {
      const Component &cc = comps[c];
      const size_t stride = static_cast<size_t>(cc.bw) * 8;
      float *r = row.data() + static_cast<size_t>(c) * W;
      if (full[c]) {
        const uint8_t *src = cc.plane.data() + y * stride;
        for (int x = 0; x < W; ++x) r[x] = src[x];
        continue;
      }
      const Tap &ty = yt[c][y];
      const uint8_t *r0 = cc.plane.data() + ty.i0 * stride;
      const uint8_t *r1 = cc.plane.data() + ty.i1 * stride;
      const Tap *tx = xt[c].data();
      for (int x = 0; x < W; ++x) {
        const float a = r0[tx[x].i0] + tx[x].w * (r0[tx[x].i1] - r0[tx[x].i0]);
        const float b = r1[tx[x].i0] + tx[x].w * (r1[tx[x].i1] - r1[tx[x].i0]);
        r[x] = a + ty.w * (b - a);
      }
    }
    uint8_t *o = pixels->data() + static_cast<size_t>(y) * W * C;
    if (C == 1) {
      for (int x = 0; x < W; ++x) o[x] = Clamp8(row[x]);
    } else {
      const float *Y = row.data(), *Cb = Y + W, *Cr = Cb + W;
      for (int x = 0; x < W; ++x) {
        const float cb = Cb[x] - 128.f, cr = Cr[x] - 128.f;
        o[x * 3 + 0] = Clamp8(Y[x] + 1.402f * cr);
        o[x * 3 + 1] = Clamp8(Y[x] - 0.344136f * cb - 0.714136f * cr);
        o[x * 3 + 2] = Clamp8(Y[x] + 1.772f * cb);
      }
    }
  }
  *width = W;
  *height = H;
  *channels = C;
  return true;
}

}  // namespace jpeg
}  // namespace singa
