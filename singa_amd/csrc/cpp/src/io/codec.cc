#include <cmath>
#include <cstdlib>
#include <random>
#include <sstream>

#include "singa/io/encoder.h"
#include "singa/io/jpeg.h"
#include "singa/io/transformer.h"
#include "singa/model/conf_proto.h"
#include "singa/utils/string.h"

namespace singa {

// ---- CSV -------------------------------------------------------------------------
std::string CSVEncoder::Encode(std::vector<Tensor> &data) {
  CHECK_GE(data.size(), 1u);
  Tensor feat = data[0].Clone(defaultDevice);
  std::ostringstream os;
  if (data.size() == 2) {
    Tensor lab = data[1].Clone(defaultDevice);
    os << lab.ToVector<int>().at(0) << ',';
  }
  os.precision(9);
  if (feat.data_type() == kInt) {
    auto v = feat.ToVector<int>();
    for (size_t i = 0; i < v.size(); ++i) os << (i ? "," : "") << v[i];
  } else {
    auto v = feat.ToVector<float>();
    for (size_t i = 0; i < v.size(); ++i) os << (i ? "," : "") << v[i];
  }
  return os.str();
}

std::vector<Tensor> CSVDecoder::Decode(std::string value) {
  std::vector<Tensor> out;
  std::vector<float> vals;
  std::stringstream ss(value);
  std::string item;
  int label = 0;
  bool first = true;
  while (std::getline(ss, item, ',')) {
    if (first && has_label_) {
      label = std::atoi(item.c_str());
    } else {
      vals.push_back(std::strtof(item.c_str(), nullptr));
    }
    first = false;
  }
  Tensor feat(Shape{vals.size()}, kFloat32);
  feat.CopyDataFromHostPtr(vals.data(), vals.size());
  out.push_back(feat);
  if (has_label_) {
    Tensor lab(Shape{1}, kInt);
    lab.CopyDataFromHostPtr(&label, 1);
    out.push_back(lab);
  }
  return out;
}

// ---- PPM / PGM image records ---------------------------------------------------------
std::string ImageEncoder::Encode(std::vector<Tensor> &data) {
  CHECK_GE(data.size(), 1u);
  Tensor img = data[0].Clone(defaultDevice);
  CHECK(img.nDim() == 2u || img.nDim() == 3u);
  size_t H, W, C;
  bool chw = order_ == "CHW" && img.nDim() == 3;
  if (img.nDim() == 2) { H = img.shape(0); W = img.shape(1); C = 1; }
  else if (chw) { C = img.shape(0); H = img.shape(1); W = img.shape(2); }
  else { H = img.shape(0); W = img.shape(1); C = img.shape(2); }
  CHECK(C == 1 || C == 3) << "1 or 3 channels";
  std::vector<float> px;
  if (img.data_type() == kFloat32) {
    px = img.ToVector<float>();
  } else {
    auto v = img.ToVector<int>();
    px.assign(v.begin(), v.end());
  }
  std::ostringstream os;
  os << (C == 3 ? "P6\n" : "P5\n");
  if (data.size() == 2)
    os << "# label " << data[1].Clone(defaultDevice).ToVector<int>().at(0) << "\n";
  os << W << ' ' << H << "\n255\n";
  std::string body(H * W * C, '\0');
  for (size_t h = 0; h < H; ++h)
    for (size_t w = 0; w < W; ++w)
      for (size_t c = 0; c < C; ++c) {
        float v = chw ? px[(c * H + h) * W + w] : px[(h * W + w) * C + c];
        body[(h * W + w) * C + c] =
            static_cast<char>(static_cast<unsigned char>(
                std::min(255.f, std::max(0.f, std::round(v)))));
      }
  return os.str() + body;
}

std::vector<Tensor> ImageDecoder::Decode(std::string value) {
  std::vector<Tensor> out;
  size_t pos = 0;
  auto token = [&](bool *is_label, int *label) {
    // next whitespace-delimited token, consuming comment lines
    while (pos < value.size()) {
      char ch = value[pos];
      if (ch == '#') {
        size_t e = value.find('\n', pos);
        std::string line = value.substr(pos, e - pos);
        if (line.rfind("# label ", 0) == 0) {
          *is_label = true;
          *label = std::atoi(line.c_str() + 8);
        }
        pos = e + 1;
      } else if (std::isspace(static_cast<unsigned char>(ch))) {
        ++pos;
      } else {
        break;
      }
    }
    size_t b = pos;
    while (pos < value.size() &&
           !std::isspace(static_cast<unsigned char>(value[pos])))
      ++pos;
    return value.substr(b, pos - b);
  };
  bool has_label = false;
  int label = 0;
  std::string magic = token(&has_label, &label);
  CHECK(magic == "P6" || magic == "P5") << "not a binary PPM/PGM record";
  size_t C = magic == "P6" ? 3 : 1;
  size_t W = std::stoul(token(&has_label, &label));
  size_t H = std::stoul(token(&has_label, &label));
  int maxv = std::stoi(token(&has_label, &label));
  CHECK_EQ(maxv, 255);
  ++pos;  // single whitespace after maxval
  CHECK_GE(value.size() - pos, H * W * C) << "truncated image record";
  const unsigned char *p =
      reinterpret_cast<const unsigned char *>(value.data()) + pos;
  bool chw = order_ == "CHW";
  std::vector<float> px(H * W * C);
  for (size_t h = 0; h < H; ++h)
    for (size_t w = 0; w < W; ++w)
      for (size_t c = 0; c < C; ++c) {
        float v = p[(h * W + w) * C + c];
        if (chw) px[(c * H + h) * W + w] = v;
        else px[(h * W + w) * C + c] = v;
      }
  Tensor img(chw ? Shape{C, H, W} : Shape{H, W, C}, kFloat32);
  img.CopyDataFromHostPtr(px.data(), px.size());
  out.push_back(img);
  if (has_label) {
    Tensor lab(Shape{1}, kInt);
    lab.CopyDataFromHostPtr(&label, 1);
    out.push_back(lab);
  }
  return out;
}

// ---- JPEG image records ------------------------------------------------------
std::string JPGEncoder::Encode(std::vector<Tensor> &data) {
  CHECK_GE(data.size(), 1u);
  Tensor img = data[0].Clone(defaultDevice);
  CHECK(img.nDim() == 2u || img.nDim() == 3u);
  size_t H, W, C;
  bool chw = order_ == "CHW" && img.nDim() == 3;
  if (img.nDim() == 2) { H = img.shape(0); W = img.shape(1); C = 1; }
  else if (chw) { C = img.shape(0); H = img.shape(1); W = img.shape(2); }
  else { H = img.shape(0); W = img.shape(1); C = img.shape(2); }
  CHECK(C == 1 || C == 3) << "1 or 3 channels";
  std::vector<float> px;
  if (img.data_type() == kFloat32) {
    px = img.ToVector<float>();
  } else {
    auto v = img.ToVector<int>();
    px.assign(v.begin(), v.end());
  }
  std::vector<uint8_t> hwc(H * W * C);
  for (size_t h = 0; h < H; ++h)
    for (size_t w = 0; w < W; ++w)
      for (size_t c = 0; c < C; ++c) {
        float v = chw ? px[(c * H + h) * W + w] : px[(h * W + w) * C + c];
        hwc[(h * W + w) * C + c] = static_cast<uint8_t>(
            std::min(255.f, std::max(0.f, std::round(v))));
      }
  int label = 0;
  if (data.size() == 2)
    label = data[1].Clone(defaultDevice).ToVector<int>().at(0);
  if (!proto_) {
    std::string comment;
    if (data.size() == 2) comment = "label " + std::to_string(label);
    return jpeg::Encode(hwc.data(), static_cast<int>(W), static_cast<int>(H),
                        static_cast<int>(C), quality_, comment);
  }
  // the reference's record: channel 0 is the file's blue plane
  if (C == 3)
    for (size_t i = 0; i < H * W; ++i) std::swap(hwc[3 * i], hwc[3 * i + 2]);
  ImageRecord record;
  for (size_t d = 0; d < img.nDim(); ++d)
    record.shape.push_back(static_cast<int>(img.shape(d)));
  if (data.size() == 2) record.label.push_back(label);
  record.pixel = jpeg::Encode(hwc.data(), static_cast<int>(W),
                              static_cast<int>(H), static_cast<int>(C),
                              quality_, "");
  return SerializeAsString(record);
}

std::vector<Tensor> JPGDecoder::Decode(std::string value) {
  std::vector<uint8_t> px8;
  int W = 0, H = 0, C = 0;
  std::string comment, error;
  const bool file = value.size() >= 2 && (unsigned char)value[0] == 0xff &&
                    (unsigned char)value[1] == 0xd8;
  ImageRecord record;
  if (!file)
    CHECK(ParseFromString(value, &record) && !record.pixel.empty())
        << "neither a JPEG file nor an image record";
  CHECK(jpeg::Decode(file ? value : record.pixel, &px8, &W, &H, &C, &comment,
                     &error))
      << "JPEG record: " << error;
  if (!file) {
    if (C == 3)
      for (size_t i = 0; i < px8.size() / 3; ++i)
        std::swap(px8[3 * i], px8[3 * i + 2]);
    comment = record.label.empty()
                  ? std::string()
                  : "label " + std::to_string(record.label[0]);
  }
  const bool chw = order_ == "CHW";
  std::vector<float> px(px8.size());
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w)
      for (int c = 0; c < C; ++c) {
        const float v = px8[(static_cast<size_t>(h) * W + w) * C + c];
        if (chw) px[(static_cast<size_t>(c) * H + h) * W + w] = v;
        else px[(static_cast<size_t>(h) * W + w) * C + c] = v;
      }
  const size_t uC = C, uH = H, uW = W;
  Tensor img(chw ? Shape{uC, uH, uW} : Shape{uH, uW, uC}, kFloat32);
  img.CopyDataFromHostPtr(px.data(), px.size());
  std::vector<Tensor> out{img};
  if (comment.rfind("label ", 0) == 0) {
    int label = std::atoi(comment.c_str() + 6);
    Tensor lab(Shape{1}, kInt);
    lab.CopyDataFromHostPtr(&label, 1);
    out.push_back(lab);
  }
  return out;
}

std::shared_ptr<Encoder> CreateEncoder(const std::string &type) {
  std::string t = ToLowerCase(type);
  if (t == "csv") return std::make_shared<CSVEncoder>();
  if (t == "image" || t == "ppm") return std::make_shared<ImageEncoder>();
  if (t == "jpg" || t == "jpeg" || t == "jpg2proto") {
    auto enc = std::make_shared<JPGEncoder>();
    EncoderConf conf;
    conf.type = t;
    enc->Setup(conf);
    return enc;
  }
  LOG(FATAL) << "Unknown encoder type " << type;
  return nullptr;
}

std::shared_ptr<Decoder> CreateDecoder(const std::string &type) {
  std::string t = ToLowerCase(type);
  if (t == "csv") return std::make_shared<CSVDecoder>();
  if (t == "image" || t == "ppm") return std::make_shared<ImageDecoder>();
  if (t == "jpg" || t == "jpeg" || t == "proto2jpg")
    return std::make_shared<JPGDecoder>();
  LOG(FATAL) << "Unknown decoder type " << type;
  return nullptr;
}

// ---- image transformer ---------------------------------------------------------------
namespace {
struct View {
  size_t C, H, W;
  bool chw, grey;
};
View MakeView(const Tensor &t, const std::string &order) {
  View v;
  v.grey = t.nDim() == 2;
  v.chw = order == "CHW";
  CHECK(order == "CHW" || order == "HWC") << "Unknown dimension order";
  if (v.grey) { v.C = 1; v.H = t.shape(0); v.W = t.shape(1); }
  else if (v.chw) { v.C = t.shape(0); v.H = t.shape(1); v.W = t.shape(2); }
  else { v.H = t.shape(0); v.W = t.shape(1); v.C = t.shape(2); }
  return v;
}
inline size_t At(const View &v, size_t c, size_t h, size_t w, size_t H,
                 size_t W) {
  return (v.chw || v.grey) ? (c * H + h) * W + w : (h * W + w) * v.C + c;
}
Shape OutShape(const View &v, size_t H, size_t W) {
  if (v.grey) return Shape{H, W};
  return v.chw ? Shape{v.C, H, W} : Shape{H, W, v.C};
}
}  // namespace

Tensor resize(Tensor &input, size_t rh, size_t rw, const std::string &order) {
  CHECK(input.device()->lang() == kCpp) << "image transforms run on the host";
  View v = MakeView(input, order);
  auto in = input.ToVector<float>();
  std::vector<float> out(v.C * rh * rw);
  const float sy = float(v.H) / rh, sx = float(v.W) / rw;
  for (size_t y = 0; y < rh; ++y) {
    float fy = std::max(0.f, (y + 0.5f) * sy - 0.5f);
    size_t y0 = std::min<size_t>(size_t(fy), v.H - 1), y1 = std::min(y0 + 1, v.H - 1);
    float wy = fy - y0;
    for (size_t x = 0; x < rw; ++x) {
      float fx = std::max(0.f, (x + 0.5f) * sx - 0.5f);
      size_t x0 = std::min<size_t>(size_t(fx), v.W - 1), x1 = std::min(x0 + 1, v.W - 1);
      float wx = fx - x0;
      for (size_t c = 0; c < v.C; ++c) {
        float a = in[At(v, c, y0, x0, v.H, v.W)], b = in[At(v, c, y0, x1, v.H, v.W)];
        float d = in[At(v, c, y1, x0, v.H, v.W)], e = in[At(v, c, y1, x1, v.H, v.W)];
        out[At(v, c, y, x, rh, rw)] =
            (a * (1 - wx) + b * wx) * (1 - wy) + (d * (1 - wx) + e * wx) * wy;
      }
    }
  }
  Tensor t(OutShape(v, rh, rw), kFloat32);
  t.CopyDataFromHostPtr(out.data(), out.size());
  return t;
}

Tensor crop(Tensor &input, size_t ch, size_t cw, size_t oh, size_t ow,
            const std::string &order) {
  View v = MakeView(input, order);
  CHECK_LE(ch + oh, v.H);
  CHECK_LE(cw + ow, v.W);
  auto in = input.ToVector<float>();
  std::vector<float> out(v.C * ch * cw);
  for (size_t c = 0; c < v.C; ++c)
    for (size_t y = 0; y < ch; ++y)
      for (size_t x = 0; x < cw; ++x)
        out[At(v, c, y, x, ch, cw)] = in[At(v, c, y + oh, x + ow, v.H, v.W)];
  Tensor t(OutShape(v, ch, cw), kFloat32);
  t.CopyDataFromHostPtr(out.data(), out.size());
  return t;
}

Tensor mirror(Tensor &input, bool horizontal, bool vertical,
              const std::string &order) {
  View v = MakeView(input, order);
  if (!horizontal && !vertical) return input;
  auto in = input.ToVector<float>();
  std::vector<float> out(in.size());
  for (size_t c = 0; c < v.C; ++c)
    for (size_t y = 0; y < v.H; ++y)
      for (size_t x = 0; x < v.W; ++x)
        out[At(v, c, y, x, v.H, v.W)] =
            in[At(v, c, vertical ? v.H - 1 - y : y,
                  horizontal ? v.W - 1 - x : x, v.H, v.W)];
  Tensor t(input.shape(), kFloat32);
  t.CopyDataFromHostPtr(out.data(), out.size());
  return t;
}

Tensor ImageTransformer::Apply(int flag, Tensor &input) {
  CHECK_LE(input.nDim(), 4u);
  CHECK_GE(input.nDim(), 2u);
  CHECK_EQ(input.data_type(), kFloat32) << "Data type " << input.data_type()
                                        << " is invalid for an raw image";
  static thread_local std::mt19937 rng(std::random_device{}());
  Tensor out = input;
  const std::string &order = conf_.image_dim_order;
  if (conf_.resize_height && conf_.resize_width)
    out = resize(out, conf_.resize_height, conf_.resize_width, order);
  if (conf_.crop_height && conf_.crop_width) {
    View v = MakeView(out, order);
    size_t oh, ow;
    if (flag & kTrain) {
      oh = std::uniform_int_distribution<size_t>(0, v.H - conf_.crop_height)(rng);
      ow = std::uniform_int_distribution<size_t>(0, v.W - conf_.crop_width)(rng);
    } else {
      oh = (v.H - conf_.crop_height) / 2;
      ow = (v.W - conf_.crop_width) / 2;
    }
    out = crop(out, conf_.crop_height, conf_.crop_width, oh, ow, order);
  }
  if (flag & kTrain) {
    bool hm = conf_.horizontal_mirror && (rng() & 1);
    bool vm = conf_.vertical_mirror && (rng() & 1);
    out = mirror(out, hm, vm, order);
  }
  if (conf_.rescale != 0.f) out = out * conf_.rescale;
  return out;
}

}  // namespace singa
