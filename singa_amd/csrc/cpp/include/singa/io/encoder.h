// Record encoders / decoders: turn (features, label) tensors into the byte
// strings stored in record files and back.  CSV keeps one text line per
// sample; the image codec stores 8-bit images as binary PPM / PGM with the
// label in a comment line; the JPG codec stores baseline JPEG files with the
// label in a COM segment, or the reference's ImageRecord message around such
// a file (own codec, io/jpeg.h: no image library is linked, where the
// reference's JPGEncoder/Decoder need OpenCV).  API parity: reference
// include/singa/io/{encoder,decoder}.h.
#pragma once
#include <string>
#include <vector>

#include "../core/tensor.h"
#include "../model/conf.h"

namespace singa {

class Encoder {
 public:
  virtual ~Encoder() {}
  virtual void Setup(const EncoderConf &conf) {}
  /// data[0] = features, optional data[1] = one-element int label.
  virtual std::string Encode(std::vector<Tensor> &data) = 0;
};

class Decoder {
 public:
  virtual ~Decoder() {}
  virtual void Setup(const DecoderConf &conf) {}
  virtual std::vector<Tensor> Decode(std::string value) = 0;
};

class CSVEncoder : public Encoder {
 public:
  std::string Encode(std::vector<Tensor> &data) override;
};

class CSVDecoder : public Decoder {
 public:
  void Setup(const DecoderConf &conf) override { has_label_ = conf.has_label; }
  std::vector<Tensor> Decode(std::string value) override;
  bool has_label() const { return has_label_; }

 private:
  bool has_label_ = true;
};

/// uint8 image tensor (HWC or CHW, 1 or 3 channels) <-> PPM/PGM bytes.
class ImageEncoder : public Encoder {
 public:
  void Setup(const EncoderConf &conf) override { order_ = conf.image_dim_order; }
  std::string Encode(std::vector<Tensor> &data) override;

 private:
  std::string order_ = "HWC";
};

class ImageDecoder : public Decoder {
 public:
  void Setup(const DecoderConf &conf) override { order_ = conf.image_dim_order; }
  /// -> {float32 image in the configured order, int label (if present)}
  std::vector<Tensor> Decode(std::string value) override;

 private:
  std::string order_ = "CHW";
};

/// uint8 image tensor (HWC or CHW, 1 or 3 channels) <-> baseline JPEG bytes
/// (reference JPGEncoder / JPGDecoder, src/io/jpg_{encoder,decoder}.cc).
/// Two record formats:
///   * "jpg2proto" (EncoderConf's default type): the reference's record, an
///     ImageRecord message (src/proto/io.proto:55) holding the tensor's
///     extents, the label and the JPEG file.  The reference hands the tensor's
///     channels to OpenCV as blue, green, red, so channel 0 of a colour image
///     is the file's blue plane; files written by either side read back the
///     same tensors on the other.
///   * "jpg": the JPEG file itself, channel 0 red, the label in a comment
///     segment; such a record opens in any image viewer.
/// The decoder tells the two apart by the JPEG start marker.
class JPGEncoder : public Encoder {
 public:
  void Setup(const EncoderConf &conf) override {
    order_ = conf.image_dim_order;
    proto_ = conf.type != "jpg" && conf.type != "jpeg";
  }
  std::string Encode(std::vector<Tensor> &data) override;
  void set_quality(int q) { quality_ = q; }

 private:
  std::string order_ = "HWC";
  bool proto_ = true;
  int quality_ = 100;  // the reference encodes at quality 100
};

class JPGDecoder : public Decoder {
 public:
  void Setup(const DecoderConf &conf) override { order_ = conf.image_dim_order; }
  /// -> {float32 image in the configured order, int label (if present)}
  std::vector<Tensor> Decode(std::string value) override;

 private:
  std::string order_ = "CHW";
};

std::shared_ptr<Encoder> CreateEncoder(const std::string &type);
std::shared_ptr<Decoder> CreateDecoder(const std::string &type);

}  // namespace singa
