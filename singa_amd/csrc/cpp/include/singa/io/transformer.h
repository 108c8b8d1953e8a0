// Image transformations for data feeding on the host: resize (bilinear),
// crop (random in training, centred in evaluation), mirror.
// API parity: reference include/singa/io/transformer.h, src/io/
// image_transformer.cc (ImageTransformer::Apply, resize / crop / mirror).
#pragma once
#include <string>

#include "../core/tensor.h"
#include "../model/conf.h"

namespace singa {

class Transformer {
 public:
  virtual ~Transformer() {}
  virtual void Setup(const TransformerConf &conf) {}
  virtual Tensor Apply(int flag, Tensor &input) = 0;
};

class ImageTransformer : public Transformer {
 public:
  void Setup(const TransformerConf &conf) override { conf_ = conf; }
  /// `input` is one image (3-d) or a grey image (2-d) on the host.
  Tensor Apply(int flag, Tensor &input) override;
  const TransformerConf &conf() const { return conf_; }

 private:
  TransformerConf conf_;
};

Tensor resize(Tensor &input, size_t resize_height, size_t resize_width,
              const std::string &image_dim_order);
Tensor crop(Tensor &input, size_t crop_height, size_t crop_width,
            size_t crop_h_offset, size_t crop_w_offset,
            const std::string &image_dim_order);
Tensor mirror(Tensor &input, bool horizontal_mirror, bool vertical_mirror,
              const std::string &image_dim_order);

}  // namespace singa
