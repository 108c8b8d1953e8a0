// CIFAR-10 binary-version reader for the C++ examples: data_batch_{1..5}.bin
// and test_batch.bin under a directory; each record is 1 label byte followed
// by 3072 pixel bytes (CHW).  When the files are absent a synthetic set with
// the same shapes is generated (class-dependent colour blobs + noise) so the
// programs run offline.  Counterpart of the reference's
// examples/cpp/cifar10/cifar10.h.
#pragma once
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "singa/core/tensor.h"

namespace singa {

class Cifar10 {
 public:
  explicit Cifar10(std::string dir_path, bool normalize = true)
      : dir_path_(std::move(dir_path)), normalize_(normalize) {}

  static const size_t kImageSize = 32, kImageVol = 3072, kBatchSize = 10000;
  static const size_t kTrainFiles = 5;

  /// (images (N,3,32,32) float32, labels (N) int32)
  std::pair<Tensor, Tensor> ReadTrainData(size_t synthetic_n = 2048) {
    std::vector<float> x;
    std::vector<int> y;
    bool ok = true;
    for (size_t i = 1; i <= kTrainFiles && ok; ++i)
      ok = ReadFile(dir_path_ + "/data_batch_" + std::to_string(i) + ".bin", &x, &y);
    if (!ok) Synthetic(synthetic_n, 1, &x, &y);
    return Pack(x, y);
  }
  std::pair<Tensor, Tensor> ReadTestData(size_t synthetic_n = 512) {
    std::vector<float> x;
    std::vector<int> y;
    if (!ReadFile(dir_path_ + "/test_batch.bin", &x, &y))
      Synthetic(synthetic_n, 2, &x, &y);
    return Pack(x, y);
  }

 private:
  bool ReadFile(const std::string &file, std::vector<float> *x,
                std::vector<int> *y) {
    std::ifstream in(file, std::ios::in | std::ios::binary);
    if (!in.is_open()) return false;
    std::vector<unsigned char> rec(1 + kImageVol);
    for (size_t i = 0; i < kBatchSize; ++i) {
      in.read(reinterpret_cast<char *>(rec.data()), rec.size());
      if (!in) return i > 0;
      y->push_back(rec[0]);
      for (size_t k = 0; k < kImageVol; ++k) x->push_back(rec[1 + k]);
    }
    return true;
  }
  void Synthetic(size_t n, unsigned seed, std::vector<float> *x,
                 std::vector<int> *y) {
    std::printf("CIFAR-10 binaries not found under %s: synthetic data\n",
                dir_path_.c_str());
    x->clear();
    y->clear();
    std::mt19937 g(seed);
    std::mt19937 tg(99);  // class templates are shared by train and test
    std::vector<float> templ(10 * 3 * 16);
    for (auto &t : templ) t = std::uniform_real_distribution<float>(40, 215)(tg);
    std::normal_distribution<float> noise(0.f, 25.f);
    for (size_t i = 0; i < n; ++i) {
      int label = g() % 10;
      y->push_back(label);
      for (size_t c = 0; c < 3; ++c)
        for (size_t h = 0; h < 32; ++h)
          for (size_t w = 0; w < 32; ++w) {
            float base = templ[(label * 3 + c) * 16 + (h / 8) * 4 + w / 8];
            x->push_back(std::min(255.f, std::max(0.f, base + noise(g))));
          }
    }
  }
  std::pair<Tensor, Tensor> Pack(std::vector<float> &x, std::vector<int> &y) {
    if (normalize_)
      for (auto &v : x) v = v / 255.f - 0.5f;
    size_t n = y.size();
    Tensor tx(Shape{n, 3, kImageSize, kImageSize}), ty(Shape{n}, kInt);
    tx.CopyDataFromHostPtr(x.data(), x.size());
    ty.CopyDataFromHostPtr(y.data(), y.size());
    return {tx, ty};
  }
  std::string dir_path_;
  bool normalize_;
};

}  // namespace singa
