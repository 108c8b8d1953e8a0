// ImageNet (ILSVRC12) data pipeline of the C++ AlexNet example.
//
// Creation: every image of a list ("<path> <label>" per line) is decoded,
// resized to 256x256, re-encoded as JPEG with its label and appended to record
// files of `file_size` images each (train1.bin, train2.bin, ...; test.bin);
// the per-pixel mean of the training images goes to mean.bin.  Without the
// dataset, CreateSyntheticData writes the same files from generated images, so
// the whole JPEG / record-file / prefetch path runs offline.
//
// Loading: LoadData reads the next `read_size` records of a file and has
// `nthreads` workers decode them, subtract the mean and crop 227x227 (random
// position + random mirror when training, centre crop otherwise) straight
// into the batch; AsyncLoadData does that on a thread of its own so that the
// next file is prepared while the network trains on the current one.
//
// Counterpart of the reference's examples/cpp/imagenet/ilsvrc12.h.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "singa/io/jpeg.h"
#include "singa/singa.h"

namespace singa {

class ILSVRC {
 public:
  explicit ILSVRC(size_t image_size = 256, size_t crop_size = 227)
      : image_size_(image_size), crop_size_(crop_size) {
    CHECK_LE(crop_size, image_size);
    EncoderConf ec;
    ec.type = "jpg";  // plain JPEG files, the label in a comment segment
    ec.image_dim_order = "CHW";
    encoder_.Setup(ec);
  }

  size_t image_size() const { return image_size_; }
  size_t crop_size() const { return crop_size_; }

  /// Returns the number of record files written.
  size_t CreateTrainData(const std::string &image_list,
                         const std::string &image_folder,
                         const std::string &bin_folder, size_t file_size) {
    std::vector<std::pair<std::string, int>> items = ReadList(image_list);
    // one pass in a fixed random order, so that every file mixes the classes
    std::mt19937 g(1);
    std::shuffle(items.begin(), items.end(), g);
    std::vector<double> sum(3 * image_size_ * image_size_, 0.0);
    size_t files = 0, in_file = 0, total = 0;
    io::BinFileWriter writer;
    for (auto &it : items) {
      Tensor img = ReadImage(image_folder + "/" + it.first);
      if (img.empty()) continue;
      if (in_file == 0) {
        writer.Close();
        CHECK(writer.Open(TrainFile(bin_folder, ++files), io::kCreate));
      }
      Accumulate(img, &sum);
      WriteRecord(&writer, img, it.second, total);
      ++total;
      if (++in_file == file_size) in_file = 0;
    }
    writer.Close();
    CHECK_GT(total, 0u) << "no readable image in " << image_list;
    std::vector<float> mean(sum.size());
    for (size_t i = 0; i < sum.size(); ++i) mean[i] = float(sum[i] / total);
    WriteMean(mean, bin_folder + "/mean.bin");
    return files;
  }

  size_t CreateTestData(const std::string &image_list,
                        const std::string &image_folder,
                        const std::string &bin_folder) {
    io::BinFileWriter writer;
    CHECK(writer.Open(bin_folder + "/test.bin", io::kCreate));
    size_t total = 0;
    for (auto &it : ReadList(image_list)) {
      Tensor img = ReadImage(image_folder + "/" + it.first);
      if (img.empty()) continue;
      WriteRecord(&writer, img, it.second, total++);
    }
    writer.Close();
    return total;
  }

  /// Class-dependent colour layouts plus noise, written like real data.
  /// Returns the number of training files.
  size_t CreateSyntheticData(const std::string &bin_folder, size_t num_train,
                             size_t num_test, size_t file_size, int classes,
                             unsigned seed = 3) {
    const size_t S = image_size_;
    std::mt19937 tg(17);
    std::vector<float> templ(size_t(classes) * 3 * 16);
    for (auto &t : templ) t = std::uniform_real_distribution<float>(30, 225)(tg);
    std::mt19937 g(seed);
    std::normal_distribution<float> noise(0.f, 12.f);
    auto make = [&](int label) {
      std::vector<float> px(3 * S * S);
      for (size_t c = 0; c < 3; ++c)
        for (size_t h = 0; h < S; ++h)
          for (size_t w = 0; w < S; ++w) {
            float base = templ[(size_t(label) * 3 + c) * 16 + (h * 4 / S) * 4 +
                               w * 4 / S];
            px[(c * S + h) * S + w] =
                std::min(255.f, std::max(0.f, base + noise(g)));
          }
      Tensor img(Shape{3, S, S});
      img.CopyDataFromHostPtr(px.data(), px.size());
      return img;
    };
    std::vector<double> sum(3 * S * S, 0.0);
    io::BinFileWriter writer;
    size_t files = 0;
    for (size_t i = 0; i < num_train; ++i) {
      if (i % file_size == 0) {
        writer.Close();
        CHECK(writer.Open(TrainFile(bin_folder, ++files), io::kCreate));
      }
      int label = static_cast<int>(g() % classes);
      Tensor img = make(label);
      Accumulate(img, &sum);
      WriteRecord(&writer, img, label, i);
    }
    writer.Close();
    std::vector<float> mean(sum.size());
    for (size_t i = 0; i < sum.size(); ++i)
      mean[i] = float(sum[i] / std::max<size_t>(num_train, 1));
    WriteMean(mean, bin_folder + "/mean.bin");
    CHECK(writer.Open(bin_folder + "/test.bin", io::kCreate));
    for (size_t i = 0; i < num_test; ++i) {
      int label = static_cast<int>(g() % classes);
      Tensor img = make(label);
      WriteRecord(&writer, img, label, i);
    }
    writer.Close();
    return files;
  }

  static std::string TrainFile(const std::string &bin_folder, size_t index) {
    return bin_folder + "/train" + std::to_string(index) + ".bin";
  }

  void ReadMean(const std::string &path) {
    io::BinFileReader reader;
    CHECK(reader.Open(path)) << "cannot open " << path;
    std::string key, value;
    CHECK(reader.Read(&key, &value));
    CHECK_EQ(value.size(), 3 * image_size_ * image_size_ * sizeof(float))
        << "mean image of another size";
    mean_.resize(value.size() / sizeof(float));
    std::memcpy(mean_.data(), value.data(), value.size());
  }

  /// Reads up to `read_size` records of `file`, continuing where the previous
  /// call on the same file stopped (a different file starts from its
  /// beginning), and fills x (n,3,crop,crop) and y (n) on the host.  Returns
  /// and stores in *n_read the number of images read; 0 = end of file.
  size_t LoadData(int flag, const std::string &file, size_t read_size,
                  Tensor *x, Tensor *y, size_t *n_read, int nthreads) {
    CHECK(!mean_.empty()) << "ReadMean first";
    if (file != open_file_) {
      reader_.reset(new io::BinFileReader());
      CHECK(reader_->Open(file)) << "cannot open " << file;
      open_file_ = file;
    }
    std::vector<std::string> records;
    std::string key, value;
    while (records.size() < read_size && reader_->Read(&key, &value))
      records.push_back(std::move(value));
    if (records.size() < read_size) open_file_.clear();  // exhausted
    const size_t n = records.size();
    *n_read = n;
    if (n == 0) return 0;
    const size_t vol = 3 * crop_size_ * crop_size_;
    std::vector<float> xs(n * vol);
    std::vector<int> ys(n);
    nthreads = std::max(1, std::min<int>(nthreads, static_cast<int>(n)));
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
      pool.emplace_back([&, t] {
        for (size_t i = t; i < n; i += nthreads)
          DecodeTransform(flag, records[i], xs.data() + i * vol, &ys[i]);
      });
    for (auto &t : pool) t.join();
    x->Resize(Shape{n, 3, crop_size_, crop_size_});
    y->Resize(Shape{n});
    x->CopyDataFromHostPtr(xs.data(), xs.size());
    y->CopyDataFromHostPtr(ys.data(), ys.size());
    return n;
  }

  std::thread AsyncLoadData(int flag, const std::string &file,
                            size_t read_size, Tensor *x, Tensor *y,
                            size_t *n_read, int nthreads) {
    return std::thread([=] {
      LoadData(flag, file, read_size, x, y, n_read, nthreads);
    });
  }

 private:
  static std::vector<std::pair<std::string, int>> ReadList(
      const std::string &path) {
    std::ifstream in(path);
    CHECK(in.is_open()) << "cannot open " << path;
    std::vector<std::pair<std::string, int>> out;
    std::string line;
    while (std::getline(in, line)) {
      std::istringstream ss(line);
      std::string name;
      int label;
      if (ss >> name >> label) out.emplace_back(name, label);
    }
    return out;
  }

  /// A JPEG file as a (3, image_size, image_size) float tensor; an empty
  /// tensor when the file is missing or not a baseline JPEG.
  Tensor ReadImage(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) {
      LOG(WARNING) << "cannot open " << path;
      return Tensor();
    }
    std::string bytes((std::istreambuf_iterator<char>(in)),
                      std::istreambuf_iterator<char>());
    std::vector<uint8_t> px;
    int w = 0, h = 0, c = 0;
    std::string err;
    if (!jpeg::Decode(bytes, &px, &w, &h, &c, nullptr, &err)) {
      LOG(WARNING) << path << ": " << err;
      return Tensor();
    }
    std::vector<float> chw(size_t(3) * h * w);
    for (int k = 0; k < 3; ++k)  // grey images fill all three planes
      for (int i = 0; i < h * w; ++i)
        chw[size_t(k) * h * w + i] = px[size_t(i) * c + (c == 3 ? k : 0)];
    Tensor img(Shape{3, size_t(h), size_t(w)});
    img.CopyDataFromHostPtr(chw.data(), chw.size());
    if (size_t(h) != image_size_ || size_t(w) != image_size_)
      img = resize(img, image_size_, image_size_, "CHW");
    return img;
  }

  void Accumulate(const Tensor &img, std::vector<double> *sum) {
    auto v = img.ToVector<float>();
    for (size_t i = 0; i < v.size(); ++i) (*sum)[i] += v[i];
  }

  void WriteRecord(io::BinFileWriter *writer, const Tensor &img, int label,
                   size_t index) {
    Tensor lab(Shape{1}, kInt);
    lab.CopyDataFromHostPtr(&label, 1);
    std::vector<Tensor> rec{img, lab};
    char key[16];
    std::snprintf(key, sizeof(key), "%08zu", index);
    CHECK(writer->Write(key, encoder_.Encode(rec)));
  }

  void WriteMean(const std::vector<float> &mean, const std::string &path) {
    io::BinFileWriter writer;
    CHECK(writer.Open(path, io::kCreate));
    std::string value(reinterpret_cast<const char *>(mean.data()),
                      mean.size() * sizeof(float));
    CHECK(writer.Write("mean", value));
    writer.Close();
    mean_ = mean;
  }

  // Runs on the worker threads.  One pass from the decoder's HWC bytes to the
  // CHW float crop: subtract the mean, cut the window (random position and
  // random mirror when training, centred otherwise: the policy of
  // ImageTransformer::Apply) and write straight into the batch.
  void DecodeTransform(int flag, const std::string &record, float *out,
                       int *label) {
    std::vector<uint8_t> px;
    int w = 0, h = 0, c = 0;
    std::string comment, err;
    CHECK(jpeg::Decode(record, &px, &w, &h, &c, &comment, &err)) << err;
    CHECK(size_t(w) == image_size_ && size_t(h) == image_size_ && c == 3)
        << "record of another geometry";
    CHECK_EQ(comment.rfind("label ", 0), 0u) << "record without a label";
    *label = std::atoi(comment.c_str() + 6);
    static thread_local std::mt19937 rng(std::random_device{}());
    const size_t S = image_size_, K = crop_size_;
    size_t oh = (S - K) / 2, ow = (S - K) / 2;
    bool flip = false;
    if (flag & kTrain) {
      oh = std::uniform_int_distribution<size_t>(0, S - K)(rng);
      ow = std::uniform_int_distribution<size_t>(0, S - K)(rng);
      flip = rng() & 1;
    }
    for (size_t ch = 0; ch < 3; ++ch)
      for (size_t y = 0; y < K; ++y) {
        const uint8_t *src = px.data() + ((oh + y) * S + ow) * 3 + ch;
        const float *m = mean_.data() + (ch * S + oh + y) * S + ow;
        float *dst = out + (ch * K + y) * K;
        if (flip)
          for (size_t x = 0; x < K; ++x) dst[K - 1 - x] = src[x * 3] - m[x];
        else
          for (size_t x = 0; x < K; ++x) dst[x] = src[x * 3] - m[x];
      }
  }

  size_t image_size_, crop_size_;
  std::vector<float> mean_;
  JPGEncoder encoder_;
  std::unique_ptr<io::BinFileReader> reader_;
  std::string open_file_;
};

}  // namespace singa
