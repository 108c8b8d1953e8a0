// Converts ImageNet image lists into the record files that singa_alexnet
// trains from (see ilsvrc12.h):
//   singa_ilsvrc12 -trainlist train.txt -trainfolder train/ \
//                  -testlist val.txt -testfolder val/ -outdata bin/ -filesize 1280
// or, without the dataset,
//   singa_ilsvrc12 -synthetic -outdata bin/ -ntrain 512 -ntest 128 -classes 10
// Images must be baseline JPEG files.  Counterpart of the reference's
// examples/cpp/imagenet/ilsvrc12.cc.
#include <cstdlib>
#include <string>

#include "ilsvrc12.h"

int main(int argc, char **argv) {
  singa::LogToStderr();
  auto arg = [&](const char *name, const std::string &dflt) {
    int pos = singa::ArgPos(argc, argv, name);
    return pos != -1 && pos + 1 < argc ? std::string(argv[pos + 1]) : dflt;
  };
  std::string out = arg("-outdata", "imagenet_data");
  size_t file_size = std::stoul(arg("-filesize", "1280"));
  size_t size = std::stoul(arg("-size", "256"));
  singa::ILSVRC data(size, size * 227 / 256);
  if (singa::ArgPos(argc, argv, "-synthetic") != -1) {
    size_t files = data.CreateSyntheticData(
        out, std::stoul(arg("-ntrain", "512")), std::stoul(arg("-ntest", "128")),
        file_size, std::atoi(arg("-classes", "10").c_str()));
    std::printf("wrote %zu training files, test.bin and mean.bin to %s\n",
                files, out.c_str());
    return 0;
  }
  std::string train_list = arg("-trainlist", "imagenet/label/train.txt");
  std::string train_folder = arg("-trainfolder", "imagenet/ILSVRC2012_img_train");
  std::string test_list = arg("-testlist", "imagenet/label/val.txt");
  std::string test_folder = arg("-testfolder", "imagenet/ILSVRC2012_img_val");
  size_t files = data.CreateTrainData(train_list, train_folder, out, file_size);
  size_t n = data.CreateTestData(test_list, test_folder, out);
  std::printf("wrote %zu training files and %zu test images to %s\n", files, n,
              out.c_str());
  return 0;
}
