// Baseline JPEG (ITU T.81 sequential DCT, Huffman, 8 bit) encoder and decoder
// written for this library: the record codecs below need no image library
// (the reference's JPGEncoder / JPGDecoder call OpenCV,
// src/io/jpg_encoder.cc:29-80, src/io/jpg_decoder.cc).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace singa {
namespace jpeg {

/// Encode interleaved 8-bit pixels (channels = 1 grey or 3 RGB) as a baseline
/// JFIF file, 4:4:4 sampling, standard Huffman tables, libjpeg quality scale.
/// `comment` (may be empty) is stored in a COM segment.
std::string Encode(const uint8_t *pixels, int width, int height, int channels,
                   int quality = 95, const std::string &comment = "");

/// Decode a baseline (SOF0) JPEG: 1 or 3 components, sampling factors up to
/// 4 (subsampled components are interpolated bilinearly between sample
/// centres), restart intervals.  Pixels come back interleaved (grey or RGB).  Returns
/// false, with a reason in *error, for progressive / arithmetic / 12-bit files
/// or damaged streams.
bool Decode(const std::string &data, std::vector<uint8_t> *pixels, int *width,
            int *height, int *channels, std::string *comment = nullptr,
            std::string *error = nullptr);

}  // namespace jpeg
}  // namespace singa
