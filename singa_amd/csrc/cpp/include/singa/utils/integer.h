// Variable-length integer codec used by the message layer.
// API parity: reference include/singa/utils/integer.h (appendInteger /
// readInteger; big-endian, 1..9 bytes selected by magnitude).
#pragma once
#include <cstddef>
#include <cstdint>

namespace singa {

/// Encode `val` at `buf`; returns the number of bytes written (1, 2, 3, 5, 9).
inline std::size_t appendInteger(char *buf, uint64_t val) {
  auto *p = reinterpret_cast<unsigned char *>(buf);
  if (val < 128) { p[0] = static_cast<unsigned char>(val); return 1; }
  if (val < (1ull << 14)) {
    p[0] = 0x80 | static_cast<unsigned char>(val >> 8);
    p[1] = static_cast<unsigned char>(val);
    return 2;
  }
  if (val < (1ull << 21)) {
    p[0] = 0xC0 | static_cast<unsigned char>(val >> 16);
    p[1] = static_cast<unsigned char>(val >> 8);
    p[2] = static_cast<unsigned char>(val);
    return 3;
  }
  if (val < (1ull << 32)) {
    p[0] = 0xE0;
    for (int i = 0; i < 4; ++i)
      p[1 + i] = static_cast<unsigned char>(val >> (24 - 8 * i));
    return 5;
  }
  p[0] = 0xF0;
  for (int i = 0; i < 8; ++i)
    p[1 + i] = static_cast<unsigned char>(val >> (56 - 8 * i));
  return 9;
}

/// Decode at `buf` into `val`; returns the number of bytes consumed.
inline std::size_t readInteger(const char *buf, uint64_t &val) {
  auto *p = reinterpret_cast<const unsigned char *>(buf);
  if (p[0] < 0x80) { val = p[0]; return 1; }
  if (p[0] < 0xC0) { val = (uint64_t(p[0] & 0x3F) << 8) | p[1]; return 2; }
  if (p[0] < 0xE0) {
    val = (uint64_t(p[0] & 0x1F) << 16) | (uint64_t(p[1]) << 8) | p[2];
    return 3;
  }
  int n = p[0] < 0xF0 ? 4 : 8;
  val = 0;
  for (int i = 0; i < n; ++i) val = (val << 8) | p[1 + i];
  return n + 1;
}

}  // namespace singa
