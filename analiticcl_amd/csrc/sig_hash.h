// sig_hash.h -- hash of a group-sum signature: the host table builders and the device probes must agree
#pragma once
#include <cstdint>

#ifdef __HIP__
#define ANX_HOST_DEVICE __attribute__((host)) __attribute__((device))
#else
#define ANX_HOST_DEVICE
#endif

namespace anx {
ANX_HOST_DEVICE inline uint32_t sig_hash(uint32_t lo, uint32_t hi) {
  uint32_t h = lo * 0x9E3779B1u ^ hi * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 13;
  return h;
}
}  // namespace anx
