// kernels_prefix.hpp -- exclusive prefix sum (k_scan_local / k_scan_sums / k_scan_add)
// Part of the single translation unit engine.hip (included inside namespace anx); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------------
// Exclusive prefix sum (u32), three small kernels.  out has n+1 entries.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8, SCAN_THREADS = 256, SCAN_TILE = SCAN_ITEMS * SCAN_THREADS;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int i = 0; i < SCAN_THREADS / 64; ++i) {
    if (i < wid) base += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_local(const uint32_t* __restrict__ in, uint32_t n,
                                                             uint32_t* __restrict__ out,
                                                             uint32_t* __restrict__ blocksum, uint32_t* __restrict__ maxout) {
  const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = i0 + i < n ? in[i0 + i] : 0;
    s += v[i];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (i0 + i < n) out[i0 + i] = ex;
    ex += v[i];
  }
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
  if (maxout) {  // also the largest input value (one atomic per wave): the row capacity a fixed-stride export needs
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) mx = max(mx, v[i]);
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(maxout, mx);
  }
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(uint32_t* __restrict__ blocksum, uint32_t nb) {
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t v = i < nb ? blocksum[i] : 0;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, &tot);
    if (i < nb) blocksum[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) blocksum[nb] = carry;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add(uint32_t* __restrict__ out, uint32_t n,
                                                           const uint32_t* __restrict__ blocksum, uint32_t nb) {
  const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  const uint32_t add = blocksum[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (i0 + i < n) out[i0 + i] += add;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = blocksum[nb];
}

