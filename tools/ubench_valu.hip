// VALU issue-rate microbenchmark for the integer ops the scan kernel could be built from (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096
#define NACC 8

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[NACC], x = seed + threadIdx.x, y = seed * 3 + blockIdx.x;
  float f[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { a[i] = i + threadIdx.x; f[i] = (float)(i + threadIdx.x); }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (OP == 0) a[i] = __builtin_amdgcn_sad_u8(x, a[i] ^ y, a[i]);            // xor + sad (2 ops)
      if (OP == 1) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) % NACC]), "v"(1.0001f));
      if (OP == 6) asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 7) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 8) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "s"(seed), "v"(y));
      if (OP == 9) asm volatile("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 10) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 11) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 12) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 13) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 14) asm volatile("v_cmp_le_u32 vcc, %0, %1" :: "v"(a[i]), "v"(x) : "vcc");
      if (OP == 15) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 16) asm volatile("v_and_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]), "=&v"(f[i]) : "s"(seed), "v"(y));
      if (OP == 17) asm volatile("v_and_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]), "=&v"(f[i]) : "v"(x), "v"(y));
      if (OP == 18) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
      if (OP == 19) asm volatile("v_max_i32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 20) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if (OP == 21) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
      if (OP == 22) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(a[i]) : "v"(x));
      if (OP == 23) asm volatile("v_bfe_u32 %0, %1, %0, 1" : "+v"(a[i]) : "v"(x));
      if (OP == 24) asm volatile("v_xad_u32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "s"(seed));
      if (OP == 25) asm volatile("v_cmp_eq_u32_sdwa s[10:11], %0, %1 src0_sel:BYTE_1 src1_sel:BYTE_0" :: "v"(a[i]), "v"(x) : "s10", "s11");
      if (OP == 26) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      if (OP == 27) asm volatile("v_max3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 28) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x10" : "+v"(a[i]) : "v"(x), "v"(y));
      if (OP == 29) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += a[i] + (uint32_t)f[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, int opsPerIter, uint32_t* d) {
  const int blocks = 256 * 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waveinstr = (double)blocks * 4 * ITERS * NACC * opsPerIter;
  double rate = waveinstr / (ms * 1e-3);                 // wave-instructions per second, whole chip
  double cyc = 1024.0 * 2.4e9 / rate;                    // SIMD-cycles per wave-instruction at 2.4 GHz
  printf("%-22s %8.3f ms  %7.2f Gwaveinstr/s  %5.2f cyc/instr/SIMD (@2.4GHz)  %6.1f Tlaneops/s\n", name, ms, rate / 1e9, cyc, rate * 64 / 1e12);
}

int main() {
  uint32_t* d; hipMalloc(&d, 256 * 16 * 256 * 4);
  run<5>("v_fma_f32", 1, d);
  run<1>("v_sad_u8", 1, d);
  run<8>("v_sad_u8 (sgpr src)", 1, d);
  run<10>("v_sad_u16", 1, d);
  run<11>("v_sad_u32", 1, d);
  run<2>("v_add_u32", 1, d);
  run<3>("v_and_b32", 1, d);
  run<13>("v_xor_b32", 1, d);
  run<4>("v_bcnt_u32_b32", 1, d);
  run<6>("v_min3_u32", 1, d);
  run<7>("v_and_or_b32", 1, d);
  run<9>("v_lshl_add_u32", 1, d);
  run<12>("v_pk_add_u16", 1, d);
  run<14>("v_cmp_le_u32", 1, d);
  run<15>("v_dot4_u32_u8", 1, d);
  run<18>("v_and_b32 (sgpr src)", 1, d);
  run<19>("v_max_i32", 1, d);
  run<20>("v_lshrrev_b32", 1, d);
  run<21>("v_ffbl_b32", 1, d);
  run<22>("v_alignbyte_b32", 1, d);
  run<23>("v_bfe_u32", 1, d);
  run<24>("v_xad_u32", 1, d);
  run<25>("v_cmp_eq_u32_sdwa sgpr", 1, d);
  run<26>("v_cndmask_b32 vcc", 1, d);
  run<27>("v_max3_u32", 1, d);
  run<28>("v_bitop3_b32", 1, d);
  run<29>("v_min_u32", 1, d);
  run<16>("and(sgpr)+bcnt pair", 2, d);
  run<17>("and(vgpr)+bcnt pair", 2, d);
  return 0;
}
