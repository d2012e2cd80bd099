// Sustained fp8 (OCP e4m3, MX-scaled form) MFMA rate of the whole chip, with operands that toggle (random e4m3 bytes) or not (zeros):
// the power-limited ceiling the fp8 GEMM / MXFP8 attention kernels face (the bf16 twin: mfma_rate_probe.hip, profiles/r01_mfma_rate_probe.txt).
// Two launch forms: 512-thread workgroups (two waves per SIMD) and 256-thread workgroups with the full register file (one wave per SIMD, as
// gemm_fp8_w4 runs); every wave a stream of independent accumulators (32 x 16x16x128 or 8 x 32x32x64 = 128 accumulator registers).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe_fp8.hip -o tools/probes/mfma_rate_probe_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const uint32_t* __restrict__ seed, float* out, int iters) {
  const int tid = blockIdx.x * THREADS + threadIdx.x;
  i32x8 ra[4], rb[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      // e4m3 bytes with exponents around 1.0 (0x30..0x3f), random mantissas / signs (seed == 0: all operands zero)
      const uint32_t s0 = seed[(tid * 64 + i * 16 + j * 2) & 0xfffff], s1 = seed[(tid * 64 + i * 16 + j * 2 + 1) & 0xfffff];
      ra[i][j] = s0 ? (int)((s0 & 0x8f8f8f8fu) | 0x30303030u) : 0;
      rb[i][j] = s1 ? (int)((s1 & 0x8f8f8f8fu) | 0x30303030u) : 0;
    }
  const int sc = seed[tid & 0xfffff] ? 0x7f7e7f7e : 0x7f7f7f7f;  // E8M0 scale bytes (2^0 / 2^-1)
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[32];
    for (int m = 0; m < 32; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
        acc[u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ra[u & 3], rb[(u >> 2) & 3], acc[u], 0, 0, 0, sc, 0, sc);
    }
    for (int m = 0; m < 32; ++m) s += acc[m][0] + acc[m][3];
  } else {
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m)
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        acc[u & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ra[u & 3], rb[(u >> 2) & 3], acc[u & 7], 0, 0, 0, sc, 0, sc);
    }
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][15];
  }
  if (s == 123.456f) out[tid] = s;
}

int main() {
  const int nseed = 1 << 20;
  uint32_t* h = new uint32_t[nseed];
  uint32_t x = 12345;
  for (int i = 0; i < nseed; ++i) { x = x * 1664525u + 1013904223u; h[i] = x | 1u; }
  uint32_t *d_rand, *d_zero;
  float* out;
  hipMalloc(&d_rand, nseed * 4); hipMalloc(&d_zero, nseed * 4); hipMalloc(&out, 256 * 4 * 512 * 4);
  hipMemcpy(d_rand, h, nseed * 4, hipMemcpyHostToDevice);
  hipMemset(d_zero, 0, nseed * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10000;  // x 32 (16x16x128) or x 16 (32x32x64) MFMAs = 2.1e10 flop per wave either way
  for (int rep = 0; rep < 2; ++rep)
    for (int threads : {512, 256})
      for (int shape : {16, 32})
        for (int zero = 0; zero < 2; ++zero) {
          const uint32_t* sd = zero ? d_zero : d_rand;
          const int waves = 1024 * (threads / 64);
          const double flop = 2.0 * 16 * 16 * 128 * 32.0 * iters * waves;
          hipEventRecord(e0);
          if (shape == 16 && threads == 512) hipLaunchKernelGGL((k<16, 512>), dim3(1024), dim3(512), 0, 0, sd, out, iters);
          else if (shape == 16) hipLaunchKernelGGL((k<16, 256>), dim3(1024), dim3(256), 0, 0, sd, out, iters);
          else if (threads == 512) hipLaunchKernelGGL((k<32, 512>), dim3(1024), dim3(512), 0, 0, sd, out, iters);
          else hipLaunchKernelGGL((k<32, 256>), dim3(1024), dim3(256), 0, 0, sd, out, iters);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          printf("rep %d  %d-thread WGs  %s fp8 (MX-scaled)  operands %-6s : %8.3f ms  %7.1f TFLOP/s\n", rep, threads,
                 shape == 16 ? "16x16x128" : "32x32x64 ", zero ? "zero" : "random", ms, flop / ms / 1e9);
        }
  return 0;
}
