// Sustained bf16 MFMA rate of the whole chip for the two dense shapes, with operands that toggle (random bits) or not (zeros):
// 512-thread workgroups (two waves per SIMD, as in the GEMM / attention kernels), 256 x 4 of them, every wave a stream of
// independent accumulators (8 x 16x16x32 or 4 x 32x32x16: the same 128 accumulator registers).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe.hip -o tools/probes/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const uint32_t* __restrict__ seed, float* out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  u32x4 ra[4], rb[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      // bf16 pairs with exponents near 1.0 and random mantissas / signs (seed == 0: all operands zero)
      const uint32_t s0 = seed[(tid * 32 + i * 8 + j * 2) & 0xfffff], s1 = seed[(tid * 32 + i * 8 + j * 2 + 1) & 0xfffff];
      ra[i][j] = s0 ? ((s0 & 0x807f807fu) | 0x3f003f00u) : 0u;
      rb[i][j] = s1 ? ((s1 & 0x807f807fu) | 0x3f003f00u) : 0u;
    }
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[32];
    for (int m = 0; m < 32; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[u & 3]), __builtin_bit_cast(bf16x8, rb[(u >> 2) & 3]), acc[u], 0, 0, 0);
    }
    for (int m = 0; m < 32; ++m) s += acc[m][0] + acc[m][3];
  } else {
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m)
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        acc[u & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[u & 3]), __builtin_bit_cast(bf16x8, rb[(u >> 2) & 3]), acc[u & 7], 0, 0, 0);
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
  const int iters = 20000;  // x 32 (16x16x32) or x 16 (32x32x16) MFMAs = 1.05e10 flop per wave either way
  const double flop = 2.0 * 16 * 16 * 32 * 32.0 * iters * (256.0 * 4 * 8);
  for (int rep = 0; rep < 2; ++rep)
    for (int shape : {16, 32})
      for (int zero = 0; zero < 2; ++zero) {
        const uint32_t* sd = zero ? d_zero : d_rand;
        hipEventRecord(e0);
        if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(1024), dim3(512), 0, 0, sd, out, iters);
        else hipLaunchKernelGGL(k<32>, dim3(1024), dim3(512), 0, 0, sd, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d  %s  operands %-6s : %8.3f ms  %7.1f TFLOP/s\n", rep, shape == 16 ? "16x16x32" : "32x32x16", zero ? "zero" : "random", ms,
               flop / ms / 1e9);
      }
  return 0;
}
