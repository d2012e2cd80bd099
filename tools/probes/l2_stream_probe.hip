// How fast can a CU pull the GEMM's operand stream out of the L2 - with LDS-DMA (global_load_lds_dwordx4, what gemm_bf16_256
// uses) and with plain global_load_dwordx4 into registers?  Same traffic pattern as the 14400x15360x5120 GEMM: 3420 "tiles",
// each sweeping K = 5120 in 64-wide slabs of 256 A rows + 256 W rows (16-B chunk per lane, 8 rows of 128 B per wave
// instruction), XCD-contiguous tile order with 4-row groups.  No MFMAs, no LDS reads.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/l2_stream_probe.hip -o tools/probes/l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

template <int MODE>  // 0: LDS-DMA, 1: plain loads into registers, 2: half and half
__global__ __launch_bounds__(512) void stream(const char* __restrict__ A, const char* __restrict__ W, unsigned* out, int M, int N, int K,
                                              int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int group_sz = 4 * tiles_n, gid = wg / group_sz, first_m = gid * 4;
  const int gm = min(tiles_m - first_m, 4);
  const int tm = first_m + (wg % group_sz) % gm, tn = (wg % group_sz) / gm;
  const int m0 = tm * 256, n0 = tn * 256;
  // per wave instruction: 8 rows x 128 B; a K-slab of one operand half (128 rows) = 16 instructions = 2 per wave
  uint32_t a_off[2][2], w_off[2][2];
  for (int h = 0; h < 2; ++h)
    for (int r = 0; r < 2; ++r) {
      const int row = (r * 8 + wave) * 8 + (lane >> 3);
      a_off[h][r] = (uint32_t)min(m0 + h * 128 + row, M - 1) * (uint32_t)(K * 2) + (lane & 7) * 16;
      w_off[h][r] = (uint32_t)min(n0 + h * 128 + row, N - 1) * (uint32_t)(K * 2) + (lane & 7) * 16;
    }
  u32x4 acc = {0, 0, 0, 0};
  const int kt_n = K / 64;
  for (int kt = 0; kt < kt_n; ++kt) {
    const char* a = A + (size_t)kt * 128;
    const char* w = W + (size_t)kt * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        unsigned char* dstA = smem + ((kt & 1) * 4 + h) * 16384 + (r * 8 + wave) * 1024;
        unsigned char* dstW = smem + ((kt & 1) * 4 + 2 + h) * 16384 + (r * 8 + wave) * 1024;
        const bool dma = MODE == 0 || (MODE == 2 && h == 0);
        if (dma) {
          __builtin_amdgcn_global_load_lds((gbl_void*)(a + a_off[h][r]), (lds_void*)dstA, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gbl_void*)(w + w_off[h][r]), (lds_void*)dstW, 16, 0, 0);
        } else {
          const u32x4 x = *reinterpret_cast<const u32x4*>(a + a_off[h][r]);
          const u32x4 y = *reinterpret_cast<const u32x4*>(w + w_off[h][r]);
          acc ^= x;
          acc ^= y;
        }
      }
    if ((kt & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // keep ~2 slabs in flight, like the GEMM
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[tid] = acc[0];
}

template <int MODE>
float run(const char* A, const char* W, unsigned* out, int M, int N, int K) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
  hipFuncSetAttribute((const void*)stream<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  stream<MODE><<<tiles_m * tiles_n, 512, 131072>>>(A, W, out, M, N, K, tiles_m, tiles_n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) stream<MODE><<<tiles_m * tiles_n, 512, 131072>>>(A, W, out, M, N, K, tiles_m, tiles_n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const int M = 14400, N = 15360, K = 5120;
  char *A, *W;
  unsigned* out;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&W, (size_t)N * K * 2);
  hipMalloc(&out, 4096);
  hipMemset(A, 1, (size_t)M * K * 2);
  hipMemset(W, 2, (size_t)N * K * 2);
  const double bytes = 3420.0 * 80 * 65536;
  const char* names[] = {"LDS-DMA (global_load_lds_dwordx4)", "plain global_load_dwordx4 -> VGPR", "half LDS-DMA, half plain"};
  for (int rep = 0; rep < 2; ++rep) {
    float t0 = run<0>(A, W, out, M, N, K), t1 = run<1>(A, W, out, M, N, K), t2 = run<2>(A, W, out, M, N, K);
    printf("%-40s %.3f ms  %.1f TB/s\n", names[0], t0, bytes / t0 / 1e9);
    printf("%-40s %.3f ms  %.1f TB/s\n", names[1], t1, bytes / t1 / 1e9);
    printf("%-40s %.3f ms  %.1f TB/s\n", names[2], t2, bytes / t2 / 1e9);
  }
  return 0;
}
