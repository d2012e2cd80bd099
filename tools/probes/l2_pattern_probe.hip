// What bounds the GEMM's operand stream (L2 -> LDS by LDS-DMA) at ~15-19 TB/s when the L2 is quoted at ~34 TB/s?
// Same 14400x15360x5120 traffic volume as tools/probes/l2_stream_probe.hip (3420 tiles x 80 K-slabs x 64 KiB), LDS-DMA only,
// three address patterns:
//   0  the GEMM's own raster: 4x8-tile XCD footprint, 8 rows x 128 B per wave instruction, row pitch K*2 B   (~81 % L2 hits)
//   1  every workgroup of an XCD streams the SAME tile (all L2 hits after the first toucher)                   (L2 -> LDS ceiling)
//   2  the GEMM's raster, but each 16 KiB half-tile slab is ONE contiguous block (tile-packed operands)       (pitch / TLB effect)
// and two in-flight depths (DEPTH = K-slabs between counted waits).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/l2_pattern_probe.hip -o tools/probes/l2_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

template <int PAT, int DEPTH>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ A, const char* __restrict__ W, int M, int N, int K, int tiles_m,
                                              int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  if (PAT == 1) wg = (blockIdx.x & 7) * 37;  // one tile per XCD
  const int group_sz = 4 * tiles_n, gid = wg / group_sz, first_m = gid * 4;
  const int gm = min(tiles_m - first_m, 4);
  const int tm = first_m + (wg % group_sz) % gm, tn = (wg % group_sz) / gm;
  const int m0 = tm * 256, n0 = tn * 256;
  const int kt_n = K / 64;
  uint32_t a_off[2][2], w_off[2][2];
  for (int h = 0; h < 2; ++h)
    for (int r = 0; r < 2; ++r) {
      if (PAT == 2) {  // slab (half-tile h of k-slab kt) = contiguous 16 KiB at ((row_block) * kt_n + kt) * 16384
        a_off[h][r] = (uint32_t)(min(m0 / 128 + h, M / 128 - 1)) * (uint32_t)(kt_n * 16384) + (r * 8 + wave) * 1024 + lane * 16;
        w_off[h][r] = (uint32_t)(min(n0 / 128 + h, N / 128 - 1)) * (uint32_t)(kt_n * 16384) + (r * 8 + wave) * 1024 + lane * 16;
      } else {
        const int row = (r * 8 + wave) * 8 + (lane >> 3);
        a_off[h][r] = (uint32_t)min(m0 + h * 128 + row, M - 1) * (uint32_t)(K * 2) + (lane & 7) * 16;
        w_off[h][r] = (uint32_t)min(n0 + h * 128 + row, N - 1) * (uint32_t)(K * 2) + (lane & 7) * 16;
      }
    }
  for (int kt = 0; kt < kt_n; ++kt) {
    const size_t ko = PAT == 2 ? (size_t)kt * 16384 : (size_t)kt * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        unsigned char* dstA = smem + ((kt & 1) * 4 + h) * 16384 + (r * 8 + wave) * 1024;
        unsigned char* dstW = smem + ((kt & 1) * 4 + 2 + h) * 16384 + (r * 8 + wave) * 1024;
        __builtin_amdgcn_global_load_lds((gbl_void*)(A + ko + a_off[h][r]), (lds_void*)dstA, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(W + ko + w_off[h][r]), (lds_void*)dstW, 16, 0, 0);
      }
    if ((kt % DEPTH) == DEPTH - 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PAT, int DEPTH>
float run(const char* A, const char* W, int M, int N, int K) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
  hipFuncSetAttribute((const void*)stream<PAT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  stream<PAT, DEPTH><<<tiles_m * tiles_n, 512, 131072>>>(A, W, M, N, K, tiles_m, tiles_n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) stream<PAT, DEPTH><<<tiles_m * tiles_n, 512, 131072>>>(A, W, M, N, K, tiles_m, tiles_n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const int M = 14400, N = 15360, K = 5120;
  char *A, *W;
  hipMalloc(&A, (size_t)(M + 256) * K * 2);
  hipMalloc(&W, (size_t)N * K * 2);
  hipMemset(A, 1, (size_t)(M + 256) * K * 2);
  hipMemset(W, 2, (size_t)N * K * 2);
  const double bytes = 3420.0 * 80 * 65536;
  for (int rep = 0; rep < 2; ++rep) {
    float t;
    t = run<0, 4>(A, W, M, N, K); printf("GEMM raster, row-strided, wait every 4 slabs   %.3f ms  %.1f TB/s\n", t, bytes / t / 1e9);
    t = run<0, 1>(A, W, M, N, K); printf("GEMM raster, row-strided, wait every slab      %.3f ms  %.1f TB/s\n", t, bytes / t / 1e9);
    t = run<1, 4>(A, W, M, N, K); printf("one tile per XCD (all L2 hits), every 4 slabs  %.3f ms  %.1f TB/s\n", t, bytes / t / 1e9);
    t = run<2, 4>(A, W, M, N, K); printf("GEMM raster, tile-packed 16 KiB slabs, every 4 %.3f ms  %.1f TB/s\n", t, bytes / t / 1e9);
  }
  return 0;
}
