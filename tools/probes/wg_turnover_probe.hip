// How long does a CU need to turn over ONE fat workgroup (256 threads, 128 KiB of LDS, 512 registers per lane = the footprint of the
// large-tile GEMMs)?  Launches grids of R x #CUs workgroups that (a) do nothing, (b) read their kernel arguments, compute an address and
// issue one 1 KiB LDS-DMA piece per wave and wait for it (the GEMM's first dependent memory round trip).  time / R = the per-tile cost a
// persistent kernel with a prefetched first K-tile could hide.   hipcc --offload-arch=gfx950 -O3 tools/probes/wg_turnover_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fat_wg(const unsigned short* __restrict__ src, float* __restrict__ out,
                                                                                      int rows, int ld) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");  // the whole register file is allocated
  if (MODE == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = (blockIdx.x * 32 + 8 * wave + (lane >> 3)) % rows;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0xffffffffu, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + wave * 1024), 16, (unsigned)row * (unsigned)(ld * 2) + (lane & 7) * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (MODE == 2) {  // + one 16-byte store per thread (an epilogue's last instruction)
    const float v = *reinterpret_cast<const float*>(smem + threadIdx.x * 4);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
  }
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int rows = 14400, ld = 5120;
  unsigned short* src;
  float* out;
  hipMalloc(&src, (size_t)rows * ld * 2);
  hipMemset(src, 0, (size_t)rows * ld * 2);
  hipMalloc(&out, (size_t)64 * cus * 256 * 4);
  const int lds = 128 * 1024;
  hipFuncSetAttribute((const void*)fat_wg<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)fat_wg<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)fat_wg<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int R : {1, 4, 12, 48}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(fat_wg<0>, dim3(R * cus), dim3(256), lds, 0, src, out, rows, ld);
        if (mode == 1) hipLaunchKernelGGL(fat_wg<1>, dim3(R * cus), dim3(256), lds, 0, src, out, rows, ld);
        if (mode == 2) hipLaunchKernelGGL(fat_wg<2>, dim3(R * cus), dim3(256), lds, 0, src, out, rows, ld);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("mode %d (%s) rounds %2d: %.1f us total, %.2f us per round\n", mode,
             mode == 0 ? "empty" : mode == 1 ? "one LDS-DMA round trip" : "round trip + store", R, best * 1e3f, best * 1e3f / R);
    }
  return 0;
}
