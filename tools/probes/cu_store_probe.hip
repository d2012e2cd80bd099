// How fast can ONE compute unit push a GEMM C tile to memory (and pull a residual tile), alone and with every other CU doing the same?
// Each workgroup (256 threads, 140 KiB of LDS: one per CU) stores / loads a 384 x 256 bf16 tile (196 608 B) of a [M][5120] matrix with
// the row-contiguous 16-byte pattern of the GEMM epilogue and stamps s_memtime around it (stores: until vmcnt(0)).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/cu_store_probe.hip -o tools/probes/cu_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, bool NT = false>  // 0 store, 1 load, 2 load then store; NT: non-temporal accesses
__global__ __launch_bounds__(256) void tile_io(unsigned short* C, const unsigned short* R, long long* stamps, int ld, int tiles_n, int reps) {
  extern __shared__ unsigned char smem[];
  const int tid = threadIdx.x;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  unsigned short* c0 = C + ((size_t)tm * 384) * ld + tn * 256;
  const unsigned short* r0 = R + ((size_t)tm * 384) * ld + tn * 256;
  if (tid == 0) smem[0] = 1;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  u32x4 acc = {1u, 2u, 3u, (unsigned)tid};
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 4
    for (int i = 0; i < 48; ++i) {  // 48 chunks per thread: rows (tid >> 5) + 8 i, 16-byte chunk tid & 31
      const size_t off = (size_t)((tid >> 5) + 8 * i) * ld + (tid & 31) * 8;
      if (MODE >= 1) {
        const u32x4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r0 + off)) : *reinterpret_cast<const u32x4*>(r0 + off);
        acc[0] += v[0]; acc[1] ^= v[1]; acc[2] += v[2]; acc[3] ^= v[3];
      }
      if (MODE == 0) { if (NT) __builtin_nontemporal_store(acc, reinterpret_cast<u32x4*>(c0 + off)); else *reinterpret_cast<u32x4*>(c0 + off) = acc; }
    }
    if (MODE == 2) {
#pragma unroll 4
      for (int i = 0; i < 48; ++i) {
        const size_t off = (size_t)((tid >> 5) + 8 * i) * ld + (tid & 31) * 8;
        if (NT) __builtin_nontemporal_store(acc, reinterpret_cast<u32x4*>(c0 + off)); else *reinterpret_cast<u32x4*>(c0 + off) = acc;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  if (MODE == 1 && acc[0] == 0x12345678u) C[0] = 1;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

int main() {
  const int ld = 5120, tiles_n = 20, M = 384 * 64;  // 1280 tiles available
  unsigned short *C, *R;
  long long* st;
  hipMalloc(&C, (size_t)M * ld * 2);
  hipMalloc(&R, (size_t)M * ld * 2);
  hipMalloc(&st, 4096 * 8);
  hipMemset(C, 0, (size_t)M * ld * 2);
  hipMemset(R, 1, (size_t)M * ld * 2);
  const int lds = 140 * 1024;
  hipFuncSetAttribute((const void*)tile_io<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)tile_io<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)tile_io<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)tile_io<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)tile_io<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)tile_io<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const char* names[6] = {"store 196 KB", "load 196 KB", "load + store", "store nt", "load nt", "load + store nt"};
  for (int mode = 0; mode < 6; ++mode)
    for (int n : {1, 64, 128, 256}) {
      std::vector<long long> h(n);
      double best = 1e30, bestmax = 0;
      for (int it = 0; it < 5; ++it) {
        if (mode == 0) hipLaunchKernelGGL(tile_io<0>, dim3(n), dim3(256), lds, 0, C, R, st, ld, tiles_n, 1);
        if (mode == 1) hipLaunchKernelGGL(tile_io<1>, dim3(n), dim3(256), lds, 0, C, R, st, ld, tiles_n, 1);
        if (mode == 2) hipLaunchKernelGGL(tile_io<2>, dim3(n), dim3(256), lds, 0, C, R, st, ld, tiles_n, 1);
        if (mode == 3) hipLaunchKernelGGL((tile_io<0, true>), dim3(n), dim3(256), lds, 0, C, R, st, ld, tiles_n, 1);
        if (mode == 4) hipLaunchKernelGGL((tile_io<1, true>), dim3(n), dim3(256), lds, 0, C, R, st, ld, tiles_n, 1);
        if (mode == 5) hipLaunchKernelGGL((tile_io<2, true>), dim3(n), dim3(256), lds, 0, C, R, st, ld, tiles_n, 1);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, n * 8, hipMemcpyDeviceToHost);
        double s = 0, mx = 0;
        for (auto v : h) { s += v; mx = std::max(mx, (double)v); }
        if (s / n < best) { best = s / n; bestmax = mx; }
      }
      printf("%-14s %3d workgroups (one per CU): mean %8.0f cycles  max %8.0f  -> %5.1f B/cycle/CU\n", names[mode], n, best, bestmax,
             (mode % 3 == 2 ? 2 : 1) * 196608.0 / best);
    }
  return 0;
}
