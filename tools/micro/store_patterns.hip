// Micro-benchmark (tools only): how fast can ONE workgroup of four waves per CU store a 256-pixel x 128-channel fp32 tile (128 KB, rows of
// 512 B) from registers, by access shape of the store instruction — the question behind the epilogue of csrc/conv3x3_bf16.hip.
//   0: 16 B per lane, 64 lanes contiguous (1 KB = 8 full 128-B lines per instruction)            — what an LDS-staged epilogue issues
//   1: 16 B per lane, lane = pixel: 32 pixels x (2 x 16 B adjacent) per instruction                — D[channel][pixel] accumulators as they stand
//   2:  4 B per lane, lanes 0-31 one 128-B line, lanes 32-63 another                              — D[pixel][channel] accumulators as they stand
//   3: 16 B per lane, 8 lanes (4 quads x 2 halves) per pixel: 8 pixels x 128 B per instruction     — after a 4 x 4 lane/register transpose
// Build: hipcc -O3 --offload-arch=gfx950 -w store_patterns.hip -o store_patterns (the binary is git-ignored); run on the GPU box: tools/micro/store_patterns [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int tiles_per_wg, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles_per_wg; ++t) {
    const unsigned tile = blockIdx.x * tiles_per_wg + t;
    const unsigned base = tile * (256u * 512u) + wave * (64u * 512u);     // this wave's 64 pixels
    const u32x4 v = {tile, (unsigned)lane, 3u, 4u};
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + i * 1024 + lane * 16, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (mi * 32 + col) * 512 + (ni * 32 + 8 * q + 4 * half) * 4, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(v.x, rs, base + (mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 512 + (ni * 32 + col) * 4, 0, 0);
    } else {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (mi * 32 + (col & ~3) + i) * 512 + (ni * 32 + 8 * (col & 3) + 4 * half) * 4, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(float* out, int wgs, int tiles, unsigned long long* cyc) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, tiles, cyc);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, tiles, cyc);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  unsigned long long s = 0; for (int i = 0; i < wgs; ++i) s += h[i];
  const double bytes = (double)wgs * tiles * 131072.0;
  printf("mode %d: %4d workgroups x %d tiles: %.1f us, %.2f TB/s, %.0f s_memtime ticks per tile and workgroup (%.1f B/tick/CU)\n", MODE, wgs, tiles, ms * 1e3,
         bytes / ms / 1e9, (double)s / wgs / tiles, 131072.0 / ((double)s / wgs / tiles));
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256, tiles = 4;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)1024 * tiles * 131072); hipMalloc(&cyc, sizeof(unsigned long long) * 1024);
  run<0>(out, wgs, tiles, cyc); run<1>(out, wgs, tiles, cyc); run<2>(out, wgs, tiles, cyc); run<3>(out, wgs, tiles, cyc);
  return 0;
}
