// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probe/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
// Diagnostic: how fast can 256 CUs write a [12288 x 3072] bf16 matrix in the GEMM-epilogue pattern vs linearly?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short bf16_t;
#define M 12288
#define N 3072
__global__ __launch_bounds__(256) void linear_fill(uint4* p, long n16) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += gridDim.x * 256L) p[i] = make_uint4(1, 2, 3, 4);
}
// 128x128 tiles, 8 waves (4x2) of 32 rows x 64 cols; lane writes 8 B; 16 lanes = one 128-B row segment
template <int MODE>
__global__ __launch_bounds__(512) void tile_fill(bf16_t* C) {
  const int nbn = N / 128, nbm = M / 128;
  int bid = blockIdx.x;
  int tm, tn;
  if (MODE & 1) {   // xcd-contiguous supertile order (8 M-tiles x N-major)
    const int nwg = nbm * nbn, q = nwg / 8, xcd = bid % 8, idx = bid / 8;
    bid = xcd * q + idx;
    const int per_group = 8 * nbn, grp = bid / per_group, in = bid - grp * per_group;
    tn = in / 8; tm = grp * 8 + in % 8;
  } else { tm = bid / nbn; tn = bid % nbn; }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
  if (MODE & 2) {   // whole-tile rows: each wave writes 4 rows x 256 B per instruction (16 lanes... 8 B x 32 lanes per row)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int q = p * 64 + lane + wid * 512, r = q / 32, c = q % 32;      // tile = 128 rows x 32 chunks of 8 B
      *reinterpret_cast<uint2*>(C + (long)(tm * 128 + r) * N + tn * 128 + c * 4) = make_uint2(q, p);
    }
  } else {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int q = p * 64 + lane, r = q / 16, c = q % 16;
      *reinterpret_cast<uint2*>(C + (long)(tm * 128 + wm * 32 + r) * N + tn * 128 + wn * 64 + c * 4) = make_uint2(q, p);
    }
  }
}
// row-panel writer: a workgroup owns 8 full rows (8 x 6 KB contiguous)
__global__ __launch_bounds__(512) void panel_fill(bf16_t* C) {
  const int r0 = blockIdx.x * 8;
  for (int i = threadIdx.x; i < 8 * N / 8; i += 512) {
    const int r = i / (N / 8), c = i % (N / 8);
    *reinterpret_cast<uint4*>(C + (long)(r0 + r) * N + c * 8) = make_uint4(i, r, c, 7);
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000;
}
int main() {
  bf16_t* C; hipMalloc(&C, (size_t)M * N * 2 * 2);
  const double mb = (double)M * N * 2 / 1e6;
  float t;
  t = timeit([&] { hipLaunchKernelGGL(linear_fill, dim3(2048), dim3(256), 0, 0, (uint4*)C, (long)M * N * 2 / 16); }); printf("linear        %7.1f us %6.2f TB/s\n", t, mb / t);
  t = timeit([&] { hipLaunchKernelGGL(tile_fill<0>, dim3(M / 128 * N / 128), dim3(512), 0, 0, C); }); printf("tile rowmajor %7.1f us %6.2f TB/s\n", t, mb / t);
  t = timeit([&] { hipLaunchKernelGGL(tile_fill<1>, dim3(M / 128 * N / 128), dim3(512), 0, 0, C); }); printf("tile xcd      %7.1f us %6.2f TB/s\n", t, mb / t);
  t = timeit([&] { hipLaunchKernelGGL(tile_fill<3>, dim3(M / 128 * N / 128), dim3(512), 0, 0, C); }); printf("tile xcd 256B %7.1f us %6.2f TB/s\n", t, mb / t);
  t = timeit([&] { hipLaunchKernelGGL(panel_fill, dim3(M / 8), dim3(512), 0, 0, C); }); printf("row panels    %7.1f us %6.2f TB/s\n", t, mb / t);
  // two outputs like the GELU epilogue
  t = timeit([&] { hipLaunchKernelGGL(tile_fill<1>, dim3(M / 128 * N / 128), dim3(512), 0, 0, C); hipLaunchKernelGGL(tile_fill<1>, dim3(M / 128 * N / 128), dim3(512), 0, 0, C + (size_t)M * N); });
  printf("2x tile xcd   %7.1f us %6.2f TB/s\n", t, 2 * mb / t);
  return 0;
}
