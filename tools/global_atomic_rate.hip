// Microbenchmark: how fast do fp32 atomic adds from many workgroups reach HBM-resident accumulators on gfx950, by the
// shape in which a workgroup issues them?  Every workgroup owns one 16 x 16 tile of a 1920 x 1080 image and adds a
// 17 x 17 footprint (its tile shifted by a sub-pixel flow) onto five accumulator planes, as the splat's flush does:
//   0  window sweep: 40 x 40 window cells in lane order, lanes outside the footprint idle (the splat's round-2 flush)
//   1  footprint sweep: the 289 footprint cells in lane order (17 consecutive pixels per row), plane by plane
//   2  footprint sweep, planes interleaved in memory ([pixel][5]): lanes = 85 consecutive floats per row
//   3  plain stores in shape 1 (what the same traffic costs without the atomic units)
// Printed: microseconds per launch of `tiles` workgroups and atomics per second.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/global_atomic_rate.hip -o global_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int W = 1920, H = 1080, P = W * H;

template <int SHAPE>
__global__ void __launch_bounds__(256) flush_kernel(float *acc, int tiles_x, int n_tiles, float v) {
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int ox = (tile % tiles_x) * 16 + 3, oy = (tile / tiles_x) * 16 + 2;
    if (SHAPE == 0) {
      for (int i = threadIdx.x; i < 1600; i += 256) {
        const int wy = i / 40, wx = i % 40;
        const int ty = oy + wy, tx = ox + wx;
        if (wy >= 17 || wx >= 17 || tx >= W || ty >= H) continue;
#pragma unroll
        for (int pl = 0; pl < 5; ++pl)
          __hip_atomic_fetch_add(acc + (size_t)pl * P + (size_t)ty * W + tx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (SHAPE == 1 || SHAPE == 3) {
      for (int i = threadIdx.x; i < 289; i += 256) {
        const int ty = oy + i / 17, tx = ox + i % 17;
        if (tx >= W || ty >= H) continue;
#pragma unroll
        for (int pl = 0; pl < 5; ++pl) {
          if (SHAPE == 1)
            __hip_atomic_fetch_add(acc + (size_t)pl * P + (size_t)ty * W + tx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            acc[(size_t)pl * P + (size_t)ty * W + tx] = v;
        }
      }
    } else {
      for (int i = threadIdx.x; i < 289 * 5; i += 256) {
        const int row = i / 85, col = i % 85;
        const int ty = oy + row;
        if (ox + col / 5 >= W || ty >= H) continue;
        __hip_atomic_fetch_add(acc + ((size_t)ty * W + ox) * 5 + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

template <int SHAPE>
static void run(const char *name, float *acc, int n_tiles, int grid) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(flush_kernel<SHAPE>, dim3(grid), dim3(256), 0, 0, acc, 120, n_tiles, 1.0f);
  (void)hipEventRecord(e0, 0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(flush_kernel<SHAPE>, dim3(grid), dim3(256), 0, 0, acc, 120, n_tiles, 1.0f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 10.0;
  printf("%-46s %5d tiles: %7.1f us per launch, %6.1f G adds/s\n", name, n_tiles, us, n_tiles * 289.0 * 5 / us * 1e-3);
}

int main() {
  float *acc;
  (void)hipMalloc(&acc, (size_t)P * 5 * sizeof(float) + 4096);
  (void)hipMemset(acc, 0, (size_t)P * 5 * sizeof(float));
  for (int n_tiles : {1375, 8160}) {
    const int grid = n_tiles < 2048 ? n_tiles : 2048;
    run<0>("window sweep (round-2 flush)", acc, n_tiles, grid);
    run<1>("footprint sweep, plane by plane", acc, n_tiles, grid);
    run<2>("footprint sweep, planes interleaved", acc, n_tiles, grid);
    run<3>("plain stores, footprint sweep", acc, n_tiles, grid);
  }
  (void)hipFree(acc);
  return 0;
}
