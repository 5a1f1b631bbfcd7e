// gfx950 bf16 MFMA probes: sustained rate of v_mfma_f32_16x16x32_bf16, and whether vector-ALU
// instructions issued between them cost matrix-pipe time (they do for the fp32 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV>
__global__ void kbf(float *out, int iters) {
  floatx4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8 x, y;
  for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(threadIdx.x * 1e-3f + k); y[k] = (__bf16)(1.0f + k * 0.01f); }
  float v[8], xs = threadIdx.x * 1e-3f, ys = 1.0f + blockIdx.x * 1e-6f;
  for (int k = 0; k < 8; ++k) v[k] = xs + k;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[k % 8] = __builtin_fmaf(v[k % 8], ys, xs);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[(k + 2) % 8] = __builtin_fmaf(v[(k + 2) % 8], ys, xs);
    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[(k + 4) % 8] = __builtin_fmaf(v[(k + 4) % 8], ys, xs);
    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[(k + 6) % 8] = __builtin_fmaf(v[(k + 6) % 8], ys, xs);
  }
  float sv = 0;
  for (int k = 0; k < 8; ++k) sv += v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + sv;
}
template <int NV>
void run(float *out, hipEvent_t e0, hipEvent_t e1) {
  for (int wpb : {256, 512, 1024}) {
    const int iters = 200000;
    kbf<NV><<<256, wpb>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kbf<NV><<<256, wpb>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * (wpb / 256));
    double tf = 256.0 * (wpb / 64) * iters * 4.0 * (2.0 * 16 * 16 * 32) / ms / 1e9;
    printf("bf16 16x16x32, NV=%2d valu per 4 mfma, waves/SIMD=%d: %.1f cycles per 4 MFMAs, %.0f TFLOP/s\n", NV, wpb / 256, cyc, tf);
  }
}
int main() {
  float *out;
  hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  run<0>(out, e0, e1); run<8>(out, e0, e1); run<16>(out, e0, e1); run<32>(out, e0, e1);
  return 0;
}
