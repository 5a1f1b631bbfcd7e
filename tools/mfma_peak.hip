// Sustained fp32-MFMA rate of the device (calibration for the roofline fraction of the GNT kernels):
// register-only loops of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, W waves per SIMD.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void k32(float *out, int iters) {
  floatx16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ void k16(float *out, int iters) {
  floatx4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
// does vector-ALU work run in the shadow of the matrix pipe?  NV independent v_fma per 4 MFMAs
// (4 x 32 = 128 matrix cycles); same wave (W = 1) or spread over W waves per SIMD
// dependent accumulation chains: NA independent accumulators, round-robin (NA = 1: every MFMA
// waits for its predecessor's result)
template <int NA, bool BIG>
__global__ void kchain(float *out, int iters) {
  floatx16 a[4] = {};
  floatx4 b[4] = {};
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (BIG) a[k % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[k % NA], 0, 0, 0);
      else b[k % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b[k % NA], 0, 0, 0);
    }
  }
  float r = 0;
  for (int k = 0; k < 4; ++k) r += a[k][0] + b[k][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NA, bool BIG>
void run_chain(float *out, hipEvent_t e0, hipEvent_t e1) {
  for (int wpb : {256, 512, 1024}) {
    const int iters = 50000;
    kchain<NA, BIG><<<256, wpb>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kchain<NA, BIG><<<256, wpb>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * (wpb / 256));
    printf("chain %s accumulators=%d waves/SIMD=%d: %.1f cycles per MFMA (pipe: %d)\n", BIG ? "32x32x2" : "16x16x4", NA, wpb / 256, cyc, BIG ? 64 : 32);
  }
}

// MFMAs whose A operand arrives from LDS (one ds_read_b32 per MFMA, double-buffered in chunks
// of 8 like the GNT kernels): does the feed cost matrix-pipe cycles?
template <bool BIG>
__global__ void klds(float *out, int iters) {
  __shared__ float lds[16384];
  for (int k = threadIdx.x; k < 16384; k += blockDim.x) lds[k] = k * 1e-6f;
  __syncthreads();
  floatx16 a[2] = {};
  floatx4 b[4] = {};
  float y = 1.0f + blockIdx.x * 1e-6f;
  const float *wb = lds + (threadIdx.x & 63);
  float w0[8], w1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) w0[u] = wb[u * 64];
  for (int i = 0; i < iters; ++i) {
    const float *wi = wb + (i & 15) * 1024;
#pragma unroll
    for (int u = 0; u < 8; ++u) w1[u] = wi[512 + u * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (BIG) a[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[u], y, a[u & 1], 0, 0, 0);
      else b[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[u], y, b[u & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) w0[u] = wi[u * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (BIG) a[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[u], y, a[u & 1], 0, 0, 0);
      else b[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[u], y, b[u & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float r = 0;
  for (int k = 0; k < 2; ++k) r += a[k][0];
  for (int k = 0; k < 4; ++k) r += b[k][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <bool BIG>
void run_lds(float *out, hipEvent_t e0, hipEvent_t e1) {
  for (int wpb : {256, 512, 1024}) {
    const int iters = 30000;
    klds<BIG><<<256, wpb>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    klds<BIG><<<256, wpb>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (wpb / 256));
    printf("lds-fed %s waves/SIMD=%d: %.1f cycles per MFMA (pipe: %d)\n", BIG ? "32x32x2" : "16x16x4", wpb / 256, cyc, BIG ? 64 : 32);
  }
}

template <int OP>
__device__ __forceinline__ float vop(float a, float y, float x) {
  if (OP == 0) return __builtin_fmaf(a, y, x);                                          // v_fma_f32
  if (OP == 1) return fmaxf(a, y) ;                                                      // v_max_f32 (chain kept alive below)
  if (OP == 2) return __builtin_amdgcn_exp2f(a);                                         // v_exp_f32 (transcendental)
  if (OP == 3) return __int_as_float((__float_as_int(a) + __float_as_int(y)) ^ 0x5a5a);  // integer add + xor (2 ops)
  return __int_as_float(__float_as_int(a) > 7 ? __float_as_int(y) : __float_as_int(x)); // v_cmp + v_cndmask (2 ops)
}
template <int NV, int OP = 0>
__global__ void kmix(float *out, int iters) {
  floatx4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = x + k;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[k % 8] = vop<OP>(v[k % 8], y, x);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[(k + 2) % 8] = vop<OP>(v[(k + 2) % 8], y, x);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[(k + 4) % 8] = vop<OP>(v[(k + 4) % 8], y, x);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) v[(k + 6) % 8] = vop<OP>(v[(k + 6) % 8], y, x);
  }
  float sv = 0;
  for (int k = 0; k < 8; ++k) sv += v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + sv;
}
template <int NV, int OP = 0>
void run_mix(float *out, hipEvent_t e0, hipEvent_t e1) {
  for (int wpb : {256, 512, 1024}) {
    const int iters = 100000;
    kmix<NV, OP><<<256, wpb>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kmix<NV, OP><<<256, wpb>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * (wpb / 256));
    printf("mix op=%d NV=%2d valu per 4 mfma (128 matrix cycles), waves/SIMD=%d: %.1f cycles per iteration-wave @2.4GHz\n", OP, NV, wpb / 256, cyc);
  }
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wpb : {256, 512, 1024}) {          // 1, 2, 4 waves per SIMD (one block per CU)
    for (int which = 0; which < 2; ++which) {
      for (int iters : {20000, 200000}) {      // ~1 ms and ~10 ms+ launches: does the clock hold?
        if (which == 0) k32<<<256, wpb>>>(out, 100); else k16<<<256, wpb>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (which == 0) k32<<<256, wpb>>>(out, iters); else k16<<<256, wpb>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = 256.0 * (wpb / 64) * iters * 4.0 * (which == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4);
        printf("%s waves/SIMD=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", which == 0 ? "32x32x2 " : "16x16x4 ", wpb / 256, iters, ms,
               flop / ms / 1e9);
      }
    }
  }
  run_chain<1, true>(out, e0, e1);
  run_chain<2, true>(out, e0, e1);
  run_chain<1, false>(out, e0, e1);
  run_chain<2, false>(out, e0, e1);
  run_chain<4, false>(out, e0, e1);
  run_lds<true>(out, e0, e1);
  run_lds<false>(out, e0, e1);
  run_mix<0>(out, e0, e1);
  run_mix<8>(out, e0, e1);
  run_mix<16>(out, e0, e1);
  run_mix<24>(out, e0, e1);
  run_mix<32>(out, e0, e1);
  run_mix<48>(out, e0, e1);
  run_mix<16, 1>(out, e0, e1);
  run_mix<16, 2>(out, e0, e1);
  run_mix<16, 3>(out, e0, e1);
  run_mix<16, 4>(out, e0, e1);
  return 0;
}
