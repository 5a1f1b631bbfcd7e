// Microbenchmark: sustained issue cost of vector instructions on gfx950, by opcode and by waves per SIMD.
// For each op a wave runs ITER x 16 independent instructions between two s_memtime stamps; the figure printed is
// shader cycles per wave-instruction PER SIMD = (stamp difference) / (instructions of one wave x waves per SIMD),
// median over all waves, every CU busy.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int ITER = 2048;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float *out, unsigned long long *cyc, float seed) {
  float a[16], b = seed + threadIdx.x, c = seed * 0.5f;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = f2{a[i], a[i] + 1.0f};
  f2 pb = {b, c};
  if (OP == 15 || OP == 16) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 1) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 2) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (OP == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
      REP16(X)
#undef X
    } else if (OP == 4) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      REP16(X)
#undef X
    } else if (OP == 5) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (OP == 6) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      REP16(X)
#undef X
    } else if (OP == 7) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (OP == 8) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (OP == 10) {
#define X(i) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 11) {
#define X(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 12) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (OP == 13) {  // the insertion chain's shape: slot i = med3(slot i-1, slot i, c)
#define X(i) asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(a[(i + 15) & 15]), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 14) {
#define X(i) asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(a[(i + 15) & 15]), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 15) {  // IEEE mode bit cleared (no sNaN quieting in min / max / med3)
#define X(i) asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(a[(i + 15) & 15]), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 16) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (OP == 17) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (OP == 18) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (OP == 9) {  // dependent chain on one register (latency)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, int per_op) {
  printf("%-28s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;  // 256-thread blocks = one wave per SIMD each; wps blocks per CU
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // wall clock: nanoseconds per wave-instruction per SIMD (1024 SIMDs), independent of what s_memtime counts
    const double ns_per = (double)ms * 1e6 / 4.0 / ((double)blocks * 4 * ITER * 16 * per_op / 1024.0);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("  %dw: %5.2f tk %5.3f ns", wps, med / ((double)ITER * 16 * per_op * wps), ns_per);
    hipFree(out);
    hipFree(cyc);
  }
  printf("\n");
}

int main() {
  printf("per wave64 instruction per SIMD, by waves per SIMD: s_memtime ticks of the median wave / its instructions / waves per SIMD (tk), "
         "and wall-clock ns (hipEvent over 4 launches; 2 cycles at 2.4 GHz = 0.833 ns)\n");
  run<0>("v_fma_f32", 1);
  run<1>("v_med3_f32", 1);
  run<2>("v_min_f32", 1);
  run<7>("v_sub_f32", 1);
  run<8>("v_mul_f32", 1);
  run<5>("v_add_u32", 1);
  run<3>("v_pk_fma_f32", 1);
  run<4>("v_cmp_lt_f32 + v_cndmask", 2);
  run<6>("v_rcp_f32", 1);
  run<9>("v_fma_f32 dependent chain", 1);
  run<18>("v_fma_f32 a, a, b, b", 1);
  run<10>("v_med3_u32", 1);
  run<11>("v_med3_i32", 1);
  run<12>("v_min_u32", 1);
  run<17>("v_max3_f32", 1);
  run<13>("v_med3_f32 chain shape", 1);
  run<14>("v_med3_u32 chain shape", 1);
  run<15>("v_med3_f32 chain, IEEE=0", 1);
  run<16>("v_min_f32, IEEE=0", 1);
  return 0;
}
