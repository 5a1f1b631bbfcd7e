// Microbenchmark: what an LDS float atomic costs on gfx950, next to a plain LDS store and an integer atomic.
// Every workgroup (256 threads) issues ITER x 8 LDS operations per thread between two s_memtime stamps; addresses follow the
// splat window's pattern (a 16 x 16 source tile translated into a 40-wide window: lane (lx, ly) -> row ly, column lx) or are
// simply consecutive.  Printed: wall-clock nanoseconds per wave-level instruction per CU (hipEvent over four launches), for 1 and 4
// workgroups per CU.   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITER = 512;

template <int OP, int PATTERN>
__global__ void __launch_bounds__(256) lds_kernel(float *out, unsigned long long *cyc, float seed) {
  __shared__ float s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.0f;
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  int base = PATTERN == 0 ? (int)threadIdx.x : ly * 40 + lx;  // (PATTERN 3: the window pattern with fp32 denormals flushed)
  if (PATTERN == 2) base = (threadIdx.x & 63) / 4 + (threadIdx.x >> 6) * 64;  // four lanes share an address
  float v = seed + threadIdx.x;
  if (PATTERN == 3) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0");  // fp32 denormals flushed
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int a = base + (k & 1) + (k >> 1) * (PATTERN == 0 ? 256 : 40) + (it & 3) * 1700;
      if (OP == 0) __hip_atomic_fetch_add(&s[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 1) reinterpret_cast<volatile float *>(s)[a] = v;
      if (OP == 2) __hip_atomic_fetch_add(reinterpret_cast<int *>(s) + a, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 3) v += __hip_atomic_fetch_add(&s[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * 1e-30f;
      if (OP == 5) {  // compare-and-swap loop on the bit pattern
        unsigned *w = reinterpret_cast<unsigned *>(s) + a;
        unsigned old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (;;) {
          const unsigned want = __float_as_uint(__uint_as_float(old) + v);
          const unsigned seen = atomicCAS(w, old, want);
          if (seen == old) break;
          old = seen;
        }
      }
      if (OP == 4) {  // read-modify-write without atomicity
        const float o = reinterpret_cast<volatile float *>(s)[a];
        reinterpret_cast<volatile float *>(s)[a] = o + v;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x] + v;
}

template <int OP, int PATTERN>
static void run(const char *name, int per_cu) {
  int dev = 0;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, dev);
  const int blocks = prop.multiProcessorCount * per_cu;
  float *out;
  unsigned long long *cyc;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  (void)hipMalloc(&cyc, (size_t)blocks * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((lds_kernel<OP, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
  (void)hipEventRecord(e0, 0);
  for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL((lds_kernel<OP, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // wall clock per launch; all CUs run per_cu workgroups side by side
  const double ns = (double)ms * 1e6 / 4.0;
  const double insts = (double)ITER * 8 * 4 * per_cu;  // wave-level instructions per CU
  printf("%-44s %d WG/CU: %8.1f us per launch, %6.2f ns per wave-instruction per CU\n", name, per_cu, ns * 1e-3, ns / insts);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  for (int per_cu : {1, 4}) {
    run<0, 0>("ds_add_f32, consecutive addresses", per_cu);
    run<0, 1>("ds_add_f32, splat window pattern", per_cu);
    run<0, 2>("ds_add_f32, four lanes per address", per_cu);
    run<3, 1>("ds_add_rtn_f32, splat window pattern", per_cu);
    run<2, 1>("ds_add_u32, splat window pattern", per_cu);
    run<1, 1>("ds_write_b32, splat window pattern", per_cu);
    run<4, 1>("ds_read + add + ds_write, splat pattern", per_cu);
    run<5, 1>("compare-and-swap loop, splat pattern", per_cu);
    run<5, 2>("compare-and-swap loop, four lanes per address", per_cu);
    run<0, 3>("ds_add_f32, fp32 denormals flushed (MODE)", per_cu);
  }
  return 0;
}
