// Micro-benchmark: sustained cycles per wave64 VALU instruction on one SIMD, by instruction kind (8 waves/SIMD, every CU busy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed, unsigned useed)
{
  float a[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; u[i] = useed + i * 77u + threadIdx.x; }
  for (int it = 0; it < N_ITER; ++it)
  {
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      if (KIND == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
      if (KIND == 1) a[i] = a[i] * 1.0001f;
      if (KIND == 2) u[i] = u[i] * 1103515245u + (u[i] >> 8);   // v_mul_lo_u32 + shift + add
      if (KIND == 3) a[i] = __builtin_amdgcn_rcpf(a[i]) + 1.5f; // v_rcp + add
      if (KIND == 4) a[i] = 1.0f / a[i] + 1.5f;                 // IEEE divide + add
      if (KIND == 5) a[i] = __builtin_sqrtf(a[i]) + 1.5f;       // IEEE sqrt + add
      if (KIND == 6) u[i] = (u[i] >> 3) ^ (u[i] + 12345u);      // shift, add, xor
      if (KIND == 7) a[i] = (a[i] < 2.0f) ? a[i] + 1.0f : a[i] - 0.5f; // cmp+cndmask+2 adds
      if (KIND == 8 && (i & 1) == 0) // v_pk_fma_f32: two IEEE FMAs per lane and instruction (4 statements per 8)
      {
        typedef float pair2 __attribute__((ext_vector_type(2)));
        pair2 v{a[i], a[i + 1]};
        v = __builtin_elementwise_fma(v, pair2{1.0001f, 0.9999f}, pair2{0.5f, 0.25f});
        a[i] = v.x, a[i + 1] = v.y;
      }
      if (KIND == 9) a[i] = (float)((u[i] >> 8) & 0xffu) + a[i], u[i] += 0x01010101u; // v_cvt_f32_ubyte1 + add + add
      if (KIND == 10) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) & 7]), a[(i + 3) & 7]) * 0.999f; // v_max3_f32 + mul
      if (KIND == 12) a[i] = __uint_as_float(__builtin_amdgcn_perm(0x3f800000u, u[i], 0x07060500u + (i & 3))) + a[i], u[i] += 0x01010101u; // v_perm_b32 + add + add
      if (KIND == 11) { const unsigned x = u[i], y = u[(i + 1) & 7]; u[i] = (x < y ? x : y) + 3u; } // v_min_u32 + add
    }
  }
  float r = 0; unsigned ru = 0;
  for (int i = 0; i < 8; ++i) { r += a[i]; ru ^= u[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = r + (float)ru;
}
template <int KIND> float run(float* d, int blocks)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, 7u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, 7u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8; // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
  float* d; hipMalloc(&d, blocks * 256 * sizeof(float));
  const char* names[] = {"fma_f32", "mul_f32", "mul_lo_u32+lshr+add", "rcp+add", "ieee_div+add", "ieee_sqrt+add", "lshr+add+xor", "cmp+cndmask+2add",
                         "pk_fma_f32 (4 per 8)", "cvt_f32_ubyte+2add", "max3_f32+mul", "min_u32+add", "perm_b32+2add"};
  float ms[13] = {run<0>(d, blocks), run<1>(d, blocks), run<2>(d, blocks), run<3>(d, blocks), run<4>(d, blocks), run<5>(d, blocks), run<6>(d, blocks), run<7>(d, blocks),
                  run<8>(d, blocks), run<9>(d, blocks), run<10>(d, blocks), run<11>(d, blocks), run<12>(d, blocks)};
  // per SIMD: 8 waves x N_ITER x 8 statements
  for (int i = 0; i < 13; ++i)
  {
    const double stmts_per_simd = 8.0 * N_ITER * 8.0;
    printf("%-22s %8.3f ms  -> %.2f ns per wave-statement per SIMD (x clock GHz = cycles; at 2.4 GHz: %.2f cycles)\n", names[i], ms[i], ms[i] * 1e6 / stmts_per_simd,
           ms[i] * 1e6 / stmts_per_simd * 2.4);
  }
  printf("clock MHz %d CUs %d\n", p.clockRate / 1000, p.multiProcessorCount);
  return 0;
}
