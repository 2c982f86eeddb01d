// Micro-benchmark: issue cost of single VALU instructions, measured on pure instruction streams (inline asm, 8 independent
// destination registers, 8 waves per SIMD, every CU busy).  Prints nanoseconds per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
#define REP8(OP)                                                                                                          \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                            \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])           \
               : "v"(x), "v"(y), "s"(k))
#define OP_ADD(i) "v_add_f32 %" #i ", %8, %" #i "\n"
#define OP_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_CVT(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %8, %" #i ", %10\n"
#define OP_MAX3(i) "v_max3_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define OP_MINU(i) "v_min_u32 %" #i ", %8, %" #i "\n"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define OP_CMP(i) "v_cmp_lt_f32 vcc, %8, %" #i "\n"
#define OP_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float x, float y, unsigned k)
{
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = x + i + threadIdx.x;
  for (int it = 0; it < N_ITER; ++it)
  {
    if (KIND == 0) REP8(OP_ADD);
    if (KIND == 1) REP8(OP_FMA);
    if (KIND == 2) REP8(OP_CVT);
    if (KIND == 3) REP8(OP_PERM);
    if (KIND == 4) REP8(OP_MAX3);
    if (KIND == 5) REP8(OP_CNDMASK);
    if (KIND == 6) REP8(OP_MINU);
    if (KIND == 7) REP8(OP_LSHLADD);
    if (KIND == 8) REP8(OP_CMP);
    if (KIND == 9) REP8(OP_ANDOR);
    if (KIND == 10) REP8(OP_BFE);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> float run(float* d, int blocks)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, 0.75f, 0x07060500u);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, 0.75f, 0x07060500u);
  (void)hipEventRecord(e1), (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8; // 8 blocks x 4 waves = 8 waves per SIMD
  float* d;
  (void)hipMalloc(&d, blocks * 256 * sizeof(float));
  const char* names[] = {"v_add_f32", "v_fma_f32", "v_cvt_f32_ubyte1", "v_perm_b32", "v_max3_f32", "v_cndmask_b32 (vcc)", "v_min_u32", "v_lshl_add_u32",
                         "v_cmp_lt_f32 (vcc)", "v_and_or_b32", "v_bfe_u32"};
  const float ms[] = {run<0>(d, blocks), run<1>(d, blocks), run<2>(d, blocks), run<3>(d, blocks), run<4>(d, blocks), run<5>(d, blocks),
                      run<6>(d, blocks), run<7>(d, blocks), run<8>(d, blocks), run<9>(d, blocks), run<10>(d, blocks)};
  for (int i = 0; i < 11; ++i)
    printf("%-22s %8.3f ms -> %.3f ns per wave-instruction per SIMD\n", names[i], ms[i], ms[i] * 1e6 / (8.0 * N_ITER * 8.0));
  return 0;
}
