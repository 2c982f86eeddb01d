// Micro-benchmark: how fast can a wave fetch one 64-byte record PER LANE from scattered addresses (the BVH node fetch)?
//   A  per-lane: every lane issues four 16-byte loads of its own record (what k_trace did in round 1)
//   B  quad-cooperative through registers + LDS: lanes 4k..4k+3 load the four quarters of record k of a 16-record group
//      (one 64-byte line per quad), write them to LDS, every lane reads its own record back
//   C  quad-cooperative LDS-DMA (global_load_lds_dwordx4, per-lane source address): no VGPR round trip
// Each wave runs a dependent chain (the next index comes out of the fetched record) so latency is exposed as in traversal.
// Reported: wave-steps per ms over the chip, and cycles per wave-step per CU at the 2.4 GHz nominal clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
  do                                                                                          \
  {                                                                                           \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess)                                                                     \
    {                                                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);           \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

constexpr int kSteps = 256;

__device__ __forceinline__ uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// activeLanes: lanes >= activeLanes idle (their EXEC bit is off during the fetch)
template <int VARIANT>
__global__ __launch_bounds__(256, 6) void k_gather(const uint4* __restrict__ table, uint32_t mask, uint32_t* out, uint32_t activeLanes)
{
  __shared__ uint4 sm[4][64 * 4]; // one 4-KiB staging area per wave
  const uint32_t lane = laneId(), wv = threadIdx.x >> 6;
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
  uint32_t acc = 0;
  uint4* my = sm[wv];
  const bool on = lane < activeLanes;
  for (int s = 0; s < kSteps; ++s)
  {
    uint4 q0, q1, q2, q3;
    if (VARIANT == 0)
    {
      if (on)
      {
        const uint4* p = table + (size_t)idx * 4;
        q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
      }
    }
    else if (VARIANT == 1)
    {
      // group g: lane L loads quarter (L&3) of the record of lane g*16 + (L>>2)
#pragma unroll
      for (int g = 0; g < 4; ++g)
      {
        const uint32_t src = g * 16 + (lane >> 2);
        const uint32_t i = __shfl(idx, src);
        const bool srcOn = src < activeLanes;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (srcOn)
          v = table[(size_t)i * 4 + (lane & 3u)];
        // record k's quarter q goes to slot k*4 + ((q + (k>>2)) & 3): conflict-free 16-byte reads afterwards
        const uint32_t k = src, q = lane & 3u;
        my[k * 4 + ((q + (k >> 2)) & 3u)] = v;
      }
      const uint32_t r = lane >> 2;
      q0 = my[lane * 4 + ((0 + r) & 3u)], q1 = my[lane * 4 + ((1 + r) & 3u)], q2 = my[lane * 4 + ((2 + r) & 3u)], q3 = my[lane * 4 + ((3 + r) & 3u)];
    }
    else if (VARIANT == 3)
    {
      // D: one 16-byte access per lane (a 16-byte record at a 64-byte stride)
      q1 = q2 = q3 = make_uint4(1, 2, 3, 4);
      if (on)
        q0 = table[(size_t)idx * 4];
    }
    else if (VARIANT == 4)
    {
      // E: the quad-cooperative loads alone (no redistribution: each lane keeps the quarter it fetched)
      uint4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
      {
        const uint32_t src = g * 16 + (lane >> 2);
        const uint32_t i = __shfl(idx, src);
        v[g] = make_uint4(0, 0, 0, 0);
        if (src < activeLanes)
          v[g] = table[(size_t)i * 4 + (lane & 3u)];
      }
      q0 = v[0], q1 = v[1], q2 = v[2], q3 = v[3];
    }
    else if (VARIANT == 5)
    {
      // F: four 4-byte accesses per lane, 16 bytes apart (same four lines per lane as A)
      q0 = q1 = q2 = q3 = make_uint4(1, 2, 3, 4);
      if (on)
      {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(table + (size_t)idx * 4);
        q0.x = p[0], q1.x = p[4], q2.y = p[8], q3.z = p[12];
      }
    }
    else if (VARIANT == 6)
    {
      // G: two 16-byte accesses per lane (a 32-byte record)
      q2 = q3 = make_uint4(1, 2, 3, 4);
      if (on)
      {
        const uint4* p = table + (size_t)idx * 4;
        q0 = p[0], q1 = p[1];
      }
    }
    else if (VARIANT == 7)
    {
      // H: like A but groups of 4 neighbouring lanes fetch the SAME record (coherent rays)
      if (on)
      {
        const uint4* p = table + (size_t)__shfl(idx, lane & ~3u) * 4;
        q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
      }
    }
    else
    {
#pragma unroll
      for (int g = 0; g < 4; ++g)
      {
        // LDS slot (g*64 + L) receives what lane L fetches; slot s = k*4 + ((q + (k>>2)) & 3) => lane L of group g holds
        // k = g*16 + (L>>2), q = ((L&3) - (k>>2)) & 3
        const uint32_t k = g * 16 + (lane >> 2);
        const uint32_t q = ((lane & 3u) - (k >> 2)) & 3u;
        const uint32_t i = __shfl(idx, k);
        const uint4* gp = table + (size_t)i * 4 + q;
        if (k < activeLanes)
          __builtin_amdgcn_global_load_lds(gp, (__attribute__((address_space(3))) void*)(my + g * 64), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t r = lane >> 2;
      q0 = my[lane * 4 + ((0 + r) & 3u)], q1 = my[lane * 4 + ((1 + r) & 3u)], q2 = my[lane * 4 + ((2 + r) & 3u)], q3 = my[lane * 4 + ((3 + r) & 3u)];
    }
    if (on)
    {
      acc += q1.x ^ q2.y ^ q3.z ^ q0.w;
      idx = (q0.x + acc) & mask; // dependent chain
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
float run(const uint4* table, uint32_t mask, uint32_t* out, int blocks, uint32_t activeLanes)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_gather<V>), dim3(blocks), dim3(256), 0, 0, table, mask, out, activeLanes);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_gather<V>), dim3(blocks), dim3(256), 0, 0, table, mask, out, activeLanes);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main()
{
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int blocks = cus * 6; // 24 waves per CU, as k_trace
  uint32_t* out;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  const char* names[8] = {"A per-lane 4x16B", "B quad-coop reg+LDS", "C quad-coop LDS-DMA", "D per-lane 1x16B", "E quad-coop loads only", "F per-lane 4x4B", "G per-lane 2x16B", "H 4 lanes share a record"};
  for (int logRecords : {15, 21}) // 2 MiB (L2), 128 MiB (Infinity Cache), 1 GiB (HBM)
  {
    const uint32_t n = 1u << logRecords;
    std::vector<uint32_t> host((size_t)n * 16);
    uint32_t x = 12345u;
    for (size_t i = 0; i < host.size(); ++i)
    {
      x = x * 1664525u + 1013904223u;
      host[i] = x >> 4;
    }
    uint4* table;
    CHECK(hipMalloc(&table, (size_t)n * 64));
    CHECK(hipMemcpy(table, host.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    for (uint32_t active : {64u, 32u})
      for (int v = 0; v < 8; ++v)
      {
        float ms = 0;
        switch (v)
        {
        case 0: ms = run<0>(table, n - 1, out, blocks, active); break;
        case 1: ms = run<1>(table, n - 1, out, blocks, active); break;
        case 2: ms = run<8>(table, n - 1, out, blocks, active); break;
        case 3: ms = run<3>(table, n - 1, out, blocks, active); break;
        case 4: ms = run<4>(table, n - 1, out, blocks, active); break;
        case 5: ms = run<5>(table, n - 1, out, blocks, active); break;
        case 6: ms = run<6>(table, n - 1, out, blocks, active); break;
        default: ms = run<7>(table, n - 1, out, blocks, active); break;
        }
        const double waveSteps = (double)blocks * 4 * kSteps;
        printf("table %4u MiB  lanes %2u  %-22s %8.3f ms  %7.3f M wave-steps/ms  %6.0f cycles per wave-step per CU (2.4 GHz)\n", (unsigned)((size_t)n * 64 >> 20),
               active, names[v], ms, waveSteps / ms / 1e6, ms * 1e-3 * 2.4e9 / (waveSteps / cus));
      }
    CHECK(hipFree(table));
  }
  printf("CUs %d\n", cus);
  return 0;
}
