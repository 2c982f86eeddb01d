// rt_probes.h: kernels behind the bit-exactness probes of include/rthip.h (tests only) -- part of the gfx950 kernel library (see rthip.hip for the pipeline overview and DESIGN.md section 5).
#pragma once
#include "rt_kernel_common.h"

namespace rtk
{
__global__ void k_probe_rand01(const uint32_t* xyz, uint32_t n, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const vec3 r = rtd::rand01(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
}
__global__ void k_probe_sincos(const float* x, uint32_t n, float* s, float* c)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    rtd::sincosDet(x[i], s[i], c[i]);
}
__global__ void k_probe_rays(const CameraConsts* camp, uint32_t width, uint32_t height, const uint32_t* whs, uint32_t n, float* o3, float* d3)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const CameraConsts k = *camp;
  const uint32_t w = whs[3 * i], h = whs[3 * i + 1], s = whs[3 * i + 2];
  vec3 o, d;
  rtd::primaryRay(k, width, height, w, h, s, rtd::rand01(w, h, s), o, d);
  o3[3 * i] = o.x, o3[3 * i + 1] = o.y, o3[3 * i + 2] = o.z;
  d3[3 * i] = d.x, d3[3 * i + 1] = d.y, d3[3 * i + 2] = d.z;
}
__global__ void k_probe_bounce(const float* n3, const float* r3, uint32_t n, float* d3)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const vec3 d = rtd::bounceDirection(vec3{n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]}, vec3{r3[3 * i], r3[3 * i + 1], r3[3 * i + 2]});
  d3[3 * i] = d.x, d3[3 * i + 1] = d.y, d3[3 * i + 2] = d.z;
}
__global__ void k_probe_pack(const float* o3, const float* d3, uint32_t n, float4* ro, float4* rd, uint8_t* live)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  ro[i] = make_float4(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2], 0.0f);
  rd[i] = make_float4(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2], 0.0f);
  live[i] = 0;
}
__global__ void k_probe_unpack(const DeviceScene scene, const uint2* hit, uint32_t n, uint32_t* triId, float* t)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const uint2 h = hit[i];
  t[i] = __uint_as_float(h.x);
  triId[i] = (h.y == RTFLAT_INVALID) ? RTFLAT_INVALID : scene.tris[h.y].triId;
}
} // namespace rtk
