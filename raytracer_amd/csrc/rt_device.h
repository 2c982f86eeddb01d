// rt_device.h -- device-side restatement of the reference's per-ray arithmetic (gfx950, wave64).
// Every function names the reference lines (relative to /root/reference) whose result it must reproduce
// bit for bit.  Compiled with -ffp-contract=off: an fp32 multiply followed by an add stays two roundings,
// exactly as on the reference's CPU path; fused multiply-adds appear only where rt::fma is written out.
#pragma once
#include "../src/util/Math.h"
#include "rthip.h"
#include <cfloat>

namespace rtd
{
using rt::vec2;
using rt::vec3;
using rt::uvec3;

#define RTD __device__ __forceinline__

// ---- src/cpu/CPURenderer.cpp:111-116
RTD vec3 rand01(uint32_t x, uint32_t y, uint32_t z)
{
#pragma unroll
  for (int round = 0; round < 3; ++round)
  {
    const uint32_t nx = ((x >> 8u) ^ y) * 1103515245u;
    const uint32_t ny = ((y >> 8u) ^ z) * 1103515245u;
    const uint32_t nz = ((z >> 8u) ^ x) * 1103515245u;
    x = nx, y = ny, z = nz;
  }
  const float k = 1.0f / float(0xffffffffu); // == 2^-32
  return vec3{float(x) * k, float(y) * k, float(z) * k};
}

// Everything of the ray-generation lambda that does not depend on the pixel (CPURenderer.cpp:159-165,178,
// 182-185,189).  The reference recomputes these per pixel; they are pure functions of the camera, so one
// device thread computes them once per frame with the same operations.
struct CameraConsts
{
  vec3 origin, direction; // cam
  vec3 cx, cy;
  vec3 lc;                // lens centre
  vec3 lensN, lensX, lensY;
  vec3 focalPoint;
  vec2 sdim;
  float A;
  float focalDot; // dot(focalPoint - lc, lensN): the numerator of CPURenderer.cpp:190, the same for every pixel
};

RTD CameraConsts cameraSetup(const rthip_camera& c)
{
  CameraConsts k;
  const vec3 pos{c.position[0], c.position[1], c.position[2]}, tgt{c.target[0], c.target[1], c.target[2]};
  k.origin = pos;
  k.direction = rt::normalize(tgt - pos);                                                                 // :159
  const vec3 up = ((double)fabsf(k.direction.y) < 0.9) ? vec3{0.0f, 1.0f, 0.0f} : vec3{0.0f, 0.0f, 1.0f}; // :161 (double literal)
  k.cx = rt::normalize(rt::cross(k.direction, up));
  k.cy = rt::cross(k.cx, k.direction);                                                                    // :162
  k.sdim = vec2{c.sensorSize[0], c.sensorSize[1]};
  const float S_I = (c.S_O * c.f) / (c.S_O - c.f);                                                        // :165
  k.lc = k.origin + k.direction * 0.035f;                                                                 // :178
  k.lensN = -k.direction;                                                                                 // :183
  k.lensX = rt::cross(k.lensN, vec3{0.0f, 1.0f, 0.0f});                                                   // :184
  k.lensY = rt::cross(k.lensN, k.lensX);                                                                  // :185
  k.focalPoint = k.origin + (c.S_O + S_I) * k.direction;                                                  // :189
  k.A = c.A;
  k.focalDot = rt::dot(k.focalPoint - k.lc, k.lensN);                                                     // :190 (r.origin == lc)
  return k;
}

// ---- CPURenderer.cpp:170-192, the per-(pixel, sample) part
RTD void primaryRay(const CameraConsts& k, uint32_t width, uint32_t height, uint32_t w, uint32_t h, uint32_t samp, vec3 rnd, vec3& o,
                    vec3& d)
{
  const float r2x = 2.0f * rnd.x, r2y = 2.0f * rnd.y;                                            // :171
  // :172-173  rnd2 < 1 ? sqrt(rnd2) - 1 : 1 - sqrt(2 - rnd2), written with one square root per axis
  const float qx = rt::rt_sqrt(r2x < 1 ? r2x : 2 - r2x), qy = rt::rt_sqrt(r2y < 1 ? r2y : 2 - r2y);
  const float tentx = r2x < 1 ? qx - 1 : 1 - qx;
  const float tenty = r2y < 1 ? qy - 1 : 1 - qy;
  const int isamp = (int)samp;
  const float strx = (float)((isamp / 2) % 2), stry = (float)(isamp % 2);
  const float sx = (((float)w + 0.5f * ((0.5f + strx) + tentx)) / (float)width - 0.5f) * k.sdim.x; // :174-177
  const float sy = (((float)h + 0.5f * ((0.5f + stry) + tenty)) / (float)height - 0.5f) * k.sdim.y;
  const vec3 spos = k.origin + k.cx * sx + k.cy * sy;                                             // :178
  const vec3 rdir = rt::normalize(k.lc - spos);                                                   // :179 (origin lc)
  const vec3 lensSample = k.lc + (rnd.x * k.A) * k.lensX + (rnd.y * k.A) * k.lensY;               // :187
  const float t = k.focalDot / rt::dot(rdir, k.lensN);                                            // :190
  const vec3 focus = k.lc + t * rdir;                                                             // :191
  o = lensSample;
  d = rt::normalize(focus - lensSample);                                                          // :192
}

// ---- the shared deterministic sin/cos (same algorithm as oracle/rt_math.h sincos_det)
RTD void sincosDet(float xf, float& s, float& c)
{
  const double x = (double)xf;
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double PIO2_HI = 1.57079632673412561417e+00;
  const double PIO2_LO = 6.07710050650619224932e-11;
  const int q = (int)(x * TWO_OVER_PI + 0.5);
  const double qd = (double)q;
  const double y = (x - qd * PIO2_HI) - qd * PIO2_LO;
  const double z = y * y;
  double ps = -1.0 / 1307674368000.0;
  ps = ps * z + 1.0 / 6227020800.0;
  ps = ps * z - 1.0 / 39916800.0;
  ps = ps * z + 1.0 / 362880.0;
  ps = ps * z - 1.0 / 5040.0;
  ps = ps * z + 1.0 / 120.0;
  ps = ps * z - 1.0 / 6.0;
  ps = ps * z + 1.0;
  const double sy = y * ps;
  double pc = 1.0 / 20922789888000.0;
  pc = pc * z - 1.0 / 87178291200.0;
  pc = pc * z + 1.0 / 479001600.0;
  pc = pc * z - 1.0 / 3628800.0;
  pc = pc * z + 1.0 / 40320.0;
  pc = pc * z - 1.0 / 720.0;
  pc = pc * z + 1.0 / 24.0;
  pc = pc * z - 0.5;
  const double cy = pc * z + 1.0;
  double sr, cr;
  switch (q & 3)
  {
  case 0: sr = sy; cr = cy; break;
  case 1: sr = cy; cr = -sy; break;
  case 2: sr = -sy; cr = -cy; break;
  default: sr = -cy; cr = sy; break;
  }
  s = (float)sr;
  c = (float)cr;
}

// ---- src/cpu/CPUScene.cpp:54-60
RTD vec3 bounceDirection(vec3 normalLight, vec3 rnd)
{
  const float r1 = (float)(2 * 3.141592653589793238462643383279502884 * (double)rnd.x);
  const float r2 = rnd.y;
  const float r2s = rt::rt_sqrt(r2);
  const vec3 w = normalLight;
  const vec3 axis = ((double)fabsf(w.x) > 0.1) ? vec3{0.0f, 1.0f, 0.0f} : vec3{1.0f, 0.0f, 0.0f};
  const vec3 u = rt::normalize(rt::cross(axis, w));
  const vec3 v = rt::cross(w, u);
  float sn, cs;
  sincosDet(r1, sn, cs);
  return rt::normalize(u * cs * r2s + v * sn * r2s + w * rt::rt_sqrt(1 - r2));
}

// ---- src/util/BRDF.cpp:4-14 with the intersector's hard-coded material (CPUScene.cpp:245-246)
RTD vec3 brdfEvaluate(vec3 albedo, vec3 normal, vec3 lightDir, vec3 lightColor)
{
  const float diffuse = rt::maxRef(rt::dot(normal, lightDir), 0.0f);
  return (albedo * diffuse * lightColor) + vec3{0.03f, 0.03f, 0.03f};
}

// ---- src/scene/AABB.h:47-60, used for the model-level boxes only
RTD bool refSlab(const float* lo, const float* hi, vec3 o, vec3 invD, float tmin, float tmax)
{
  const vec3 t0{(lo[0] - o.x) * invD.x, (lo[1] - o.y) * invD.y, (lo[2] - o.z) * invD.z};
  const vec3 t1{(hi[0] - o.x) * invD.x, (hi[1] - o.y) * invD.y, (hi[2] - o.z) * invD.z};
  const vec3 sm{rt::minRef(t0.x, t1.x), rt::minRef(t0.y, t1.y), rt::minRef(t0.z, t1.z)};
  const vec3 bg{rt::maxRef(t0.x, t1.x), rt::maxRef(t0.y, t1.y), rt::maxRef(t0.z, t1.z)};
  tmin = rt::maxRef(tmin, rt::maxRef(sm.x, rt::maxRef(sm.y, sm.z)));
  tmax = rt::minRef(tmax, rt::minRef(bg.x, rt::minRef(bg.y, bg.z)));
  return tmin < tmax;
}

// ---- the triangle test of intersectModel / testModel (CPUScene.cpp:162-181 == :210-229) on a record's
// operands.  Returns 1 when the reference would keep the triangle, 0 when it `continue`s, 2 when every
// rejection passed but a value is not finite (outside the parity domain; counted by the caller).
RTD int triangleTest(vec3 p0, vec3 e0, vec3 e1, vec3 o, vec3 d, float tmin, float tmax, float& tOut)
{
  const vec3 s = o - p0;
  const vec3 s1 = rt::cross(d, e1);
  const vec3 s2 = rt::cross(s, e0);
  const float fraction = 1.0f / rt::dot(s1, e0);
  const float t = rt::dot(s2, e1) * fraction;
  const float b1 = rt::dot(s1, s) * fraction;
  const float b2 = rt::dot(s2, d) * fraction;
  const float b3 = 1.0f - b1 - b2;
  if (b3 < 0 || b3 > 1)
    return 0;
  if (b1 < 0 || b1 > 1)
    return 0;
  if (b2 < 0 || b2 > 1)
    return 0;
  if (t < tmin || t > tmax)
    return 0;
  tOut = t;
  // accepted by the reference; NaN operands pass every comparison above
  if (!(fabsf(t) <= FLT_MAX && fabsf(b1) <= FLT_MAX && fabsf(b2) <= FLT_MAX))
    return 2;
  return 1;
}

// The same test with s1 = cross(d, e1) and fraction = 1 / dot(s1, e0) brought in: for the shadow rays toward one directional light
// they depend on the triangle alone and are computed once per scene (k_shadow_operands) by these very operations, so every
// value below has the bits triangleTest gives it.
RTD int triangleTestWith(vec3 p0, vec3 e0, vec3 e1, vec3 o, vec3 d, vec3 s1, float fraction, float tmin, float tmax, float& tOut)
{
  const vec3 s = o - p0;
  const vec3 s2 = rt::cross(s, e0);
  const float t = rt::dot(s2, e1) * fraction;
  const float b1 = rt::dot(s1, s) * fraction;
  const float b2 = rt::dot(s2, d) * fraction;
  const float b3 = 1.0f - b1 - b2;
  if (b3 < 0 || b3 > 1)
    return 0;
  if (b1 < 0 || b1 > 1)
    return 0;
  if (b2 < 0 || b2 > 1)
    return 0;
  if (t < tmin || t > tmax)
    return 0;
  tOut = t;
  if (!(fabsf(t) <= FLT_MAX && fabsf(b1) <= FLT_MAX && fabsf(b2) <= FLT_MAX))
    return 2;
  return 1;
}

// ---- materials (include/rtflat.h: rtflat_shade / rtflat_material)
// The texture coordinates the reference's intersector attaches to a hit (CPUScene.cpp:217-219): the triangle test's own
// barycentrics, recomputed here for the winning record from the same operands in the same order, so the same bits.
RTD void hitTexCoords(vec3 p0, vec3 e0, vec3 e1, vec3 o, vec3 d, const float uv[3][2], float& u, float& v)
{
  const vec3 s = o - p0;
  const vec3 s1 = rt::cross(d, e1);
  const vec3 s2 = rt::cross(s, e0);
  const float fraction = 1.0f / rt::dot(s1, e0);
  const float b1 = rt::dot(s1, s) * fraction;
  const float b2 = rt::dot(s2, d) * fraction;
  const float b3 = 1.0f - b1 - b2;
  u = uv[0][0] * b1 + uv[1][0] * b2 + uv[2][0] * b3; // t0 * y + t1 * z + t2 * b3 (:219)
  v = uv[0][1] * b1 + uv[1][1] * b2 + uv[2][1] * b3;
}
// src/util/Texture.cpp:5-6: texel column / row = uint32(coordinate * size).  The reference converts unchecked and reads out
// of bounds for coordinates outside [0,1); here those clamp to the edge texels (NaN -> 0), the same on host and device.
RTD uint32_t texelIndex(float coord, uint32_t size)
{
  const float f = coord * (float)size;
  if (!(f >= 0.0f))
    return 0u;
  if (f >= (float)size)
    return size - 1u;
  return (uint32_t)f;
}
RTD vec3 textureSample(const float* texels, uint32_t first, uint32_t w, uint32_t h, float u, float v)
{
  const uint32_t x = texelIndex(u, w), y = texelIndex(v, h);
  const float* t = texels + 3 * ((size_t)first + (size_t)y * w + x); // textureData[y * width + x] (Texture.cpp:7)
  return vec3{t[0], t[1], t[2]};
}
// what src/util/Material.cpp:4-12 stands for: the BRDF inputs of a hit from its material and texture coordinates
RTD void materialEvaluate(const rtflat_material& m, const float* texels, float u, float v, vec3& albedo, vec3& emissive)
{
  if (m.kind == RTFLAT_MAT_STUB) // the reference's body as it stands
  {
    albedo = vec3{u, v, 0.0f};
    emissive = vec3{0.0f, 0.0f, 0.0f};
    return;
  }
  albedo = vec3{m.baseColor[0], m.baseColor[1], m.baseColor[2]};
  emissive = vec3{m.emissive[0], m.emissive[1], m.emissive[2]};
  if (m.albedoW)
    albedo = albedo * textureSample(texels, m.albedoTexel, m.albedoW, m.albedoH, u, v);
  if (m.emissiveW)
    emissive = emissive * textureSample(texels, m.emissiveTexel, m.emissiveW, m.emissiveH, u, v);
}

// 1/d for the padded-box tests, clamped to +-1e18 so that an exactly axis-parallel ray (d == 0 -> 1/d == inf) keeps
// finite slab arithmetic (inf - inf would turn into a NaN that fminf/fmaxf silently drop)
RTD vec3 padInvDir(vec3 d)
{
  const vec3 r = rt::rcp3(d);
  return vec3{fminf(fmaxf(r.x, -1e18f), 1e18f), fminf(fmaxf(r.y, -1e18f), 1e18f), fminf(fmaxf(r.z, -1e18f), 1e18f)};
}

// ---- conservative slab tests for the 4-wide quantised triangle-BVH nodes (no counterpart in the reference).
// child box on axis a: origin[a] + q * step[a]; its slab distances are q * (step*invD) + (origin*invD - o*invD), so the
// per-node part is 3 multiplies + 3 FMAs and every plane of every child costs one FMA.
struct WideNode
{
  float ax, ay, az, bx, by, bz; // per-axis slope and offset of the slab distance as a function of the grid coordinate
  uint32_t nx, ny, nz, fx, fy, fz; // 4 children x 8 bits each: the grid planes the ray enters / leaves through
};
// The plane a ray enters a slab through is the low one when it travels in +axis and the high one otherwise, so the
// near/far choice is made once per node (six selects) instead of per child with min/max.  An unused slot (qlo 255,
// qhi 0) then has entry > exit for every ray.
RTD WideNode wideNodeSetup(float4 w0, float4 w2, float4 w3, vec3 invD, vec3 oi)
{
  WideNode n;
  const uint32_t e = __float_as_uint(w0.w);
  const float sx = __uint_as_float((e & 0xffu) << 23), sy = __uint_as_float(((e >> 8) & 0xffu) << 23), sz = __uint_as_float(((e >> 16) & 0xffu) << 23);
  n.ax = sx * invD.x, n.ay = sy * invD.y, n.az = sz * invD.z;
  n.bx = rt::fma(w0.x, invD.x, -oi.x), n.by = rt::fma(w0.y, invD.y, -oi.y), n.bz = rt::fma(w0.z, invD.z, -oi.z);
  const uint32_t qlx = __float_as_uint(w2.x), qly = __float_as_uint(w2.y), qlz = __float_as_uint(w2.z), qhx = __float_as_uint(w2.w);
  const uint32_t qhy = __float_as_uint(w3.x), qhz = __float_as_uint(w3.y);
  const bool bx = invD.x < 0.0f, by = invD.y < 0.0f, bz = invD.z < 0.0f;
  n.nx = bx ? qhx : qlx, n.fx = bx ? qlx : qhx;
  n.ny = by ? qhy : qly, n.fy = by ? qly : qhy;
  n.nz = bz ? qhz : qlz, n.fz = bz ? qlz : qhz;
  return n;
}
// two children at a time: the six plane distances of children K and K+1 are three packed FMAs per side
// (v_pk_fma_f32: two IEEE fp32 FMAs per instruction, same bits as two scalar ones)
typedef float pair2 __attribute__((ext_vector_type(2)));
template <int K>
RTD void wideSlabPair(const WideNode& n, float& entry0, float& exit0, float& entry1, float& exit1)
{
  const pair2 ax{n.ax, n.ax}, ay{n.ay, n.ay}, az{n.az, n.az}, bx{n.bx, n.bx}, by{n.by, n.by}, bz{n.bz, n.bz};
  const pair2 qnx{(float)((n.nx >> (8 * K)) & 0xffu), (float)((n.nx >> (8 * K + 8)) & 0xffu)};
  const pair2 qny{(float)((n.ny >> (8 * K)) & 0xffu), (float)((n.ny >> (8 * K + 8)) & 0xffu)};
  const pair2 qnz{(float)((n.nz >> (8 * K)) & 0xffu), (float)((n.nz >> (8 * K + 8)) & 0xffu)};
  const pair2 qfx{(float)((n.fx >> (8 * K)) & 0xffu), (float)((n.fx >> (8 * K + 8)) & 0xffu)};
  const pair2 qfy{(float)((n.fy >> (8 * K)) & 0xffu), (float)((n.fy >> (8 * K + 8)) & 0xffu)};
  const pair2 qfz{(float)((n.fz >> (8 * K)) & 0xffu), (float)((n.fz >> (8 * K + 8)) & 0xffu)};
  const pair2 nx = __builtin_elementwise_fma(qnx, ax, bx), ny = __builtin_elementwise_fma(qny, ay, by), nz = __builtin_elementwise_fma(qnz, az, bz);
  const pair2 fx = __builtin_elementwise_fma(qfx, ax, bx), fy = __builtin_elementwise_fma(qfy, ay, by), fz = __builtin_elementwise_fma(qfz, az, bz);
  entry0 = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, 0.0f));
  exit0 = fminf(fminf(fx.x, fy.x), fz.x);
  entry1 = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, 0.0f));
  exit1 = fminf(fminf(fx.y, fy.y), fz.y);
}
// compare-exchange on (key, ref): afterwards ka <= kb.  Keys are the bit patterns of non-negative distances
// (0xffffffff = missed), so unsigned order == distance order.
RTD void orderPair(uint32_t& ka, uint32_t& ra, uint32_t& kb, uint32_t& rb)
{
  const bool sw = kb < ka;
  const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb, r0 = sw ? rb : ra, r1 = sw ? ra : rb;
  ka = k0, kb = k1, ra = r0, rb = r1;
}
} // namespace rtd
