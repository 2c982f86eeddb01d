// Math.h -- the vector arithmetic of the hot path, compiled for host (g++) and device (hipcc) alike.
//
// The reference does all vector math through glm (un-vendored vcpkg dependency, vcpkg.json:13).  These
// functions fix the operation ORDER glm's scalar code paths use, because the hit-id parity target is
// bit-exactness: dot = (x+y)+z, normalize = v * (1/sqrt(dot)), min/max in the (b<a)?b:a forms, and so on.
// Both compilers must be run with -ffp-contract=off so that no multiply-add pair is fused; the places
// where a fused multiply-add is WANTED (conservative BVH box tests) call rt::fma explicitly.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt
{
struct vec2
{
  float x, y;
};
struct vec3
{
  float x, y, z;
};
struct uvec3
{
  uint32_t x, y, z;
};
// column-major like glm::mat4: c[col][row]
struct mat4
{
  float c[4][4];
  static mat4 identity()
  {
    mat4 m{};
    for (int i = 0; i < 4; ++i)
      m.c[i][i] = 1.0f;
    return m;
  }
};

RT_HD vec3 make3(float a) { return vec3{a, a, a}; }
RT_HD vec3 operator+(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD vec3 operator-(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD vec3 operator-(vec3 a) { return vec3{-a.x, -a.y, -a.z}; }
RT_HD vec3 operator*(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD vec3 operator*(vec3 a, float s) { return vec3{a.x * s, a.y * s, a.z * s}; }
RT_HD vec3 operator*(float s, vec3 a) { return vec3{s * a.x, s * a.y, s * a.z}; }
RT_HD vec3 operator/(vec3 a, float s) { return vec3{a.x / s, a.y / s, a.z / s}; } // true division, as glm
RT_HD vec3 rcp3(vec3 a) { return vec3{1.0f / a.x, 1.0f / a.y, 1.0f / a.z}; }      // glm: 1.0f / vec3

RT_HD vec2 operator+(vec2 a, vec2 b) { return vec2{a.x + b.x, a.y + b.y}; }
RT_HD vec2 operator*(vec2 a, vec2 b) { return vec2{a.x * b.x, a.y * b.y}; }
RT_HD vec2 operator/(vec2 a, vec2 b) { return vec2{a.x / b.x, a.y / b.y}; }

RT_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD vec3 cross(vec3 a, vec3 b) { return vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
// sqrtf is correctly rounded on both sides: SSE sqrtss on the host; on gfx950 hipcc expands it to v_sqrt_f32 plus
// the two-FMA fix-up under -fhip-fp32-correctly-rounded-divide-sqrt (__fsqrt_rn does NOT: it stays at 1 ulp).
RT_HD float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RT_HD float length(vec3 v) { return rt_sqrt(dot(v, v)); }
RT_HD vec3 normalize(vec3 v) { return v * (1.0f / rt_sqrt(dot(v, v))); }
// the comparison forms of glm::min/max and std::min/max (they differ from fminf/fmaxf when a NaN is involved)
RT_HD float minRef(float a, float b) { return (b < a) ? b : a; }
RT_HD float maxRef(float a, float b) { return (a < b) ? b : a; }

RT_HD float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// glm mat4 * vec4(p,1): (m0*x + m1*y) + (m2*z + m3*w); xyz kept
inline vec3 transformPoint(const mat4& m, vec3 p)
{
  float r[3];
  for (int k = 0; k < 3; ++k)
  {
    const float a = m.c[0][k] * p.x + m.c[1][k] * p.y;
    const float b = m.c[2][k] * p.z + m.c[3][k] * 1.0f;
    r[k] = a + b;
  }
  return vec3{r[0], r[1], r[2]};
}
// glm mat3(m) * v
inline vec3 transformDirection(const mat4& m, vec3 v)
{
  return vec3{m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z, m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z,
              m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z};
}
} // namespace rt
