// TEST INFRASTRUCTURE -- part of the CPU oracle (see oracle/README.md).  Nothing
// outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// include, link or execute this.
//
// rt_math.h : restatement of the glm operations the reference's hot path uses.
// glm is an un-vendored vcpkg dependency of the reference (vcpkg.json:13, no
// version pinned), so the operation ORDER of each function below is taken from
// glm's long-stable scalar code paths (SURVEY.md section 8a "arithmetic
// contract").  PARITY UNPINNED at this boundary: no reference test fixes these
// orders; the known-answer tests in tests/test_oracle_kat.py are the pin.
//
// Build with -ffp-contract=off: every operation here must round once, alone.
#pragma once
#include <cmath>
#include <cstdint>

namespace orc
{
struct V2
{
  float x, y;
};
struct V3
{
  float x, y, z;
};
struct U3
{
  uint32_t x, y, z;
};
// column-major 4x4, m[c][r] like glm::mat4
struct M4
{
  float m[4][4];
};

inline V3 v3(float a) { return {a, a, a}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
// glm: vec / scalar divides each component (no reciprocal multiply)
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glm: scalar / vec
inline V3 operator/(float s, V3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline V3 operator+(V3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }

inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator+(float s, V2 a) { return {s + a.x, s + a.y}; }
inline V2 operator-(V2 a, float s) { return {a.x - s, a.y - s}; }
inline V2 operator*(float s, V2 a) { return {s * a.x, s * a.y}; }
inline V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }
inline V2 operator*(V2 a, V2 b) { return {a.x * b.x, a.y * b.y}; }
inline V2 operator/(V2 a, V2 b) { return {a.x / b.x, a.y / b.y}; }

// glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z  (left to right)
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross
inline V3 cross(V3 x, V3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
// glm::length = sqrt(dot(v,v))
inline float length(V3 v) { return sqrtf(dot(v, v)); }
// glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// glm::min(a,b) = (b<a)?b:a ; glm::max(a,b) = (a<b)?b:a  -- also the std::min/std::max forms
inline float fmin_ref(float a, float b) { return (b < a) ? b : a; }
inline float fmax_ref(float a, float b) { return (a < b) ? b : a; }
inline V3 vmin(V3 a, V3 b) { return {fmin_ref(a.x, b.x), fmin_ref(a.y, b.y), fmin_ref(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {fmax_ref(a.x, b.x), fmax_ref(a.y, b.y), fmax_ref(a.z, b.z)}; }

// glm: mat4 * vec4 = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]*v.w, summed left to right;
// returns xyz (the reference drops w: Model.cpp:7, AABB.h:42-44)
inline V3 mulPoint(const M4& t, V3 p)
{
  float r[3];
  for (int k = 0; k < 3; ++k)
  {
    // glm (scalar path): Mov0*m[0] , Mov1*m[1], Add0 = Mul0+Mul1 ; Mul2 = m[2]*Mov2 ; Mul3 = m[3]*Mov3 ; Add1 = Mul2+Mul3 ; Add0+Add1
    float mul0 = t.m[0][k] * p.x;
    float mul1 = t.m[1][k] * p.y;
    float add0 = mul0 + mul1;
    float mul2 = t.m[2][k] * p.z;
    float mul3 = t.m[3][k] * 1.0f;
    float add1 = mul2 + mul3;
    r[k] = add0 + add1;
  }
  return {r[0], r[1], r[2]};
}
// glm: mat3 * vec3 = m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z
inline V3 mulDir(const M4& t, V3 n)
{
  return {t.m[0][0] * n.x + t.m[1][0] * n.y + t.m[2][0] * n.z, t.m[0][1] * n.x + t.m[1][1] * n.y + t.m[2][1] * n.z,
          t.m[0][2] * n.x + t.m[1][2] * n.y + t.m[2][2] * n.z};
}

// ---------------------------------------------------------------------------
// Deterministic sin/cos for the bounce direction (src/cpu/CPUScene.cpp:60).
// The reference calls the platform libm (float overloads on the platforms it
// builds on); no two libms agree bit for bit, so the oracle and the HIP kernels
// share ONE algorithm made of IEEE double operations only: Cody-Waite reduction
// by pi/2 and Taylor polynomials to x^15 / x^16 evaluated by Horner WITHOUT
// fused multiply-add, then one rounding to float.  The double result is within
// ~1e-16 of the true value, so the float is the correctly rounded one except
// for about one argument in 1e8.  tests/test_oracle_kat.py measures how often
// this differs from glibc's sinf/cosf (always <= 1 ulp).
// Valid for 0 <= x <= 8 (the caller's argument is 2*pi*u, u in [0,1]).
inline void sincos_det(float xf, float* s, float* c)
{
  const double x = (double)xf;
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double PIO2_HI = 1.57079632673412561417e+00; // first 33 bits of pi/2
  const double PIO2_LO = 6.07710050650619224932e-11; // pi/2 - PIO2_HI
  const int q = (int)(x * TWO_OVER_PI + 0.5);
  const double qd = (double)q;
  const double y = (x - qd * PIO2_HI) - qd * PIO2_LO;
  const double z = y * y;
  // sin(y) = y * (1 - z/3! + z^2/5! - ... - z^7/15!)
  double ps = -1.0 / 1307674368000.0;
  ps = ps * z + 1.0 / 6227020800.0;
  ps = ps * z - 1.0 / 39916800.0;
  ps = ps * z + 1.0 / 362880.0;
  ps = ps * z - 1.0 / 5040.0;
  ps = ps * z + 1.0 / 120.0;
  ps = ps * z - 1.0 / 6.0;
  ps = ps * z + 1.0;
  const double sy = y * ps;
  // cos(y) = 1 - z/2! + z^2/4! - ... + z^8/16!
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
  *s = (float)sr;
  *c = (float)cr;
}
} // namespace orc
