// AABB.h -- axis-aligned box used on the host side of the scene (model bounds, BVH build).
// Mirrors what the reference's AABB offers (src/scene/AABB.h:8-70); the ray/box slab test of the
// reference (AABB.h:47-60) lives in the device code (csrc/rt_device.h refSlab) because only kernels
// trace rays in this backend.
#pragma once
#include "util/Math.h"
#include <cfloat>

namespace rt
{
struct AABB
{
  vec3 min{FLT_MAX, FLT_MAX, FLT_MAX};
  vec3 max{-FLT_MAX, -FLT_MAX, -FLT_MAX};

  // 2(dx*dy + dy*dz + dz*dx), AABB.h:12-16
  float surfaceArea() const
  {
    const float dx = max.x - min.x, dy = max.y - min.y, dz = max.z - min.z;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
  }
  // AABB.h:17-26
  void adjust(vec3 p)
  {
    min = vec3{minRef(min.x, p.x), minRef(min.y, p.y), minRef(min.z, p.z)};
    max = vec3{maxRef(max.x, p.x), maxRef(max.y, p.y), maxRef(max.z, p.z)};
  }
  // AABB.h:27-46: bounds of the eight transformed corners
  void transform(const mat4& m)
  {
    const vec3 lo = min, hi = max;
    *this = AABB{};
    for (int corner = 0; corner < 8; ++corner)
      adjust(transformPoint(m, vec3{(corner & 4) ? hi.x : lo.x, (corner & 2) ? hi.y : lo.y, (corner & 1) ? hi.z : lo.z}));
  }
  // AABB.h:61-69
  static AABB combine(const AABB& a, const AABB& b)
  {
    AABB r;
    r.min = vec3{minRef(a.min.x, b.min.x), minRef(a.min.y, b.min.y), minRef(a.min.z, b.min.z)};
    r.max = vec3{maxRef(a.max.x, b.max.x), maxRef(a.max.y, b.max.y), maxRef(a.max.z, b.max.z)};
    return r;
  }
};
} // namespace rt
