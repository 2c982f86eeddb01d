// HIPScene.h -- the Scene subclass of the MI355X backend.  It sits where the reference's CPUScene sits
// (src/cpu/CPUScene.h:4-26): createRayTracingHierarchy() is overridden to build this backend's acceleration
// data -- the flattened model tree + triangle BVH of scene/BvhFlattener -- and to hand it to the device through
// the C ABI (include/rthip.h).  Ray tracing itself (traceRay / generateIntersections / testIntersection of the
// reference) happens in the kernels, so this class has no per-ray methods.
#pragma once
#include "rthip.h"
#include "scene/BvhFlattener.h"
#include "scene/Scene.h"

namespace rt
{
class HIPScene : public Scene
{
public:
  explicit HIPScene(rthip_ctx* context) : ctx(context) {}
  ~HIPScene() override = default;
  void createRayTracingHierarchy() override;

  FlattenOptions options;
  const FlatScene& flattened() const { return flat; }
  int lastError() const { return error; }

private:
  rthip_ctx* ctx;
  FlatScene flat;
  int error = 0;
};
} // namespace rt
