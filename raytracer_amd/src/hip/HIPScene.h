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
  // shading beyond the reference's CPU path; both off = that path's hard-coded BRDF inputs and lights
  struct ShadingOptions
  {
    bool materials = false;    // evaluate each model's Material at the hit (rthip_upload_materials)
    bool pointShadows = false; // point lights cast shadow rays as in src/metal/Compute.metal:238-254 (rthip_set_point_shadows)
  } shading;
  // what rthip_upload_materials was given (empty unless shading.materials)
  struct FlatShading
  {
    std::vector<rtflat_shade> shade;         // one per triangle, indexed by triId
    std::vector<rtflat_material> materials;  // one per model, in insertion order
    std::vector<float> texels;               // RGB triples of every texture, each texture once
  };
  const FlatShading& flattenedShading() const { return flatShading; }
  const FlatScene& flattened() const { return flat; }
  int lastError() const { return error; }

private:
  rthip_ctx* ctx;
  void buildShadingTables();
  FlatScene flat;
  FlatShading flatShading;
  int error = 0;
};
} // namespace rt
