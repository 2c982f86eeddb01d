#include "hip/HIPScene.h"

namespace rt
{
static_assert(sizeof(PointLight) == sizeof(rthip_point_light) && sizeof(DirectionalLight) == sizeof(rthip_dir_light),
              "light records cross the C ABI as they are");

void HIPScene::createRayTracingHierarchy()
{
  std::vector<AABB> modelBoxes;
  modelBoxes.reserve(models.size());
  for (const PModel& m : models)
    modelBoxes.push_back(m->boundingBox);
  ScenePoolsView view{&refs, &modelBoxes, &positionPool, &indicesPool, &edgesPool, &faceNormalsPool};
  flat = flattenScene(view, options);
  error = RTHIP_E_ARG;
  if (flat.info.numTris == 0 || flat.modelNodes.empty())
    return;
  if (!ctx)
  {
    error = 0; // offline: flattened only
    return;
  }
  error = rthip_upload_scene(ctx, &flat.info, flat.modelNodes.data(), flat.nodes.data(), flat.tris.data(),
                             reinterpret_cast<const rthip_dir_light*>(directionalLights.data()), (uint32_t)directionalLights.size(),
                             reinterpret_cast<const rthip_point_light*>(pointLights.data()), (uint32_t)pointLights.size());
}
} // namespace rt
