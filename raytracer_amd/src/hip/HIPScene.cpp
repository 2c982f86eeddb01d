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
    error = options.deviceBuild ? RTHIP_E_NODEVICE : 0; // offline: flattened only (a device build needs the device)
    return;
  }
  const auto* dl = reinterpret_cast<const rthip_dir_light*>(directionalLights.data());
  const auto* pl = reinterpret_cast<const rthip_point_light*>(pointLights.data());
  if (options.deviceBuild)
  {
    // the flattener stopped before the triangle-level build: the device builds it from the records in pool order, and the
    // host copy of the scene is completed from what it built (the oracle-side tests and the CPU baseline walk that copy)
    float ms = 0.0f;
    error = rthip_upload_scene_build(ctx, &flat.info, flat.modelNodes.data(), flat.tris.data(), options.maxLeafTriangles, dl,
                                     (uint32_t)directionalLights.size(), pl, (uint32_t)pointLights.size(), &ms);
    if (error)
      return;
    flat.deviceBuildMs = ms;
    flat.nodes.resize(flat.info.numNodes);
    error = rthip_download_scene(ctx, flat.nodes.data(), (uint32_t)flat.nodes.size(), flat.tris.data(), (uint32_t)flat.tris.size());
    return;
  }
  error = rthip_upload_scene(ctx, &flat.info, flat.modelNodes.data(), flat.nodes.data(), flat.tris.data(), dl, (uint32_t)directionalLights.size(), pl,
                             (uint32_t)pointLights.size());
}
} // namespace rt
