#include "hip/HIPScene.h"
#include <map>

namespace rt
{
static_assert(sizeof(PointLight) == sizeof(rthip_point_light) && sizeof(DirectionalLight) == sizeof(rthip_dir_light),
              "light records cross the C ABI as they are");

// The shading inputs of include/rtflat.h from the pools and the models' materials: per triangle the texture coordinates of
// its corners -- texCoordsPool[positionOffset + idx.x / .y / .z], what the reference's intersector reads at
// src/cpu/CPUScene.cpp:201-203 -- and its model's index into the material table.
void HIPScene::buildShadingTables()
{
  FlatShading& fs = flatShading;
  fs = FlatShading{};
  fs.shade.resize(indicesPool.size());
  std::map<const Texture*, uint32_t> firstTexel; // a texture shared by several materials is stored once
  auto place = [&](const PTexture& t, uint32_t& first, uint32_t& w, uint32_t& h) {
    first = w = h = 0;
    if (!t || t->width <= 0 || t->height <= 0 || t->textureData.size() < (size_t)t->width * (size_t)t->height)
      return;
    auto it = firstTexel.find(t.get());
    if (it == firstTexel.end())
    {
      it = firstTexel.emplace(t.get(), (uint32_t)(fs.texels.size() / 3)).first;
      const size_t n = (size_t)t->width * (size_t)t->height;
      fs.texels.reserve(fs.texels.size() + 3 * n);
      for (size_t k = 0; k < n; ++k)
      {
        fs.texels.push_back(t->textureData[k].x);
        fs.texels.push_back(t->textureData[k].y);
        fs.texels.push_back(t->textureData[k].z);
      }
    }
    first = it->second, w = (uint32_t)t->width, h = (uint32_t)t->height;
  };
  for (size_t k = 0; k < models.size(); ++k)
  {
    const Material& m = models[k]->material;
    rtflat_material out{};
    out.kind = m.kind == MaterialKind::ReferenceStub ? RTFLAT_MAT_STUB : RTFLAT_MAT_FACTORS;
    out.baseColor[0] = m.baseColor.x, out.baseColor[1] = m.baseColor.y, out.baseColor[2] = m.baseColor.z;
    out.emissive[0] = m.emissive.x, out.emissive[1] = m.emissive.y, out.emissive[2] = m.emissive.z;
    place(m.albedoTexture, out.albedoTexel, out.albedoW, out.albedoH);
    place(m.emissiveTexture, out.emissiveTexel, out.emissiveW, out.emissiveH);
    fs.materials.push_back(out);
    const ModelReference& ref = refs[k];
    for (uint32_t i = 0; i < ref.numIndices; ++i)
    {
      const uvec3 idx = indicesPool[ref.indicesOffset + i];
      rtflat_shade& sh = fs.shade[ref.indicesOffset + i];
      const vec2 t0 = texCoordsPool[ref.positionOffset + idx.x], t1 = texCoordsPool[ref.positionOffset + idx.y], t2 = texCoordsPool[ref.positionOffset + idx.z];
      sh.uv[0][0] = t0.x, sh.uv[0][1] = t0.y;
      sh.uv[1][0] = t1.x, sh.uv[1][1] = t1.y;
      sh.uv[2][0] = t2.x, sh.uv[2][1] = t2.y;
      sh.material = (uint32_t)k;
      sh.pad = 0;
    }
  }
}

// uploads what the options ask for beyond the geometry; the scene itself is on the device by now
static int uploadShading(rthip_ctx* ctx, const HIPScene::ShadingOptions& shading, const HIPScene::FlatShading& fs)
{
  int rc = 0;
  if (shading.materials)
    rc = rthip_upload_materials(ctx, fs.shade.data(), (uint32_t)fs.shade.size(), fs.materials.data(), (uint32_t)fs.materials.size(),
                                fs.texels.empty() ? nullptr : fs.texels.data(), fs.texels.size() / 3);
  if (!rc)
    rc = rthip_set_point_shadows(ctx, shading.pointShadows ? 1 : 0);
  return rc;
}

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
  if (shading.materials)
    buildShadingTables();
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
    error = rthip_set_build_quality(ctx, options.deviceBuildQuality);
    if (error)
      return;
    error = rthip_upload_scene_build(ctx, &flat.info, flat.modelNodes.data(), flat.tris.data(), options.maxLeafTriangles, dl,
                                     (uint32_t)directionalLights.size(), pl, (uint32_t)pointLights.size(), &ms);
    if (error)
      return;
    flat.deviceBuildMs = ms;
    flat.nodes.resize(flat.info.numNodes);
    error = rthip_download_scene(ctx, flat.nodes.data(), (uint32_t)flat.nodes.size(), flat.tris.data(), (uint32_t)flat.tris.size());
    if (!error)
      error = uploadShading(ctx, shading, flatShading);
    return;
  }
  error = rthip_upload_scene(ctx, &flat.info, flat.modelNodes.data(), flat.nodes.data(), flat.tris.data(), dl, (uint32_t)directionalLights.size(), pl,
                             (uint32_t)pointLights.size());
  if (!error)
    error = uploadShading(ctx, shading, flatShading);
}
} // namespace rt
