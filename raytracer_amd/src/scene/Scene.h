// Scene.h -- backend-independent scene: lights, models and the flattened geometry pools.
// Interface parity with the reference's Scene (src/scene/Scene.h:30-61): the same five mutators, generate(),
// and the virtual createRayTracingHierarchy() hook a backend overrides to build its acceleration structure
// from the pools.
#pragma once
#include "util/Model.h"
#include <vector>

namespace rt
{
struct ModelReference // src/scene/Scene.h:6-12
{
  uint32_t positionOffset = 0;
  uint32_t numPositions = 0;
  uint32_t indicesOffset = 0;
  uint32_t numIndices = 0;
};

struct PointLight // src/scene/Scene.h:14-20 (32 bytes)
{
  vec3 position{0, 0, 0};
  float pad = 0;
  vec3 color{1, 1, 1};
  float attenuation = 1;
};

struct DirectionalLight // src/scene/Scene.h:22-28 (32 bytes)
{
  vec3 direction{0, 1, 0};
  float pad = 0;
  vec3 color{1, 1, 1};
  float pad1 = 0;
};

class Scene
{
public:
  Scene() = default;
  virtual ~Scene() = default;
  void addPointLight(PointLight point) { pointLights.push_back(point); }
  void addDirectionalLight(DirectionalLight dir) { directionalLights.push_back(dir); }
  void addModel(PModel model, const mat4& transform);
  void addModels(std::vector<PModel> models, const mat4& transform);
  // concatenates every model into the pools, then asks the backend for its hierarchy
  void generate();

  uint32_t getNumDirLights() const { return (uint32_t)directionalLights.size(); }
  uint32_t getNumPointLights() const { return (uint32_t)pointLights.size(); }
  uint32_t getNumTriangles() const { return (uint32_t)indicesPool.size(); }

protected:
  std::vector<ModelReference> refs;
  std::vector<vec3> positionPool;
  std::vector<vec2> texCoordsPool;
  std::vector<vec3> normalsPool;
  std::vector<uvec3> indicesPool;
  std::vector<vec3> edgesPool;       // 2 per triangle
  std::vector<vec3> faceNormalsPool; // normalised

  std::vector<PointLight> pointLights;
  std::vector<DirectionalLight> directionalLights;

  std::vector<PModel> models;

  virtual void createRayTracingHierarchy() = 0;
};
} // namespace rt
