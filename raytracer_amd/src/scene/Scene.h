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

// The pools generate() fills are the ones the reference's intersector indexes (src/scene/Scene.h:45-51); the member names
// are kept because the flattener's citations refer to them.  Everything is laid out for the HIP backend's needs: the
// hierarchy hook comes first, data members are documented with what reads them.
class Scene
{
public:
  Scene() = default;
  virtual ~Scene() = default;

  // ---- set-up (same five mutators as the reference's Scene) ---------------------------------------------------------
  void addDirectionalLight(DirectionalLight dir) { directionalLights.push_back(dir); }
  void addPointLight(PointLight point) { pointLights.push_back(point); }
  void addModels(std::vector<PModel> models, const mat4& transform); // every model with the same transform
  void addModel(PModel model, const mat4& transform);                // Model::transform is applied here, once
  // Concatenates the models into the pools below (src/scene/Scene.cpp:20-46), then calls createRayTracingHierarchy().
  void generate();

  uint32_t getNumTriangles() const { return (uint32_t)indicesPool.size(); }
  uint32_t getNumPointLights() const { return (uint32_t)pointLights.size(); }
  uint32_t getNumDirLights() const { return (uint32_t)directionalLights.size(); }

protected:
  // A backend builds its acceleration structure from the pools here (CPUScene: the model tree; HIPScene: the flattened
  // arrays of include/rtflat.h, uploaded to the device).
  virtual void createRayTracingHierarchy() = 0;

  std::vector<PModel> models;               // as added, already transformed
  std::vector<ModelReference> refs;         // one per model: where its slice of each pool starts
  std::vector<uvec3> indicesPool;           // model-local vertex indices, one uvec3 per triangle
  std::vector<vec3> positionPool;
  std::vector<vec3> edgesPool;              // two per triangle: p1-p0, p2-p0 (see the indexing note in DESIGN.md section 3)
  std::vector<vec3> faceNormalsPool;        // one per triangle, normalised cross(e0, e1)
  std::vector<vec3> normalsPool;            // per-vertex attributes: carried, never read on this path
  std::vector<vec2> texCoordsPool;
  std::vector<DirectionalLight> directionalLights;
  std::vector<PointLight> pointLights;
};
} // namespace rt
