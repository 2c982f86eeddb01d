#include "scene/Scene.h"

namespace rt
{
// src/scene/Scene.cpp:3-16: a model is transformed once, on insertion, and then owned by the scene
void Scene::addModel(PModel model, const mat4& transform)
{
  model->transform(transform);
  models.push_back(std::move(model));
}

void Scene::addModels(std::vector<PModel> incoming, const mat4& transform)
{
  for (PModel& m : incoming)
    addModel(std::move(m), transform);
}

// src/scene/Scene.cpp:18-46.  Pools are appended model by model; a ModelReference remembers where each
// model's vertices and triangles start.  Face normals are normalised at this point (Scene.cpp:41).
void Scene::generate()
{
  size_t totalPositions = positionPool.size(), totalTriangles = indicesPool.size();
  for (const PModel& m : models)
  {
    totalPositions += m->positions.size();
    totalTriangles += m->indices.size();
  }
  positionPool.reserve(totalPositions);
  texCoordsPool.reserve(totalPositions);
  normalsPool.reserve(totalPositions);
  indicesPool.reserve(totalTriangles);
  edgesPool.reserve(2 * totalTriangles);
  faceNormalsPool.reserve(totalTriangles);

  for (const PModel& m : models)
  {
    ModelReference ref;
    ref.positionOffset = (uint32_t)positionPool.size();
    ref.numPositions = (uint32_t)m->positions.size();
    ref.indicesOffset = (uint32_t)indicesPool.size();
    ref.numIndices = (uint32_t)m->indices.size();

    positionPool.insert(positionPool.end(), m->positions.begin(), m->positions.end());
    texCoordsPool.insert(texCoordsPool.end(), m->texCoords.begin(), m->texCoords.end());
    normalsPool.insert(normalsPool.end(), m->normals.begin(), m->normals.end());

    indicesPool.insert(indicesPool.end(), m->indices.begin(), m->indices.end());
    edgesPool.insert(edgesPool.end(), m->edges.begin(), m->edges.end());
    for (const vec3& fn : m->faceNormals)
      faceNormalsPool.push_back(normalize(fn));
    refs.push_back(ref);
  }
  createRayTracingHierarchy();
}
} // namespace rt
