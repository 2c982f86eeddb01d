#include "util/Model.h"

namespace rt
{
PModel Model::fromArrays(const float* pos, uint32_t numPositions, const float* uv, const uint32_t* idx, uint32_t numTriangles)
{
  auto m = std::make_unique<Model>();
  m->positions.resize(numPositions);
  m->texCoords.resize(numPositions, vec2{0.0f, 0.0f});
  m->normals.resize(numPositions, vec3{0.0f, 0.0f, 0.0f});
  for (uint32_t v = 0; v < numPositions; ++v)
  {
    m->positions[v] = vec3{pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]};
    if (uv)
      m->texCoords[v] = vec2{uv[2 * v], uv[2 * v + 1]};
    m->boundingBox.adjust(m->positions[v]);
  }
  m->indices.resize(numTriangles);
  for (uint32_t t = 0; t < numTriangles; ++t)
    m->indices[t] = uvec3{idx[3 * t], idx[3 * t + 1], idx[3 * t + 2]};
  return m;
}

// Reference Model::transform (src/util/Model.cpp:3-26): move positions/normals/box by the matrix, THEN derive
// the per-triangle edge vectors and the unnormalised face normal the intersector will read.
void Model::transform(const mat4& matrix)
{
  for (vec3& p : positions)
    p = transformPoint(matrix, p);
  for (vec3& n : normals)
    n = transformDirection(matrix, n);
  boundingBox.transform(matrix);

  const size_t numTriangles = indices.size();
  edges.reserve(edges.size() + 2 * numTriangles);
  faceNormals.reserve(faceNormals.size() + numTriangles);
  for (size_t t = 0; t < numTriangles; ++t)
  {
    const vec3 a = positions[indices[t].x];
    const vec3 e0 = positions[indices[t].y] - a;
    const vec3 e1 = positions[indices[t].z] - a;
    edges.push_back(e0);
    edges.push_back(e1);
    faceNormals.push_back(cross(e0, e1));
  }
}
} // namespace rt
