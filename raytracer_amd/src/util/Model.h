// Model.h -- one triangle mesh as a loader hands it to the scene (reference: src/util/Model.h:13-25).
#pragma once
#include "scene/AABB.h"
#include "util/Material.h"
#include <memory>
#include <vector>

namespace rt
{
class Model
{
public:
  AABB boundingBox;                // bounds of `positions` as loaded (ModelLoader.cpp:18,32,39)
  std::vector<vec3> positions;
  std::vector<vec2> texCoords;     // one per position; (0,0) when the file has none
  std::vector<vec3> normals;       // one per position; unused by this backend's shading (face normals only)
  std::vector<uvec3> indices;      // one entry per triangle
  std::vector<vec3> edges;         // filled by transform(): 2 per triangle, e0 = p1-p0, e1 = p2-p0
  std::vector<vec3> faceNormals;   // filled by transform(): cross(e0,e1), NOT normalised yet
  Material material;               // read only when the scene option "materials" is set (util/Material.h)

  // builds a model from raw arrays and accumulates the bounding box the way the reference's loader does
  static std::unique_ptr<Model> fromArrays(const float* positions, uint32_t numPositions, const float* texCoords,
                                           const uint32_t* indices, uint32_t numTriangles);
  // Model.cpp:3-26
  void transform(const mat4& matrix);
};
using PModel = std::unique_ptr<Model>;
} // namespace rt
