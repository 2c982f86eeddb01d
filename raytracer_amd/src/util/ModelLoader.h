// ModelLoader.h -- file loading behind the reference's entry point ModelLoader::loadModel
// (src/util/ModelLoader.h, ModelLoader.cpp:8-43).  The reference delegates to assimp, which is not available
// here; this loader reads Wavefront OBJ (v / f records, polygons fan-triangulated like aiProcess_Triangulate does
// for convex faces; one Model per file: res/models/stanford-bunny.obj) and binary glTF 2.0 (.glb; one Model per mesh
// primitive in file order, node transforms ignored as the reference ignores them: res/models/box.glb, town_hall.glb).
// Binary FBX 7.x geometry (cube.fbx, box.fbx; zlib arrays) is read as one vertex per polygon corner, fan-triangulated.
#pragma once
#include "util/Model.h"
#include <string_view>
#include <vector>

namespace rt
{
class ModelLoader
{
public:
  static std::vector<PModel> loadModel(std::string_view filename);
};
} // namespace rt
