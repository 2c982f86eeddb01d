// ModelLoader.h -- file loading behind the reference's entry point ModelLoader::loadModel
// (src/util/ModelLoader.h, ModelLoader.cpp:8-43).  The reference delegates to assimp, which is not available
// here; this loader reads Wavefront OBJ (v / f records, polygons fan-triangulated like aiProcess_Triangulate does
// for convex faces), enough for res/models/stanford-bunny.obj.  One Model per file.
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
