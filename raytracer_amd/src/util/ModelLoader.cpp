#include "util/ModelLoader.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace rt
{
std::vector<PModel> ModelLoader::loadModel(std::string_view filename)
{
  std::vector<PModel> result;
  const std::string path(filename);
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f)
  {
    std::fprintf(stderr, "ModelLoader: cannot open %s\n", path.c_str());
    return result;
  }
  auto model = std::make_unique<Model>();
  char line[1024];
  std::vector<long> corner;
  while (std::fgets(line, sizeof(line), f))
  {
    if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t'))
    {
      char* p = line + 2;
      const float x = std::strtof(p, &p), y = std::strtof(p, &p), z = std::strtof(p, &p);
      model->positions.push_back(vec3{x, y, z});
    }
    else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t'))
    {
      corner.clear();
      char* p = line + 2;
      for (;;)
      {
        while (*p == ' ' || *p == '\t')
          ++p;
        if (*p == '\0' || *p == '\n' || *p == '\r')
          break;
        char* end;
        long idx = std::strtol(p, &end, 10);
        if (end == p)
          break;
        if (idx < 0)
          idx = (long)model->positions.size() + idx + 1; // relative index
        corner.push_back(idx - 1);
        p = end;
        while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') // skip /vt/vn
          ++p;
      }
      for (size_t k = 2; k < corner.size(); ++k)
        model->indices.push_back(uvec3{(uint32_t)corner[0], (uint32_t)corner[k - 1], (uint32_t)corner[k]});
    }
  }
  std::fclose(f);
  // drop faces that point outside the vertex list instead of reading out of bounds later
  const uint32_t nv = (uint32_t)model->positions.size();
  std::vector<uvec3> kept;
  kept.reserve(model->indices.size());
  for (const uvec3& t : model->indices)
    if (t.x < nv && t.y < nv && t.z < nv)
      kept.push_back(t);
  model->indices.swap(kept);
  model->texCoords.assign(nv, vec2{0.0f, 0.0f});
  model->normals.assign(nv, vec3{0.0f, 0.0f, 0.0f});
  for (const vec3& p : model->positions)
    model->boundingBox.adjust(p);
  if (!model->indices.empty())
    result.push_back(std::move(model));
  return result;
}
} // namespace rt
