#include "util/ModelLoader.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cctype>
#include <cstddef>
#include <string>
#include <utility>
#include <zlib.h>

namespace rt
{
namespace
{
std::vector<PModel> loadObj(const std::string& path)
{
  std::vector<PModel> result;
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

// ---------------------------------------------------------------------------------------------------------------------
// Binary glTF 2.0 (.glb: res/models/box.glb, town_hall.glb).  Like the reference's loop over aiScene::mMeshes
// (ModelLoader.cpp:14-41) this returns one Model per mesh primitive, in file order, with the vertices exactly as stored
// (POSITION accessor order) and NO node transform applied -- the reference never looks at the node hierarchy.  Only what
// that loop reads is decoded: positions, triangle indices, the first texture-coordinate set.
struct Json
{
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  double number = 0.0;
  bool boolean = false;
  std::string string;
  std::vector<Json> items;                          // Array
  std::vector<std::pair<std::string, Json>> fields; // Object
  const Json* get(const char* key) const
  {
    for (const auto& f : fields)
      if (f.first == key)
        return &f.second;
    return nullptr;
  }
  long integer(const char* key, long fallback) const
  {
    const Json* j = get(key);
    return (j && j->kind == Number) ? (long)j->number : fallback;
  }
};

struct JsonParser
{
  const char* p;
  const char* end;
  bool ok = true;
  void skip()
  {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r'))
      ++p;
  }
  bool eat(char c)
  {
    skip();
    if (p < end && *p == c)
    {
      ++p;
      return true;
    }
    return false;
  }
  std::string parseString()
  {
    std::string out;
    if (!eat('"'))
    {
      ok = false;
      return out;
    }
    while (p < end && *p != '"')
    {
      if (*p == '\\' && p + 1 < end)
      {
        ++p;
        switch (*p)
        {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'u': p += (end - p > 4) ? 4 : 0; out.push_back('?'); break; // names are never needed as text here
        default: out.push_back(*p); break;
        }
        ++p;
      }
      else
        out.push_back(*p++);
    }
    if (p >= end)
      ok = false;
    else
      ++p;
    return out;
  }
  Json parse(int depth = 0)
  {
    Json j;
    skip();
    if (!ok || p >= end || depth > 64)
    {
      ok = false;
      return j;
    }
    if (*p == '{')
    {
      ++p;
      j.kind = Json::Object;
      if (eat('}'))
        return j;
      do
      {
        std::string key = parseString();
        if (!ok || !eat(':'))
        {
          ok = false;
          return j;
        }
        j.fields.emplace_back(std::move(key), parse(depth + 1));
      } while (ok && eat(','));
      if (!eat('}'))
        ok = false;
    }
    else if (*p == '[')
    {
      ++p;
      j.kind = Json::Array;
      if (eat(']'))
        return j;
      do
        j.items.push_back(parse(depth + 1));
      while (ok && eat(','));
      if (!eat(']'))
        ok = false;
    }
    else if (*p == '"')
    {
      j.kind = Json::String;
      j.string = parseString();
    }
    else if (end - p >= 4 && std::strncmp(p, "true", 4) == 0)
      j.kind = Json::Bool, j.boolean = true, p += 4;
    else if (end - p >= 5 && std::strncmp(p, "false", 5) == 0)
      j.kind = Json::Bool, p += 5;
    else if (end - p >= 4 && std::strncmp(p, "null", 4) == 0)
      p += 4;
    else
    {
      const std::string text(p, (size_t)std::min<ptrdiff_t>(end - p, 63));
      char* stop = nullptr;
      j.number = std::strtod(text.c_str(), &stop);
      if (stop == text.c_str())
        ok = false;
      else
        j.kind = Json::Number, p += stop - text.c_str();
    }
    return j;
  }
};

uint32_t readU32(const unsigned char* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }

// a typed view of one accessor over the GLB's binary chunk; count == 0 when it cannot be used
struct AccessorView
{
  const unsigned char* base = nullptr;
  size_t stride = 0, count = 0;
  long componentType = 0;
  int components = 0;
};

AccessorView viewAccessor(const Json& root, long index, const std::vector<unsigned char>& bin)
{
  AccessorView v;
  const Json* accessors = root.get("accessors");
  const Json* views = root.get("bufferViews");
  if (!accessors || !views || index < 0 || (size_t)index >= accessors->items.size())
    return v;
  const Json& acc = accessors->items[(size_t)index];
  if (acc.get("sparse"))
    return v; // not produced by the exporters behind res/models; refused rather than mis-read
  const long bv = acc.integer("bufferView", -1);
  if (bv < 0 || (size_t)bv >= views->items.size())
    return v;
  const Json& view = views->items[(size_t)bv];
  if (view.integer("buffer", 0) != 0)
    return v; // a .glb keeps everything in buffer 0, the BIN chunk
  const Json* type = acc.get("type");
  const std::string t = type ? type->string : "";
  v.components = t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : 0;
  v.componentType = acc.integer("componentType", 0);
  const size_t compBytes = (v.componentType == 5120 || v.componentType == 5121)   ? 1
                           : (v.componentType == 5122 || v.componentType == 5123) ? 2
                           : (v.componentType == 5125 || v.componentType == 5126) ? 4
                                                                                  : 0;
  if (!v.components || !compBytes)
    return v;
  const size_t elem = compBytes * (size_t)v.components;
  const size_t offset = (size_t)view.integer("byteOffset", 0) + (size_t)acc.integer("byteOffset", 0);
  const long strideField = view.integer("byteStride", 0);
  v.stride = strideField > 0 ? (size_t)strideField : elem;
  const size_t count = (size_t)acc.integer("count", 0);
  if (count == 0 || v.stride < elem || offset > bin.size() || (count - 1) * v.stride + elem > bin.size() - offset)
    return v;
  v.base = bin.data() + offset;
  v.count = count;
  return v;
}

std::vector<PModel> loadGlb(const std::string& path)
{
  std::vector<PModel> result;
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f)
  {
    std::fprintf(stderr, "ModelLoader: cannot open %s\n", path.c_str());
    return result;
  }
  std::vector<unsigned char> file;
  unsigned char buf[65536];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0)
    file.insert(file.end(), buf, buf + got);
  std::fclose(f);
  if (file.size() < 20 || readU32(file.data()) != 0x46546C67u /* "glTF" */ || readU32(file.data() + 4) != 2u)
  {
    std::fprintf(stderr, "ModelLoader: %s is not a binary glTF 2.0 file\n", path.c_str());
    return result;
  }
  std::string jsonText;
  std::vector<unsigned char> bin;
  size_t at = 12;
  const size_t total = std::min<size_t>(file.size(), readU32(file.data() + 8));
  while (at + 8 <= total)
  {
    const size_t len = readU32(file.data() + at);
    const uint32_t type = readU32(file.data() + at + 4);
    at += 8;
    if (len > total - at)
      break;
    if (type == 0x4E4F534Au /* JSON */ && jsonText.empty())
      jsonText.assign((const char*)file.data() + at, len);
    else if (type == 0x004E4942u /* BIN */ && bin.empty())
      bin.assign(file.begin() + (ptrdiff_t)at, file.begin() + (ptrdiff_t)(at + len));
    at += (len + 3u) & ~(size_t)3u;
  }
  JsonParser parser{jsonText.data(), jsonText.data() + jsonText.size()};
  const Json root = parser.parse();
  const Json* meshes = root.get("meshes");
  if (!parser.ok || root.kind != Json::Object || !meshes)
  {
    std::fprintf(stderr, "ModelLoader: %s: malformed glTF JSON\n", path.c_str());
    return result;
  }
  for (const Json& mesh : meshes->items)
  {
    const Json* prims = mesh.get("primitives");
    if (!prims)
      continue;
    for (const Json& prim : prims->items)
    {
      if (prim.integer("mode", 4) != 4)
        continue; // triangles only: points and lines have no faces for aiProcess_Triangulate to keep either
      const Json* attrs = prim.get("attributes");
      if (!attrs)
        continue;
      const AccessorView pos = viewAccessor(root, attrs->integer("POSITION", -1), bin);
      if (!pos.count || pos.componentType != 5126 || pos.components != 3)
        continue;
      auto model = std::make_unique<Model>();
      model->positions.resize(pos.count);
      for (size_t v = 0; v < pos.count; ++v)
      {
        float xyz[3];
        std::memcpy(xyz, pos.base + v * pos.stride, 12);
        model->positions[v] = vec3{xyz[0], xyz[1], xyz[2]};
        model->boundingBox.adjust(model->positions[v]);
      }
      model->texCoords.assign(pos.count, vec2{0.0f, 0.0f});
      const AccessorView uv = viewAccessor(root, attrs->integer("TEXCOORD_0", -1), bin);
      if (uv.count == pos.count && uv.componentType == 5126 && uv.components == 2)
        for (size_t v = 0; v < pos.count; ++v)
        {
          float st[2];
          std::memcpy(st, uv.base + v * uv.stride, 8);
          model->texCoords[v] = vec2{st[0], st[1]};
        }
      model->normals.assign(pos.count, vec3{0.0f, 0.0f, 0.0f}); // never read on the hot path (face normals are recomputed)
      std::vector<uint32_t> idx;
      const long indexAccessor = prim.integer("indices", -1);
      if (indexAccessor >= 0)
      {
        const AccessorView iv = viewAccessor(root, indexAccessor, bin);
        if (!iv.count || iv.components != 1)
          continue;
        idx.resize(iv.count);
        for (size_t k = 0; k < iv.count; ++k)
        {
          const unsigned char* e = iv.base + k * iv.stride;
          idx[k] = iv.componentType == 5121 ? e[0] : iv.componentType == 5123 ? (uint32_t)(e[0] | (e[1] << 8)) : iv.componentType == 5125 ? readU32(e) : 0xffffffffu;
        }
      }
      else
      {
        idx.resize(pos.count);
        for (size_t k = 0; k < pos.count; ++k)
          idx[k] = (uint32_t)k;
      }
      for (size_t k = 0; k + 2 < idx.size(); k += 3)
        if (idx[k] < pos.count && idx[k + 1] < pos.count && idx[k + 2] < pos.count)
          model->indices.push_back(uvec3{idx[k], idx[k + 1], idx[k + 2]});
      // the primitive's material: the two factors of glTF 2.0 that map onto Material (pbrMetallicRoughness.baseColorFactor, RGB of
      // RGBA, default 1; emissiveFactor, default 0).  Image textures (PNG / JPEG inside the file) are not decoded.
      model->material.kind = MaterialKind::Factors;
      model->material.baseColor = vec3{1.0f, 1.0f, 1.0f};
      model->material.emissive = vec3{0.0f, 0.0f, 0.0f};
      const Json* materials = root.get("materials");
      const long mat = prim.integer("material", -1);
      if (materials && mat >= 0 && (size_t)mat < materials->items.size())
      {
        auto rgb = [](const Json* a, vec3& out) {
          if (a && a->kind == Json::Array && a->items.size() >= 3 && a->items[0].kind == Json::Number && a->items[1].kind == Json::Number &&
              a->items[2].kind == Json::Number)
            out = vec3{(float)a->items[0].number, (float)a->items[1].number, (float)a->items[2].number};
        };
        const Json& m = materials->items[(size_t)mat];
        if (const Json* pbr = m.get("pbrMetallicRoughness"))
          rgb(pbr->get("baseColorFactor"), model->material.baseColor);
        rgb(m.get("emissiveFactor"), model->material.emissive);
      }
      if (!model->indices.empty())
        result.push_back(std::move(model));
    }
  }
  return result;
}

// ---------------------------------------------------------------------------------------------------------------------
// Binary FBX 7.x (res/models/cube.fbx, box.fbx).  Only what the reference's loop over aiScene::mMeshes reads is decoded:
// for every Objects/Geometry node the control points (`Vertices`, doubles) and the polygon list (`PolygonVertexIndex`, the
// last corner of a polygon stored as ~index).  The result mimics what assimp's FBX importer + aiProcess_Triangulate hand
// the reference: one vertex PER POLYGON CORNER in polygon order (no sharing), polygons fan-triangulated from their first
// corner, no node transform, no unit scaling.  (assimp is absent, so that it emits exactly this order is an assumption
// read off its documented behaviour -- "parity unpinned" for ingestion.)
struct FbxArray
{
  char type = 0;
  std::vector<unsigned char> raw; // decoded little-endian elements
  size_t count = 0;
};

struct FbxReader
{
  const std::vector<unsigned char>& d;
  bool wide; // version >= 7500: 64-bit node header fields
  bool ok = true;
  std::vector<PModel>* out;

  uint64_t field(size_t& at)
  {
    const size_t n = wide ? 8 : 4;
    if (at + n > d.size())
    {
      ok = false;
      return 0;
    }
    uint64_t v = 0;
    for (size_t k = 0; k < n; ++k)
      v |= (uint64_t)d[at + k] << (8 * k);
    at += n;
    return v;
  }
  // reads one property at `at`; fills `arr` when it is an array property
  void property(size_t& at, FbxArray* arr)
  {
    if (at >= d.size())
    {
      ok = false;
      return;
    }
    const char t = (char)d[at++];
    size_t scalar = 0;
    switch (t)
    {
    case 'Y': scalar = 2; break;
    case 'C': scalar = 1; break;
    case 'I': case 'F': scalar = 4; break;
    case 'D': case 'L': scalar = 8; break;
    default: break;
    }
    if (scalar)
    {
      at += scalar;
      return;
    }
    if (t == 'S' || t == 'R')
    {
      if (at + 4 > d.size())
      {
        ok = false;
        return;
      }
      at += 4 + readU32(d.data() + at);
      return;
    }
    const size_t elem = (t == 'f' || t == 'i') ? 4 : (t == 'd' || t == 'l') ? 8 : (t == 'b') ? 1 : 0;
    if (!elem || at + 12 > d.size())
    {
      ok = false;
      return;
    }
    const size_t count = readU32(d.data() + at), encoding = readU32(d.data() + at + 4), stored = readU32(d.data() + at + 8);
    at += 12;
    if (stored > d.size() - at)
    {
      ok = false;
      return;
    }
    if (arr)
    {
      arr->type = t;
      arr->count = count;
      arr->raw.resize(count * elem);
      if (encoding == 0)
      {
        if (stored != count * elem)
          ok = false;
        else
          std::memcpy(arr->raw.data(), d.data() + at, stored);
      }
      else
      {
        uLongf got = (uLongf)arr->raw.size();
        if (uncompress(arr->raw.data(), &got, d.data() + at, (uLong)stored) != Z_OK || got != arr->raw.size())
          ok = false;
      }
    }
    at += stored;
  }
  // walks the node at `at`; returns the offset after it (0 at a null record or on error)
  size_t node(size_t at, int depth, FbxArray* vertices, FbxArray* polygons)
  {
    const size_t start = at;
    const uint64_t end = field(at), numProps = field(at), propBytes = field(at);
    (void)propBytes;
    if (!ok || at >= d.size())
      return 0;
    const size_t nameLen = d[at++];
    if (end == 0)
      return 0; // null record: end of a child list
    if (end > d.size() || end <= start || at + nameLen > d.size() || depth > 32)
    {
      ok = false;
      return 0;
    }
    const std::string name((const char*)d.data() + at, nameLen);
    at += nameLen;
    const bool isVertices = vertices && name == "Vertices", isPolygons = polygons && name == "PolygonVertexIndex";
    for (uint64_t k = 0; ok && k < numProps; ++k)
      property(at, (k == 0 && isVertices) ? vertices : (k == 0 && isPolygons) ? polygons : nullptr);
    if (!ok)
      return 0;
    FbxArray geoVertices, geoPolygons;
    const bool isGeometry = name == "Geometry";
    while (ok && at + (wide ? 25u : 13u) <= end)
    {
      const size_t next = node(at, depth + 1, isGeometry ? &geoVertices : vertices, isGeometry ? &geoPolygons : polygons);
      if (!next)
        break;
      at = next;
    }
    if (ok && isGeometry)
      emit(geoVertices, geoPolygons);
    return ok ? (size_t)end : 0;
  }
  void emit(const FbxArray& v, const FbxArray& p)
  {
    if (v.type != 'd' || p.type != 'i' || v.count < 9 || p.count < 3)
      return; // not a mesh geometry (e.g. a shape or a line)
    const size_t numPoints = v.count / 3;
    auto model = std::make_unique<Model>();
    std::vector<uint32_t> polygon;
    for (size_t k = 0; k < p.count; ++k)
    {
      int32_t raw;
      std::memcpy(&raw, p.raw.data() + 4 * k, 4);
      const bool last = raw < 0;
      const uint32_t point = (uint32_t)(last ? ~raw : raw);
      if (point >= numPoints)
      {
        polygon.clear();
        if (!last)
          continue;
      }
      else
      {
        double xyz[3];
        std::memcpy(xyz, v.raw.data() + 24 * (size_t)point, 24);
        polygon.push_back((uint32_t)model->positions.size());
        model->positions.push_back(vec3{(float)xyz[0], (float)xyz[1], (float)xyz[2]}); // one vertex per polygon corner
        model->boundingBox.adjust(model->positions.back());
      }
      if (last)
      {
        for (size_t c = 2; c < polygon.size(); ++c) // (0,1,2), (0,2,3), ...: aiProcess_Triangulate on convex polygons
          model->indices.push_back(uvec3{polygon[0], polygon[c - 1], polygon[c]});
        polygon.clear();
      }
    }
    model->texCoords.assign(model->positions.size(), vec2{0.0f, 0.0f});
    model->normals.assign(model->positions.size(), vec3{0.0f, 0.0f, 0.0f});
    if (!model->indices.empty())
      out->push_back(std::move(model));
  }
};

std::vector<PModel> loadFbx(const std::string& path)
{
  std::vector<PModel> result;
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f)
  {
    std::fprintf(stderr, "ModelLoader: cannot open %s\n", path.c_str());
    return result;
  }
  std::vector<unsigned char> file;
  unsigned char buf[65536];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0)
    file.insert(file.end(), buf, buf + got);
  std::fclose(f);
  if (file.size() < 27 || std::memcmp(file.data(), "Kaydara FBX Binary  ", 20) != 0)
  {
    std::fprintf(stderr, "ModelLoader: %s is not a binary FBX file (ASCII FBX is not read)\n", path.c_str());
    return result;
  }
  FbxReader reader{file, readU32(file.data() + 23) >= 7500u, true, &result};
  size_t at = 27;
  while (reader.ok && at + (reader.wide ? 25u : 13u) <= file.size())
  {
    const size_t next = reader.node(at, 0, nullptr, nullptr);
    if (!next)
      break;
    at = next;
  }
  if (!reader.ok)
  {
    std::fprintf(stderr, "ModelLoader: %s: malformed FBX node list\n", path.c_str());
    result.clear();
  }
  return result;
}

bool endsWith(const std::string& s, const char* suffix)
{
  const size_t n = std::strlen(suffix);
  if (s.size() < n)
    return false;
  for (size_t k = 0; k < n; ++k)
    if (std::tolower((unsigned char)s[s.size() - n + k]) != suffix[k])
      return false;
  return true;
}
} // namespace

std::vector<PModel> ModelLoader::loadModel(std::string_view filename)
{
  const std::string path(filename);
  if (endsWith(path, ".glb"))
    return loadGlb(path);
  if (endsWith(path, ".fbx"))
    return loadFbx(path);
  return loadObj(path);
}
} // namespace rt
