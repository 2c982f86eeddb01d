// TEST INFRASTRUCTURE -- the CPU oracle (see rt_oracle.h for what may use it).
//
// A CPU restatement of the reference's src/cpu path, operation for operation.
// Every function cites the reference lines it follows (paths relative to
// /root/reference).  PARITY UNPINNED by the reference's own tests (it has
// none); pinned by tests/test_oracle_kat.py (hand-derived answers).
//
// Build: g++ -std=c++20 -O2 -ffp-contract=off (no value-changing options).
#include "rt_oracle.h"
#include "column_pool.h"
#include "rt_math.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

using namespace orc;

namespace
{
// ------------------------------------------------------------------ PODs
struct Box // src/scene/AABB.h:8-11
{
  V3 lo = v3(FLT_MAX);
  V3 hi = v3(-FLT_MAX);
};
struct RayO // src/util/Ray.h:13-17
{
  V3 origin, direction;
};
struct ModelRef // src/scene/Scene.h:6-12
{
  uint32_t positionOffset = 0, numPositions = 0, indicesOffset = 0, numIndices = 0;
};
struct DirLight // Scene.h:22-28
{
  V3 direction, color;
};
struct PtLight // Scene.h:14-20
{
  V3 position, color;
  float attenuation;
};
struct Hit // Ray.h:19-28 + BRDF.h:9-18 (only the fields the path reads) + the triId AOV
{
  float t = FLT_MAX;
  V3 position{}, normal{}, normalLight{};
  V2 texCoords{};
  V3 albedo = v3(1.0f), emissive = v3(0.0f);
  uint32_t triId = 0xffffffffu;
};
struct PathState // Ray.h:4-11
{
  V3 rnd01{};
  V3 radiance = v3(0.0f);
  V3 material = v3(1.0f);
  uint32_t depth = 0;
  float emissive = 1.0f;
};

// AABB.h:12-16
float boxArea(const Box& b)
{
  V3 d = {b.hi.x - b.lo.x, b.hi.y - b.lo.y, b.hi.z - b.lo.z};
  return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
// AABB.h:61-69
Box boxUnion(const Box& a, const Box& b)
{
  Box r;
  r.lo = {fmin_ref(a.lo.x, b.lo.x), fmin_ref(a.lo.y, b.lo.y), fmin_ref(a.lo.z, b.lo.z)};
  r.hi = {fmax_ref(a.hi.x, b.hi.x), fmax_ref(a.hi.y, b.hi.y), fmax_ref(a.hi.z, b.hi.z)};
  return r;
}
// AABB.h:17-26
void boxGrow(Box& b, V3 p)
{
  b.lo = {fmin_ref(b.lo.x, p.x), fmin_ref(b.lo.y, p.y), fmin_ref(b.lo.z, p.z)};
  b.hi = {fmax_ref(b.hi.x, p.x), fmax_ref(b.hi.y, p.y), fmax_ref(b.hi.z, p.z)};
}
// AABB.h:27-46: box of the 8 transformed corners, corner order zyx-minor as listed there
Box boxTransform(const Box& b, const M4& t)
{
  Box r;
  for (int i = 0; i < 8; ++i)
  {
    V3 c = {(i & 4) ? b.hi.x : b.lo.x, (i & 2) ? b.hi.y : b.lo.y, (i & 1) ? b.hi.z : b.lo.z};
    boxGrow(r, mulPoint(t, c));
  }
  return r;
}
// AABB.h:47-60.  tmin/tmax by value; NaN-sensitive min/max forms kept.
bool boxHit(const Box& b, const RayO& ray, float tmin, float tmax)
{
  V3 invD = 1.0f / ray.direction;
  V3 t0s = (b.lo - ray.origin) * invD;
  V3 t1s = (b.hi - ray.origin) * invD;
  V3 tsmaller = vmin(t0s, t1s);
  V3 tbigger = vmax(t0s, t1s);
  tmin = fmax_ref(tmin, fmax_ref(tsmaller.x, fmax_ref(tsmaller.y, tsmaller.z)));
  tmax = fmin_ref(tmax, fmin_ref(tbigger.x, fmin_ref(tbigger.y, tbigger.z)));
  return tmin < tmax;
}

// src/util/Texture.h:8-16
struct TextureData
{
  std::vector<V3> textureData;
  int width = 0, height = 0;
  // Texture.cpp:3-8: `uint32_t x = texCoords.x * width` (int -> float, float multiply, truncation), row-major fetch.
  // The reference indexes unchecked; outside [0,1) this restatement clamps to the edge texels (NaN -> texel 0), the
  // convention include/rthip.h documents for rthip_upload_materials.
  static uint32_t texel(float coord, int size)
  {
    const float f = coord * (float)size;
    if (!(f >= 0.0f))
      return 0u;
    if (f >= (float)size)
      return (uint32_t)size - 1u;
    return (uint32_t)f;
  }
  V3 sample(V2 texCoords) const
  {
    const uint32_t x = texel(texCoords.x, width);
    const uint32_t y = texel(texCoords.y, height);
    return textureData[(size_t)y * (size_t)width + x];
  }
};
// src/util/Material.h:7-14.  kind 0 is Material::evaluate exactly as the reference has it (Material.cpp:4-12); kind 1 is
// the wiring the class is evidently meant for: factor [* texture sample] for albedo and emissive.
struct MaterialData
{
  uint32_t kind = 1;
  V3 baseColor = {0.0f, 1.0f, 0.0f}, emissiveFactor = {0.0f, 0.0f, 0.0f}; // CPUScene.cpp:245-246
  std::shared_ptr<TextureData> albedoTexture, emissiveTexture;
  void evaluate(V2 texCoords, V3& albedo, V3& emissive) const
  {
    if (kind == 0)
    {
      albedo = V3{texCoords.x, texCoords.y, 0.0f}; // glm::vec3(hitInfo.texCoords, 0)
      emissive = V3{0.0f, 0.0f, 0.0f};
      return;
    }
    albedo = baseColor;
    emissive = emissiveFactor;
    if (albedoTexture)
      albedo = albedo * albedoTexture->sample(texCoords);
    if (emissiveTexture)
      emissive = emissive * emissiveTexture->sample(texCoords);
  }
};

struct ModelData // src/util/Model.h:13-24
{
  Box bbox;
  std::vector<V3> positions, normals, edges, faceNormals;
  std::vector<V2> texCoords;
  std::vector<U3> indices;
  MaterialData material;
};

struct TreeNode // src/cpu/CPUScene.h:11-19
{
  std::unique_ptr<TreeNode> left, right;
  Box box;
  ModelRef model; // numIndices == 0 on inner nodes
  uint32_t modelIndex = 0xffffffffu;
};

struct Counters
{
  uint64_t extendPrimary = 0, extendSecondary = 0, shadow = 0, nodesVisited = 0, trisTested = 0, nonFinite = 0;
  uint64_t hits[16] = {};
  void addTo(orc_counters& o) const
  {
    o.extendPrimary += extendPrimary;
    o.extendSecondary += extendSecondary;
    o.shadow += shadow;
    o.nodesVisited += nodesVisited;
    o.trisTested += trisTested;
    o.nonFiniteAccepts += nonFinite;
    for (int i = 0; i < 16; ++i)
      o.hits[i] += hits[i];
  }
};
} // namespace

struct orc_scene
{
  bool fixEdgeOffset = false;
  // Scene.h:45-56
  std::vector<ModelRef> refs;
  std::vector<V3> positionPool, normalsPool, edgesPool, faceNormalsPool;
  std::vector<V2> texCoordsPool;
  std::vector<U3> indicesPool;
  std::vector<PtLight> pointLights;
  std::vector<DirLight> directionalLights;
  std::vector<std::unique_ptr<ModelData>> models;
  std::unique_ptr<TreeNode> hierarchy; // CPUScene.h:20
  // variant B
  const rtflat_model_node* fModelNodes = nullptr;
  const rtflat_node* fNodes = nullptr;
  const rtflat_tri* fTris = nullptr;
  rtflat_info fInfo{};
  // shading beyond the reference's CPU path (both off = that path as it stands)
  bool materialsOn = false;  // a hit's albedo / emissive come from its model's material instead of CPUScene.cpp:243-247
  bool pointShadows = false; // point lights cast shadow rays as in src/metal/Compute.metal:238-254
  const rtflat_shade* fShade = nullptr; // variant B reads the flattened shading tables (include/rtflat.h)
  const rtflat_material* fMaterials = nullptr;
  const float* fTexels = nullptr;

  // ---------------------------------------------------------- scene build
  // Model.cpp:3-26
  static void transformModel(ModelData& m, const M4& t)
  {
    for (auto& p : m.positions)
      p = mulPoint(t, p);
    for (auto& n : m.normals)
      n = mulDir(t, n);
    m.bbox = boxTransform(m.bbox, t);
    for (size_t i = 0; i < m.indices.size(); ++i)
    {
      V3 e0 = m.positions[m.indices[i].y] - m.positions[m.indices[i].x];
      V3 e1 = m.positions[m.indices[i].z] - m.positions[m.indices[i].x];
      m.edges.push_back(e0);
      m.edges.push_back(e1);
      m.faceNormals.push_back(cross(e0, e1));
    }
  }
  // Scene.cpp:18-46
  void generate()
  {
    for (uint32_t k = 0; k < models.size(); ++k)
    {
      ModelData& m = *models[k];
      ModelRef ref;
      ref.positionOffset = (uint32_t)positionPool.size();
      ref.numPositions = (uint32_t)m.positions.size();
      ref.indicesOffset = (uint32_t)indicesPool.size();
      ref.numIndices = (uint32_t)m.indices.size();
      for (uint32_t i = 0; i < m.positions.size(); ++i)
      {
        positionPool.push_back(m.positions[i]);
        texCoordsPool.push_back(m.texCoords[i]);
        normalsPool.push_back(m.normals[i]);
      }
      for (uint32_t i = 0; i < m.indices.size(); ++i)
      {
        indicesPool.push_back(m.indices[i]);
        edgesPool.push_back(m.edges[i * 2 + 0]);
        edgesPool.push_back(m.edges[i * 2 + 1]);
        faceNormalsPool.push_back(normalize(m.faceNormals[i]));
      }
      refs.push_back(ref);
    }
    buildHierarchy();
  }
  // CPUScene.cpp:67-105: agglomerative pairing, first strict minimum of the combined surface area
  void buildHierarchy()
  {
    std::vector<std::unique_ptr<TreeNode>> open;
    for (uint32_t k = 0; k < models.size(); ++k)
    {
      auto n = std::make_unique<TreeNode>();
      n->box = models[k]->bbox;
      n->model = refs[k];
      n->modelIndex = k;
      open.push_back(std::move(n));
    }
    while (open.size() > 1)
    {
      size_t lhs = open.size(), rhs = open.size();
      float best = FLT_MAX;
      for (size_t i = 0; i < open.size(); ++i)
        for (size_t j = 0; j < open.size(); ++j)
        {
          if (i == j)
            continue;
          float area = boxArea(boxUnion(open[i]->box, open[j]->box));
          if (best > area)
          {
            lhs = i;
            rhs = j;
            best = area;
          }
        }
      auto parent = std::make_unique<TreeNode>();
      parent->box = boxUnion(open[lhs]->box, open[rhs]->box);
      parent->left = std::move(open[lhs]);
      parent->right = std::move(open[rhs]);
      // (i,j) is found before (j,i), so rhs > lhs (the reference asserts it, CPUScene.cpp:98)
      open.erase(open.begin() + rhs);
      open.erase(open.begin() + lhs);
      open.push_back(std::move(parent));
    }
    if (!open.empty())
      hierarchy = std::move(open[0]);
  }

  // where the reference reads this model's edge pair for triangle i (CPUScene.cpp:157-158, 205-206)
  size_t edgeBase(const ModelRef& r) const { return fixEdgeOffset ? (size_t)2 * r.indicesOffset : (size_t)r.indicesOffset; }

  // ------------------------------------------------- variant A: the reference's walk
  // one triangle test, shared by intersectModel / testModel (CPUScene.cpp:145-181 == :193-229)
  // returns true when the reference would NOT `continue` before its last rejection
  inline bool triangleAccept(const ModelRef& r, size_t posIndex, const RayO& ray, float tmin, float tmax, V3& rv, V2& uv,
                             Counters& cnt) const
  {
    const U3 idx = indicesPool[r.indicesOffset + posIndex];
    const V3& p0 = positionPool[r.positionOffset + idx.x];
    const V2& t0 = texCoordsPool[r.positionOffset + idx.x];
    const V2& t1 = texCoordsPool[r.positionOffset + idx.y];
    const V2& t2 = texCoordsPool[r.positionOffset + idx.z];
    const size_t eb = edgeBase(r) + 2 * posIndex;
    const V3& e0 = edgesPool[eb];
    const V3& e1 = edgesPool[eb + 1];
    cnt.trisTested++;
    const V3 s = ray.origin - p0;
    const V3 s1 = cross(ray.direction, e1);
    const V3 s2 = cross(s, e0);
    const float fraction = 1.0f / dot(s1, e0);
    rv = V3{dot(s2, e1), dot(s1, s), dot(s2, ray.direction)} * fraction;
    const float b3 = 1.0f - rv.y - rv.z;
    uv = t0 * rv.y + t1 * rv.z + t2 * b3;
    if (b3 < 0 || b3 > 1)
      return false;
    if (rv.y < 0 || rv.y > 1)
      return false;
    if (rv.z < 0 || rv.z > 1)
      return false;
    if (rv.x < tmin || rv.x > tmax)
      return false;
    if (!(std::isfinite(rv.x) && std::isfinite(rv.y) && std::isfinite(rv.z)))
      cnt.nonFinite++;
    return true;
  }
  // CPUScene.cpp:187-252
  Hit scanModelClosest(const ModelRef& r, uint32_t modelIndex, const RayO& ray, float tmin, float tmax, Counters& cnt) const
  {
    Hit best;
    for (size_t i = 0; i < r.numIndices; ++i)
    {
      V3 rv;
      V2 uv;
      if (!triangleAccept(r, i, ray, tmin, tmax, rv, uv, cnt))
        continue;
      if (best.t < rv.x) // :231 -- equal t overwrites: the later triangle wins
        continue;
      const V3& n = faceNormalsPool[r.indicesOffset + i];
      best.t = rv.x;
      best.position = ray.origin + ray.direction * rv.x;
      best.normal = n;
      best.normalLight = dot(n, ray.direction) < 0 ? n : -n;
      best.texCoords = uv;
      best.albedo = V3{0.0f, 1.0f, 0.0f}; // :245
      best.emissive = V3{0.0f, 0.0f, 0.0f};
      if (materialsOn)
        models[modelIndex]->material.evaluate(uv, best.albedo, best.emissive);
      best.triId = r.indicesOffset + (uint32_t)i;
    }
    return best;
  }
  // CPUScene.cpp:139-186
  bool scanModelAny(const ModelRef& r, const RayO& ray, float tmin, float tmax, Counters& cnt) const
  {
    for (size_t i = 0; i < r.numIndices; ++i)
    {
      V3 rv;
      V2 uv;
      if (triangleAccept(r, i, ray, tmin, tmax, rv, uv, cnt))
        return true;
    }
    return false;
  }
  // CPUScene.cpp:123-137
  Hit closestA(const TreeNode* node, const RayO& ray, float tmin, float tmax, Counters& cnt) const
  {
    cnt.nodesVisited++;
    if (!boxHit(node->box, ray, tmin, tmax))
      return Hit{};
    if (node->model.numIndices > 0)
      return scanModelClosest(node->model, node->modelIndex, ray, tmin, tmax, cnt);
    Hit l = closestA(node->left.get(), ray, tmin, tmax, cnt);
    Hit r = closestA(node->right.get(), ray, tmin, tmax, cnt);
    return l.t < r.t ? l : r; // :136 -- right wins unless left is strictly nearer
  }
  // CPUScene.cpp:107-121 (both subtrees are evaluated, then or-ed)
  bool anyA(const TreeNode* node, const RayO& ray, float tmin, float tmax, Counters& cnt) const
  {
    cnt.nodesVisited++;
    if (!boxHit(node->box, ray, tmin, tmax))
      return false;
    if (node->model.numIndices > 0)
      return scanModelAny(node->model, ray, tmin, tmax, cnt);
    bool l = anyA(node->left.get(), ray, tmin, tmax, cnt);
    bool r = anyA(node->right.get(), ray, tmin, tmax, cnt);
    return l || r;
  }

  // ------------------------------------------------- variant B: walk of the flattened arrays
  // The same traversal rule the HIP extend/shadow kernels implement (DESIGN.md "Traversal rule"):
  //  * model nodes: the reference's exact slab test, both children always visited, passing leaves
  //    are traversed in DFS (left to right) order;
  //  * BVH nodes (4-wide, boxes quantised outward to 8 bits): conservative slab test with explicit fmaf, entry
  //    clamped to 0; closest-hit additionally skips a child whose entry exceeds best*(1+cullRel)+cullAbs;
  //    children visited nearest first;
  //  * triangles: the reference's Moeller-Trumbore on the record's operands; winner = smallest t,
  //    ties -> larger orderKey (== "later triangle / right subtree wins").
  struct FlatRay
  {
    V3 o, d, invD, oi;
  };
  static inline bool flatTri(const rtflat_tri& tr, const RayO& ray, float tmin, float tmax, V3& rv, Counters& cnt)
  {
    cnt.trisTested++;
    const V3 p0 = {tr.p0[0], tr.p0[1], tr.p0[2]}, e0 = {tr.e0[0], tr.e0[1], tr.e0[2]}, e1 = {tr.e1[0], tr.e1[1], tr.e1[2]};
    const V3 s = ray.origin - p0;
    const V3 s1 = cross(ray.direction, e1);
    const V3 s2 = cross(s, e0);
    const float fraction = 1.0f / dot(s1, e0);
    rv = V3{dot(s2, e1), dot(s1, s), dot(s2, ray.direction)} * fraction;
    const float b3 = 1.0f - rv.y - rv.z;
    if (b3 < 0 || b3 > 1)
      return false;
    if (rv.y < 0 || rv.y > 1)
      return false;
    if (rv.z < 0 || rv.z > 1)
      return false;
    if (rv.x < tmin || rv.x > tmax)
      return false;
    if (!(std::isfinite(rv.x) && std::isfinite(rv.y) && std::isfinite(rv.z)))
    {
      cnt.nonFinite++;
      return false; // outside the parity domain; counted, never wins
    }
    return true;
  }
  // 1/d for the padded-box tests, clamped to +-1e18 so that an exactly axis-parallel ray (d == 0 -> 1/d == inf)
  // keeps finite slab arithmetic (inf - inf would turn into a NaN that fmin/fmax silently drop)
  static inline V3 flatInvDir(V3 d)
  {
    const V3 r = 1.0f / d;
    return V3{fminf(fmaxf(r.x, -1e18f), 1e18f), fminf(fmaxf(r.y, -1e18f), 1e18f), fminf(fmaxf(r.z, -1e18f), 1e18f)};
  }
  // One 4-wide quantised node (include/rtflat.h): tests the four child boxes with the arithmetic of the kernels
  // (slab distance = q * (step*invD) + (origin*invD - o*invD), explicit fmaf, entry clamped to 0) and returns the children
  // that were hit sorted by entry distance with the kernels' 5-exchange network; key 0xffffffff marks a miss.
  static inline void wideStep(const rtflat_node& nd, const V3& invD, const V3& oi, bool useCull, float cull, uint32_t key[4], uint32_t ref[4])
  {
    const float s[3] = {stepOf(nd.exp[0]), stepOf(nd.exp[1]), stepOf(nd.exp[2])};
    const float inv[3] = {invD.x, invD.y, invD.z}, o3[3] = {oi.x, oi.y, oi.z};
    float a[3], b[3];
    for (int ax = 0; ax < 3; ++ax)
    {
      a[ax] = s[ax] * inv[ax];
      b[ax] = __builtin_fmaf(nd.origin[ax], inv[ax], -o3[ax]);
    }
    for (int k = 0; k < 4; ++k)
    {
      // the ray enters a slab through the low plane when it travels in +axis, through the high one otherwise; an
      // unused slot (qlo 255, qhi 0) therefore has entry > exit for every ray
      float nr[3], fr[3];
      for (int ax = 0; ax < 3; ++ax)
      {
        const bool neg = inv[ax] < 0.0f;
        nr[ax] = __builtin_fmaf((float)(neg ? nd.qhi[ax][k] : nd.qlo[ax][k]), a[ax], b[ax]);
        fr[ax] = __builtin_fmaf((float)(neg ? nd.qlo[ax][k] : nd.qhi[ax][k]), a[ax], b[ax]);
      }
      const float entry = fmaxf(fmaxf(nr[0], nr[1]), fmaxf(nr[2], 0.0f));
      const float exit = fminf(fminf(fr[0], fr[1]), fr[2]);
      const bool hit = entry <= exit && (!useCull || entry <= cull);
      uint32_t bits;
      std::memcpy(&bits, &entry, 4);
      key[k] = hit ? bits : 0xffffffffu;
      ref[k] = nd.child[k];
    }
    auto order = [&](int i, int j) {
      if (key[j] < key[i])
      {
        std::swap(key[i], key[j]);
        std::swap(ref[i], ref[j]);
      }
    };
    order(0, 1), order(2, 3), order(0, 2), order(1, 3), order(1, 2);
  }
  static inline float stepOf(uint8_t biasedExponent)
  {
    const uint32_t bits = (uint32_t)biasedExponent << 23;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
  }
  // shared by closest / any hit: descends from `cur`; returns the next thing to visit (leaf or node) or RTFLAT_INVALID
  // when the stack has run empty.  Far children are pushed farthest first.
  inline uint32_t wideDescend(uint32_t cur, const V3& invD, const V3& oi, bool useCull, float cull, uint32_t* stack, int& sp, Counters& cnt) const
  {
    const rtflat_node& nd = fNodes[cur];
    cnt.nodesVisited++;
    uint32_t key[4], ref[4];
    wideStep(nd, invD, oi, useCull, cull, key, ref);
    for (int k = 3; k >= 1; --k)
      if (key[k] != 0xffffffffu)
        stack[sp++] = ref[k];
    if (key[0] != 0xffffffffu)
      return ref[0];
    return sp == 0 ? RTFLAT_INVALID : stack[--sp];
  }
  // collects the BLAS roots of the model leaves the reference would scan, in DFS order
  void flatModelLeaves(uint32_t nodeIdx, const RayO& ray, float tmin, float tmax, uint32_t* out, uint32_t& n, Counters& cnt) const
  {
    const rtflat_model_node& mn = fModelNodes[nodeIdx];
    cnt.nodesVisited++;
    Box b;
    b.lo = {mn.lo[0], mn.lo[1], mn.lo[2]};
    b.hi = {mn.hi[0], mn.hi[1], mn.hi[2]};
    if (!boxHit(b, ray, tmin, tmax))
      return;
    if (mn.triCount > 0)
    {
      out[n++] = mn.blas;
      return;
    }
    flatModelLeaves(mn.left, ray, tmin, tmax, out, n, cnt);
    flatModelLeaves(mn.right, ray, tmin, tmax, out, n, cnt);
  }
  // returns record index of the winner or RTFLAT_INVALID
  uint32_t closestB(const RayO& ray, float tmin, float tmax, float& tBest, Counters& cnt) const
  {
    uint32_t roots[64], nRoots = 0;
    flatModelLeaves(fInfo.rootModelNode, ray, tmin, tmax, roots, nRoots, cnt);
    const V3 invD = flatInvDir(ray.direction);
    const V3 oi = ray.origin * invD;
    float best = FLT_MAX, cull = FLT_MAX;
    uint32_t bestKey = 0, bestRec = RTFLAT_INVALID;
    uint32_t stack[512];
    for (uint32_t r = 0; r < nRoots; ++r)
    {
      int sp = 0;
      uint32_t cur = roots[r];
      for (;;)
      {
        if (rtflat_is_leaf(cur))
        {
          const uint32_t first = rtflat_leaf_first(cur), count = rtflat_leaf_count(cur);
          for (uint32_t k = 0; k < count; ++k)
          {
            const rtflat_tri& tr = fTris[first + k];
            V3 rv;
            if (!flatTri(tr, ray, tmin, tmax, rv, cnt))
              continue;
            if (rv.x < best || (rv.x == best && tr.orderKey > bestKey))
            {
              best = rv.x;
              bestKey = tr.orderKey;
              bestRec = first + k;
              cull = __builtin_fmaf(best, 1.0f + fInfo.cullRel, fInfo.cullAbs);
            }
          }
          if (sp == 0)
            break;
          cur = stack[--sp];
          continue;
        }
        cur = wideDescend(cur, invD, oi, true, cull, stack, sp, cnt);
        if (cur == RTFLAT_INVALID)
          break;
      }
    }
    tBest = best;
    return bestRec;
  }
  bool anyB(const RayO& ray, float tmin, float tmax, Counters& cnt) const
  {
    uint32_t roots[64], nRoots = 0;
    flatModelLeaves(fInfo.rootModelNode, ray, tmin, tmax, roots, nRoots, cnt);
    const V3 invD = flatInvDir(ray.direction);
    const V3 oi = ray.origin * invD;
    uint32_t stack[512];
    for (uint32_t r = 0; r < nRoots; ++r)
    {
      int sp = 0;
      uint32_t cur = roots[r];
      for (;;)
      {
        if (rtflat_is_leaf(cur))
        {
          const uint32_t first = rtflat_leaf_first(cur), count = rtflat_leaf_count(cur);
          for (uint32_t k = 0; k < count; ++k)
          {
            V3 rv;
            if (flatTri(fTris[first + k], ray, tmin, tmax, rv, cnt))
              return true;
          }
          if (sp == 0)
            break;
          cur = stack[--sp];
          continue;
        }
        cur = wideDescend(cur, invD, oi, false, 0.0f, stack, sp, cnt);
        if (cur == RTFLAT_INVALID)
          break;
      }
    }
    return false;
  }

  // dispatch
  Hit closest(bool flat, const RayO& ray, float tmin, float tmax, Counters& cnt) const
  {
    if (!flat)
      return closestA(hierarchy.get(), ray, tmin, tmax, cnt);
    float t;
    const uint32_t rec = closestB(ray, tmin, tmax, t, cnt);
    Hit h;
    if (rec == RTFLAT_INVALID)
      return h;
    const rtflat_tri& tr = fTris[rec];
    const V3 n = {tr.n[0], tr.n[1], tr.n[2]};
    h.t = t;
    h.position = ray.origin + ray.direction * t;
    h.normal = n;
    h.normalLight = dot(n, ray.direction) < 0 ? n : -n;
    h.albedo = V3{0.0f, 1.0f, 0.0f};
    h.emissive = V3{0.0f, 0.0f, 0.0f};
    h.triId = tr.triId;
    if (materialsOn)
    {
      // texture coordinates as CPUScene.cpp:217-219 forms them, from the winning record's own barycentrics
      Counters scratch;
      V3 rv;
      flatTri(tr, ray, tmin, tmax, rv, scratch);
      const rtflat_shade& sh = fShade[tr.triId];
      const V2 t0 = {sh.uv[0][0], sh.uv[0][1]}, t1 = {sh.uv[1][0], sh.uv[1][1]}, t2 = {sh.uv[2][0], sh.uv[2][1]};
      const float b3 = 1.0f - rv.y - rv.z;
      h.texCoords = t0 * rv.y + t1 * rv.z + t2 * b3;
      flatMaterial(fMaterials[sh.material], h.texCoords, h.albedo, h.emissive);
    }
    return h;
  }
  // a flattened material record evaluated like MaterialData::evaluate
  void flatMaterial(const rtflat_material& m, V2 texCoords, V3& albedo, V3& emissive) const
  {
    auto sample = [&](uint32_t first, uint32_t w, uint32_t h) {
      const uint32_t x = TextureData::texel(texCoords.x, (int)w), y = TextureData::texel(texCoords.y, (int)h);
      const float* t = fTexels + 3 * ((size_t)first + (size_t)y * w + x);
      return V3{t[0], t[1], t[2]};
    };
    if (m.kind == RTFLAT_MAT_STUB)
    {
      albedo = V3{texCoords.x, texCoords.y, 0.0f};
      emissive = V3{0.0f, 0.0f, 0.0f};
      return;
    }
    albedo = V3{m.baseColor[0], m.baseColor[1], m.baseColor[2]};
    emissive = V3{m.emissive[0], m.emissive[1], m.emissive[2]};
    if (m.albedoW)
      albedo = albedo * sample(m.albedoTexel, m.albedoW, m.albedoH);
    if (m.emissiveW)
      emissive = emissive * sample(m.emissiveTexel, m.emissiveW, m.emissiveH);
  }
  bool any(bool flat, const RayO& ray, float tmin, float tmax, Counters& cnt) const
  {
    return flat ? anyB(ray, tmin, tmax, cnt) : anyA(hierarchy.get(), ray, tmin, tmax, cnt);
  }

  // ------------------------------------------------- shading
  // BRDF.cpp:4-14 (h / specular are computed and discarded there; nothing of them reaches the result)
  static V3 brdfEvaluate(const Hit& hit, V3 /*viewDir*/, V3 lightDir, V3 lightColor)
  {
    const float diffuse = fmax_ref(dot(hit.normal, lightDir), 0.0f);
    return (hit.albedo * diffuse * lightColor) + V3{0.03f, 0.03f, 0.03f};
  }
  // CPUScene.cpp:54-60
  static V3 bounceDirection(V3 normalLight, V3 rnd01)
  {
    const float r1 = (float)(2 * 3.141592653589793238462643383279502884 * (double)rnd01.x); // int*double*float -> double, narrowed
    const float r2 = rnd01.y;
    const float r2s = sqrtf(r2);
    const V3 w = normalLight;
    const V3 u = normalize(cross((double)fabsf(w.x) > 0.1 ? V3{0, 1, 0} : V3{1, 0, 0}, w));
    const V3 v = cross(w, u);
    float sn, cs;
    sincos_det(r1, &sn, &cs);
    return normalize(u * cs * r2s + v * sn * r2s + w * sqrtf(1 - r2));
  }
  // CPUScene.cpp:6-65.  The reference recurses in tail position; this is the same sequence as a loop.
  void tracePath(RayO ray, PathState& path, float tmin, float tmax, uint32_t maxDepth, bool flat, Counters& cnt, uint32_t* aovTri,
                 uint8_t* aovShadow, size_t aovStride, uint32_t aovDepths) const
  {
    for (;;)
    {
      if (path.depth == 0)
        cnt.extendPrimary++;
      else
        cnt.extendSecondary++;
      const Hit info = closest(flat, ray, tmin, tmax, cnt);
      if (!(info.t < FLT_MAX)) // :10
        return;
      cnt.hits[path.depth < 15 ? path.depth : 15]++;
      const float p = fmax_ref(fmax_ref(info.albedo.x, info.albedo.y), info.albedo.z); // :13
      if (path.depth >= maxDepth) // :15 (12 in the reference)
        return;
      else if (path.depth > 5) // :19
      {
        if (path.rnd01.z >= p)
          return;
        else
          path.material = path.material / p;
      }
      if (path.depth < aovDepths)
        aovTri[path.depth * aovStride] = info.triId;
      path.radiance = path.radiance + path.material * info.emissive * path.emissive; // :27
      path.material = path.material * info.albedo;                                  // :28
      uint8_t mask = 0;
      for (size_t l = 0; l < directionalLights.size(); ++l) // :31-38
      {
        const DirLight& d = directionalLights[l];
        cnt.shadow++;
        if (!any(flat, RayO{info.position, -d.direction}, 1e-4f, 1e20f, cnt))
          path.radiance = path.radiance + brdfEvaluate(info, -ray.direction, -d.direction, d.color);
        else if (l < 8)
          mask |= (uint8_t)(1u << l);
      }
      if (path.depth < aovDepths)
        aovShadow[path.depth * aovStride] = mask;
      for (size_t l = 0; l < pointLights.size(); ++l) // :39-49 (no shadow ray: commented out at :42)
      {
        const PtLight& pl = pointLights[l];
        const V3 lightDir = pl.position - info.position;
        if (pointShadows)
        {
          // Compute.metal:240-246: origin = position + normal * 1e-3, direction = lightDir (not normalised),
          // max_distance = 1 (the light itself), min_distance left at 0; lit only when nothing is hit.
          // (There `normal` is the interpolated vertex normal; this path has the face normal, CPUScene.cpp:239.)
          cnt.shadow++;
          if (any(flat, RayO{info.position + info.normal * 1e-3f, lightDir}, 0.0f, 1.0f, cnt))
          {
            const size_t bit = directionalLights.size() + l;
            if (bit < 8 && path.depth < aovDepths)
              aovShadow[path.depth * aovStride] |= (uint8_t)(1u << bit);
            continue;
          }
        }
        const float dist = length(lightDir);
        const float illuminance = fmax_ref(1 - dist / pl.attenuation, 0.0f);
        path.radiance = path.radiance + illuminance * brdfEvaluate(info, -ray.direction, lightDir, pl.color);
      }
      ray = RayO{info.position, bounceDirection(info.normalLight, path.rnd01)}; // :54-60
      path.emissive = 0; // :61
      path.depth++;      // :62
    }
  }
};

// ------------------------------------------------------------------ ray generation
namespace
{
// CPURenderer.cpp:111-116
V3 rand01(U3 x)
{
  for (int i = 3; i-- > 0;)
  {
    U3 y = {x.y, x.z, x.x};
    x = {((x.x >> 8u) ^ y.x) * 1103515245u, ((x.y >> 8u) ^ y.y) * 1103515245u, ((x.z >> 8u) ^ y.z) * 1103515245u};
  }
  const float k = 1.0f / float(0xffffffffu);
  return V3{float(x.x), float(x.y), float(x.z)} * k;
}

struct CamO
{
  V3 position, target;
  V2 sensorSize;
  float S_O, f, A;
};
CamO toCam(const orc_camera* c)
{
  return CamO{{c->position[0], c->position[1], c->position[2]}, {c->target[0], c->target[1], c->target[2]},
              {c->sensorSize[0], c->sensorSize[1]}, c->S_O, c->f, c->A};
}

// CPURenderer.cpp:158-192, one (pixel, sample)
RayO primaryRay(const CamO& camera, uint32_t width, uint32_t height, uint32_t w, uint32_t h, uint32_t samp, V3& rnd)
{
  const RayO cam = {camera.position, normalize(camera.target - camera.position)};
  const V3 cx = normalize(cross(cam.direction, (double)fabsf(cam.direction.y) < 0.9 ? V3{0, 1, 0} : V3{0, 0, 1}));
  const V3 cy = cross(cx, cam.direction);
  const V2 sdim = camera.sensorSize;
  const float S_I = (camera.S_O * camera.f) / (camera.S_O - camera.f);

  rnd = rand01(U3{w, h, samp});
  const V2 rnd2 = 2.0f * V2{rnd.x, rnd.y};
  const V2 tent = {rnd2.x < 1 ? sqrtf(rnd2.x) - 1 : 1 - sqrtf(2 - rnd2.x), rnd2.y < 1 ? sqrtf(rnd2.y) - 1 : 1 - sqrtf(2 - rnd2.y)};
  const int isamp = (int)samp;
  const V2 stratum = {(float)((isamp / 2) % 2), (float)(isamp % 2)};
  const V2 pix = {(float)w, (float)h};
  const V2 s = ((pix + 0.5f * (0.5f + stratum + tent)) / V2{(float)width, (float)height} - 0.5f) * sdim;
  const V3 spos = cam.origin + cx * s.x + cy * s.y;
  const V3 lc = cam.origin + cam.direction * 0.035f;
  RayO r = {lc, normalize(lc - spos)};

  const V3 lensP = lc;
  const V3 lensN = -cam.direction;
  const V3 lensX = cross(lensN, V3{0, 1, 0});
  const V3 lensY = cross(lensN, lensX);
  const V3 lensSample = lensP + rnd.x * camera.A * lensX + rnd.y * camera.A * lensY;
  const V3 focalPoint = cam.origin + (camera.S_O + S_I) * cam.direction;
  const float t = dot(focalPoint - r.origin, lensN) / dot(r.direction, lensN);
  const V3 focus = r.origin + t * r.direction;
  r = RayO{lensSample, normalize(focus - lensSample)};
  return r;
}

struct RenderJobCtx
{
  const orc_scene* scene;
  CamO camera;
  orc_render_params prm;
  float* accumulator;
  float* image;
  uint32_t* aovTri;
  uint8_t* aovShadow;
  std::mutex* counterLock;
  orc_counters* counters;
};

// the body of one column task, CPURenderer.cpp:153-201
ColumnJob columnTask(const RenderJobCtx* ctx, uint32_t w, uint32_t samp)
{
  const orc_render_params& prm = ctx->prm;
  Counters cnt;
  const size_t plane = (size_t)prm.width * prm.height;
  for (uint32_t h = prm.y0; h < prm.y1; ++h)
  {
    PathState path;
    RayO r = primaryRay(ctx->camera, prm.width, prm.height, w, h, samp, path.rnd01);
    const size_t pix = (size_t)w + (size_t)h * prm.width;
    uint32_t* aT = nullptr;
    uint8_t* aS = nullptr;
    if (prm.aovDepths)
    {
      aT = ctx->aovTri + ((size_t)samp * prm.aovDepths) * plane + pix;
      aS = ctx->aovShadow + ((size_t)samp * prm.aovDepths) * plane + pix;
    }
    ctx->scene->tracePath(r, path, 1e-4f, 1e20f, prm.maxDepth, prm.useFlat != 0, cnt, aT, aS, plane, prm.aovDepths);

    const float resolver = float(prm.numSamples) / float(samp + 1);
    float* acc = ctx->accumulator + 3 * pix;
    const V3 add = path.radiance / float(prm.numSamples);
    acc[0] += add.x;
    acc[1] += add.y;
    acc[2] += add.z;
    float* img = ctx->image + 3 * pix;
    for (int c = 0; c < 3; ++c)
      img[c] = powf(fmax_ref(acc[c] * resolver, 0.0f), 0.45f);
  }
  if (ctx->counters)
  {
    std::unique_lock g(*ctx->counterLock);
    cnt.addTo(*ctx->counters);
  }
  co_return;
}
} // namespace

// ------------------------------------------------------------------ C API
extern "C" {

orc_scene* orc_scene_create(int fixEdgeOffset)
{
  auto* s = new orc_scene();
  s->fixEdgeOffset = fixEdgeOffset != 0;
  return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

void orc_add_directional_light(orc_scene* s, const float dir[3], const float color[3])
{
  s->directionalLights.push_back(DirLight{{dir[0], dir[1], dir[2]}, {color[0], color[1], color[2]}});
}
void orc_add_point_light(orc_scene* s, const float pos[3], const float color[3], float attenuation)
{
  s->pointLights.push_back(PtLight{{pos[0], pos[1], pos[2]}, {color[0], color[1], color[2]}, attenuation});
}

void orc_add_model(orc_scene* s, const float* positions, uint32_t nPos, const float* texcoords, const uint32_t* indices,
                   uint32_t nTris, const float transform[16])
{
  auto m = std::make_unique<ModelData>();
  m->positions.reserve(nPos);
  for (uint32_t v = 0; v < nPos; ++v) // ModelLoader.cpp:19-33
  {
    V3 p = {positions[3 * v], positions[3 * v + 1], positions[3 * v + 2]};
    m->positions.push_back(p);
    m->texCoords.push_back(texcoords ? V2{texcoords[2 * v], texcoords[2 * v + 1]} : V2{0, 0});
    m->normals.push_back(V3{0, 0, 0}); // vertex normals never reach the cpu path's output
    boxGrow(m->bbox, p);
  }
  for (uint32_t i = 0; i < nTris; ++i) // ModelLoader.cpp:34-38
    m->indices.push_back(U3{indices[3 * i], indices[3 * i + 1], indices[3 * i + 2]});
  M4 t;
  std::memcpy(&t, transform, sizeof(t));
  orc_scene::transformModel(*m, t); // Scene.cpp:3-7
  s->models.push_back(std::move(m));
}
void orc_set_model_material(orc_scene* s, uint32_t modelIndex, uint32_t kind, const float baseColor[3], const float emissive[3],
                            const float* albedoTexels, uint32_t albedoW, uint32_t albedoH, const float* emissiveTexels, uint32_t emissiveW,
                            uint32_t emissiveH)
{
  if (modelIndex >= s->models.size())
    return;
  auto texture = [](const float* rgb, uint32_t w, uint32_t h) -> std::shared_ptr<TextureData> {
    if (!rgb || !w || !h)
      return nullptr;
    auto t = std::make_shared<TextureData>();
    t->width = (int)w, t->height = (int)h;
    for (size_t k = 0; k < (size_t)w * h; ++k)
      t->textureData.push_back(V3{rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]});
    return t;
  };
  MaterialData& m = s->models[modelIndex]->material;
  m.kind = kind;
  m.baseColor = V3{baseColor[0], baseColor[1], baseColor[2]};
  m.emissiveFactor = V3{emissive[0], emissive[1], emissive[2]};
  m.albedoTexture = texture(albedoTexels, albedoW, albedoH);
  m.emissiveTexture = texture(emissiveTexels, emissiveW, emissiveH);
}
void orc_set_shading(orc_scene* s, int materialsOn, int pointShadows)
{
  s->materialsOn = materialsOn != 0;
  s->pointShadows = pointShadows != 0;
}
void orc_set_flat_shading(orc_scene* s, const rtflat_shade* shade, const rtflat_material* materials, const float* texels)
{
  s->fShade = shade;
  s->fMaterials = materials;
  s->fTexels = texels;
}
void orc_generate(orc_scene* s) { s->generate(); }

uint32_t orc_num_tris(const orc_scene* s) { return (uint32_t)s->indicesPool.size(); }
uint32_t orc_num_models(const orc_scene* s) { return (uint32_t)s->models.size(); }
void orc_get_tri_operands(const orc_scene* s, uint32_t triId, float p0[3], float e0[3], float e1[3], float n[3])
{
  // find the model that owns triId
  for (const ModelRef& r : s->refs)
  {
    if (triId < r.indicesOffset || triId >= r.indicesOffset + r.numIndices)
      continue;
    const uint32_t i = triId - r.indicesOffset;
    const V3 a = s->positionPool[r.positionOffset + s->indicesPool[triId].x];
    const V3 b = s->edgesPool[s->edgeBase(r) + 2 * i], c = s->edgesPool[s->edgeBase(r) + 2 * i + 1];
    const V3 fn = s->faceNormalsPool[triId];
    p0[0] = a.x, p0[1] = a.y, p0[2] = a.z;
    e0[0] = b.x, e0[1] = b.y, e0[2] = b.z;
    e1[0] = c.x, e1[1] = c.y, e1[2] = c.z;
    n[0] = fn.x, n[1] = fn.y, n[2] = fn.z;
    return;
  }
}
static uint32_t dumpTree(const TreeNode* n, float* boxes, uint32_t* topo, uint32_t* modelOf, uint32_t cap, uint32_t& next)
{
  const uint32_t me = next++;
  if (me < cap)
  {
    boxes[6 * me + 0] = n->box.lo.x, boxes[6 * me + 1] = n->box.lo.y, boxes[6 * me + 2] = n->box.lo.z;
    boxes[6 * me + 3] = n->box.hi.x, boxes[6 * me + 4] = n->box.hi.y, boxes[6 * me + 5] = n->box.hi.z;
    modelOf[me] = n->modelIndex;
    topo[2 * me] = topo[2 * me + 1] = 0xffffffffu;
  }
  if (n->model.numIndices == 0 && n->left)
  {
    const uint32_t l = dumpTree(n->left.get(), boxes, topo, modelOf, cap, next);
    const uint32_t r = dumpTree(n->right.get(), boxes, topo, modelOf, cap, next);
    if (me < cap)
      topo[2 * me] = l, topo[2 * me + 1] = r;
  }
  return me;
}
uint32_t orc_get_model_tree(const orc_scene* s, float* boxes, uint32_t* topo, uint32_t* modelOf, uint32_t cap)
{
  uint32_t next = 0;
  if (s->hierarchy)
    dumpTree(s->hierarchy.get(), boxes, topo, modelOf, cap, next);
  return next;
}

void orc_set_flat(orc_scene* s, const rtflat_model_node* modelNodes, const rtflat_node* nodes, const rtflat_tri* tris,
                  const rtflat_info* info)
{
  s->fModelNodes = modelNodes;
  s->fNodes = nodes;
  s->fTris = tris;
  s->fInfo = *info;
}

int orc_render(orc_scene* s, const orc_camera* cam, const orc_render_params* prm, float* accumulator, float* image,
               uint32_t* aovTriId, uint8_t* aovShadow, float* sampleTimes, orc_counters* counters)
{
  const size_t plane = (size_t)prm->width * prm->height;
  if (prm->sampBegin == 0) // CPURenderer.cpp:140-143
  {
    std::memset(accumulator, 0, plane * 3 * sizeof(float));
    std::memset(image, 0, plane * 3 * sizeof(float));
  }
  if (prm->aovDepths && aovTriId)
  {
    // only the planes of the passes this call renders (the caller's arrays need not reach past sampEnd)
    const size_t first = (size_t)prm->sampBegin * prm->aovDepths * plane, count = (size_t)(prm->sampEnd - prm->sampBegin) * prm->aovDepths * plane;
    std::memset(aovTriId + first, 0xff, count * sizeof(uint32_t));
    std::memset(aovShadow + first, 0xff, count);
  }
  if (counters && prm->sampBegin == 0)
    std::memset(counters, 0, sizeof(*counters));
  uint32_t nw = prm->numWorkers;
  if (nw == 0)
  {
    const uint32_t hc = std::thread::hardware_concurrency();
    nw = hc > 2 ? hc - 2 : 1; // ThreadPool.h:29
  }
  std::mutex counterLock;
  RenderJobCtx ctx{s, toCam(cam), *prm, accumulator, image, aovTriId, aovShadow, &counterLock, counters};
  ColumnPool pool(nw);
  for (uint32_t samp = prm->sampBegin; samp < prm->sampEnd; ++samp) // CPURenderer.cpp:144
  {
    const auto start = std::chrono::high_resolution_clock::now();
    JobBatch batch;
    for (uint32_t w = prm->x0; w < prm->x1; ++w)
      batch.jobs.push_back(columnTask(&ctx, w, samp));
    pool.runBatch(std::move(batch));
    const auto end = std::chrono::high_resolution_clock::now();
    if (sampleTimes)
      sampleTimes[samp] = std::chrono::duration_cast<std::chrono::microseconds>(end - start).count() / 1000.0f;
  }
  return 0;
}

void orc_rand01(uint32_t x, uint32_t y, uint32_t z, float out[3])
{
  const V3 r = rand01(U3{x, y, z});
  out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_primary_ray(const orc_camera* cam, uint32_t width, uint32_t height, uint32_t w, uint32_t h, uint32_t samp, float origin[3],
                     float dir[3], float rnd[3])
{
  V3 r3;
  const RayO r = primaryRay(toCam(cam), width, height, w, h, samp, r3);
  origin[0] = r.origin.x, origin[1] = r.origin.y, origin[2] = r.origin.z;
  dir[0] = r.direction.x, dir[1] = r.direction.y, dir[2] = r.direction.z;
  rnd[0] = r3.x, rnd[1] = r3.y, rnd[2] = r3.z;
}
void orc_sincos(float x, float* sn, float* cs) { sincos_det(x, sn, cs); }
void orc_bounce_dir(const float normalLight[3], const float rnd[3], float out[3])
{
  const V3 d = orc_scene::bounceDirection(V3{normalLight[0], normalLight[1], normalLight[2]}, V3{rnd[0], rnd[1], rnd[2]});
  out[0] = d.x, out[1] = d.y, out[2] = d.z;
}
int orc_aabb_intersects(const float lo[3], const float hi[3], const float o[3], const float d[3], float tmin, float tmax)
{
  Box b;
  b.lo = {lo[0], lo[1], lo[2]};
  b.hi = {hi[0], hi[1], hi[2]};
  return boxHit(b, RayO{{o[0], o[1], o[2]}, {d[0], d[1], d[2]}}, tmin, tmax) ? 1 : 0;
}
uint32_t orc_closest_hit(orc_scene* s, int flat, const float o[3], const float d[3], float tmin, float tmax, float* tOut)
{
  Counters cnt;
  const Hit h = s->closest(flat != 0, RayO{{o[0], o[1], o[2]}, {d[0], d[1], d[2]}}, tmin, tmax, cnt);
  if (tOut)
    *tOut = h.t;
  return (h.t < FLT_MAX) ? h.triId : 0xffffffffu;
}
int orc_any_hit(orc_scene* s, int flat, const float o[3], const float d[3], float tmin, float tmax)
{
  Counters cnt;
  return s->any(flat != 0, RayO{{o[0], o[1], o[2]}, {d[0], d[1], d[2]}}, tmin, tmax, cnt) ? 1 : 0;
}
}
