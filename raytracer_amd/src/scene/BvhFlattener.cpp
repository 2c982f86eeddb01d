#include "scene/BvhFlattener.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

namespace rt
{
namespace
{
static_assert(sizeof(rtflat_node) == 64 && sizeof(rtflat_model_node) == 64 && sizeof(rtflat_tri) == 64, "records are 64 bytes");

// ------------------------------------------------------------------------------------------------
// 1. model-level tree: the reference's agglomerative pairing (src/cpu/CPUScene.cpp:67-105)
// ------------------------------------------------------------------------------------------------
struct PendingModelNode
{
  AABB box;
  int left = -1, right = -1; // indices into the scratch list
  int model = -1;
};

// Repeatedly joins the ordered pair (i, j), i != j, whose union has the smallest surface area; the FIRST strict
// minimum in (i-major, j-minor) scan order wins, the pair is removed (rhs first) and the parent is appended.
// Returns the scratch list; the root is the last element.
std::vector<PendingModelNode> pairModels(const std::vector<AABB>& boxes)
{
  std::vector<PendingModelNode> all;
  std::vector<int> open;
  for (size_t m = 0; m < boxes.size(); ++m)
  {
    all.push_back(PendingModelNode{boxes[m], -1, -1, (int)m});
    open.push_back((int)m);
  }
  while (open.size() > 1)
  {
    size_t lhs = open.size(), rhs = open.size();
    float smallest = FLT_MAX;
    for (size_t i = 0; i < open.size(); ++i)
      for (size_t j = 0; j < open.size(); ++j)
      {
        if (i == j)
          continue;
        const float area = AABB::combine(all[open[i]].box, all[open[j]].box).surfaceArea();
        if (smallest > area)
        {
          smallest = area;
          lhs = i;
          rhs = j;
        }
      }
    PendingModelNode parent;
    parent.box = AABB::combine(all[open[lhs]].box, all[open[rhs]].box);
    parent.left = open[lhs];
    parent.right = open[rhs];
    open.erase(open.begin() + rhs); // rhs > lhs always: (i,j) is met before (j,i)
    open.erase(open.begin() + lhs);
    all.push_back(parent);
    open.push_back((int)all.size() - 1);
  }
  return all;
}

// ------------------------------------------------------------------------------------------------
// 2. triangle-level BVH2: binned SAH over padded triangle boxes
// ------------------------------------------------------------------------------------------------
struct PrimBox
{
  float lo[3], hi[3];
};

// the binary tree the SAH builder produces; it is collapsed into 4-wide quantised nodes afterwards
struct BinaryNode
{
  float lo0[3], hi0[3]; // child 0 box
  float lo1[3], hi1[3]; // child 1 box
  uint32_t child0, child1;
};

inline void growBox(float* lo, float* hi, const float* blo, const float* bhi)
{
  for (int a = 0; a < 3; ++a)
  {
    lo[a] = std::min(lo[a], blo[a]);
    hi[a] = std::max(hi[a], bhi[a]);
  }
}
inline float halfArea(const float* lo, const float* hi)
{
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}

struct BlasBuilder
{
  const PrimBox* prims;   // indexed by model-local triangle index
  uint32_t* order;        // permutation of the model's triangles, partitioned in place
  BinaryNode* scratch;    // node pool shared by all models
  std::atomic<uint32_t>* nextNode;
  uint32_t recordBase;    // first rtflat_tri record of this model
  uint32_t maxLeaf, bins;
  float traversalCost = 1.0f;
  std::atomic<int> spareThreads{0};

  static constexpr uint32_t kMaxBins = 32;
  static constexpr uint32_t kSpawnThreshold = 1u << 16;

  // builds the subtree over order[b, e); returns its child reference and writes its bounds
  uint32_t build(uint32_t b, uint32_t e, float* outLo, float* outHi, uint32_t depth)
  {
    const uint32_t n = e - b;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t k = b; k < e; ++k)
    {
      const PrimBox& p = prims[order[k]];
      growBox(lo, hi, p.lo, p.hi);
      for (int a = 0; a < 3; ++a)
      {
        const float c = 0.5f * (p.lo[a] + p.hi[a]);
        clo[a] = std::min(clo[a], c);
        chi[a] = std::max(chi[a], c);
      }
    }
    std::memcpy(outLo, lo, sizeof(lo));
    std::memcpy(outHi, hi, sizeof(hi));
    if (n == 1)
      return rtflat_make_leaf(recordBase + b, 1);

    // --- choose a split: 3 axes x (bins-1) candidate planes
    uint32_t bestAxis = 3, bestPlane = 0;
    float bestCost = FLT_MAX;
    float scale[3];
    for (int a = 0; a < 3; ++a)
    {
      const float ext = chi[a] - clo[a];
      scale[a] = ext > 0.0f ? (float)bins / ext : 0.0f;
    }
    {
      uint32_t count[3][kMaxBins] = {};
      float blo[3][kMaxBins][3], bhi[3][kMaxBins][3];
      for (int a = 0; a < 3; ++a)
        for (uint32_t i = 0; i < bins; ++i)
          for (int c = 0; c < 3; ++c)
          {
            blo[a][i][c] = FLT_MAX;
            bhi[a][i][c] = -FLT_MAX;
          }
      for (uint32_t k = b; k < e; ++k)
      {
        const PrimBox& p = prims[order[k]];
        for (int a = 0; a < 3; ++a)
        {
          if (scale[a] == 0.0f)
            continue;
          const float c = 0.5f * (p.lo[a] + p.hi[a]);
          const uint32_t bin = std::min(bins - 1, (uint32_t)((c - clo[a]) * scale[a]));
          count[a][bin]++;
          growBox(blo[a][bin], bhi[a][bin], p.lo, p.hi);
        }
      }
      for (int a = 0; a < 3; ++a)
      {
        if (scale[a] == 0.0f)
          continue;
        float rightArea[kMaxBins];
        float rl[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rh[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t i = bins - 1; i > 0; --i)
        {
          if (count[a][i])
            growBox(rl, rh, blo[a][i], bhi[a][i]);
          rightArea[i] = halfArea(rl, rh);
        }
        float ll[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lh[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        uint32_t nl = 0;
        for (uint32_t i = 0; i + 1 < bins; ++i)
        {
          if (count[a][i])
            growBox(ll, lh, blo[a][i], bhi[a][i]);
          nl += count[a][i];
          const uint32_t nr = n - nl;
          if (nl == 0 || nr == 0)
            continue;
          const float cost = halfArea(ll, lh) * (float)nl + rightArea[i + 1] * (float)nr;
          if (cost < bestCost)
          {
            bestCost = cost;
            bestAxis = (uint32_t)a;
            bestPlane = i + 1; // bins [0, bestPlane) go left
          }
        }
      }
    }
    const float parentArea = halfArea(lo, hi);
    if (n <= maxLeaf)
    {
      // leaf cost n*A vs split cost Ct*A + sum(child area * count)
      if (bestAxis == 3 || (float)n * parentArea <= traversalCost * parentArea + bestCost)
        return rtflat_make_leaf(recordBase + b, n);
    }
    uint32_t mid;
    if (bestAxis == 3)
      mid = b + n / 2; // every centroid coincides: split the list in the middle
    else
    {
      const float c0 = clo[bestAxis], sc = scale[bestAxis];
      const uint32_t lastBin = bins - 1;
      uint32_t* first = order + b;
      uint32_t* last = order + e;
      uint32_t* m = std::partition(first, last, [&](uint32_t t) {
        const PrimBox& p = prims[t];
        const float c = 0.5f * (p.lo[bestAxis] + p.hi[bestAxis]);
        return std::min(lastBin, (uint32_t)((c - c0) * sc)) < bestPlane;
      });
      mid = (uint32_t)(m - order);
      if (mid == b || mid == e)
        mid = b + n / 2;
    }

    const uint32_t me = nextNode->fetch_add(1, std::memory_order_relaxed);
    BinaryNode node{};
    uint32_t c0ref, c1ref;
    bool spawned = false;
    if (n >= kSpawnThreshold && spareThreads.fetch_sub(1, std::memory_order_relaxed) > 0)
      spawned = true;
    else if (n >= kSpawnThreshold)
      spareThreads.fetch_add(1, std::memory_order_relaxed);
    if (spawned)
    {
      std::thread helper([&] { c0ref = build(b, mid, node.lo0, node.hi0, depth + 1); });
      c1ref = build(mid, e, node.lo1, node.hi1, depth + 1);
      helper.join();
      spareThreads.fetch_add(1, std::memory_order_relaxed);
    }
    else
    {
      c0ref = build(b, mid, node.lo0, node.hi0, depth + 1);
      c1ref = build(mid, e, node.lo1, node.hi1, depth + 1);
    }
    node.child0 = c0ref;
    node.child1 = c1ref;
    scratch[me] = node;
    return me;
  }
};

// Collapses the binary tree under `root` into 4-wide nodes (the child with the largest box is opened until a node has
// four children or only leaves), quantises the child boxes outward onto the node's 8-bit grid and appends the nodes to
// `out` in depth-first pre-order (first inner child right after its parent).  Returns the reference of the new root and
// the deepest chain of 4-wide nodes through `deepest`.
struct Candidate
{
  uint32_t ref;
  float lo[3], hi[3];
};

void quantiseNode(const Candidate* c, uint32_t n, rtflat_node& node)
{
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t k = 0; k < n; ++k)
    growBox(lo, hi, c[k].lo, c[k].hi);
  node = rtflat_node{};
  node.numChildren = (uint8_t)n;
  if (n == 0)
    lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f;
  for (int a = 0; a < 3; ++a)
  {
    node.origin[a] = lo[a];
    const double ext = (double)hi[a] - (double)lo[a];
    int e = -100;
    if (ext > 0.0)
    {
      e = (int)std::ceil(std::log2(ext / 255.0));
      while (std::ceil(ext / std::ldexp(1.0, e)) > 255.0)
        ++e;
    }
    e = std::max(-126, std::min(e, 126));
    node.exp[a] = (uint8_t)(e + 127);
    const double step = std::ldexp(1.0, e);
    for (uint32_t k = 0; k < 4; ++k)
    {
      if (k >= n)
      {
        node.qlo[a][k] = 255; // inverted box: never hit
        node.qhi[a][k] = 0;
        continue;
      }
      // outward: the grid value at or below lo, at or above hi (all products are exact in double)
      double ql = std::floor(((double)c[k].lo[a] - (double)lo[a]) / step);
      double qh = std::ceil(((double)c[k].hi[a] - (double)lo[a]) / step);
      ql = std::max(0.0, std::min(ql, 255.0));
      qh = std::max(0.0, std::min(qh, 255.0));
      while (ql > 0.0 && (double)lo[a] + ql * step > (double)c[k].lo[a])
        ql -= 1.0;
      while (qh < 255.0 && (double)lo[a] + qh * step < (double)c[k].hi[a])
        qh += 1.0;
      node.qlo[a][k] = (uint8_t)ql;
      node.qhi[a][k] = (uint8_t)qh;
    }
  }
  for (uint32_t k = 0; k < 4; ++k)
    node.child[k] = k < n ? c[k].ref : RTFLAT_INVALID;
}

uint32_t emitWide(const BinaryNode* scratch, uint32_t root, std::vector<rtflat_node>& out, uint32_t& deepest)
{
  if (rtflat_is_leaf(root))
    return root;
  struct Pending
  {
    uint32_t binaryIndex;
    uint32_t parentOut; // where to patch the reference (RTFLAT_INVALID for the root)
    uint32_t slot, depth;
  };
  std::vector<Pending> stack;
  stack.push_back({root, RTFLAT_INVALID, 0, 1});
  uint32_t rootOut = RTFLAT_INVALID;
  auto halfAreaOf = [](const Candidate& c) { return halfArea(c.lo, c.hi); };
  while (!stack.empty())
  {
    const Pending p = stack.back();
    stack.pop_back();
    deepest = std::max(deepest, p.depth);
    Candidate cand[4];
    uint32_t n = 2;
    const BinaryNode& b = scratch[p.binaryIndex];
    cand[0].ref = b.child0, cand[1].ref = b.child1;
    std::memcpy(cand[0].lo, b.lo0, 12), std::memcpy(cand[0].hi, b.hi0, 12);
    std::memcpy(cand[1].lo, b.lo1, 12), std::memcpy(cand[1].hi, b.hi1, 12);
    while (n < 4)
    {
      int pick = -1;
      float bestArea = -1.0f;
      for (uint32_t k = 0; k < n; ++k)
        if (!rtflat_is_leaf(cand[k].ref) && halfAreaOf(cand[k]) > bestArea)
        {
          bestArea = halfAreaOf(cand[k]);
          pick = (int)k;
        }
      if (pick < 0)
        break;
      const BinaryNode& open = scratch[cand[pick].ref];
      for (uint32_t k = n; k > (uint32_t)pick + 1; --k) // keep the children of the opened node next to each other
        cand[k] = cand[k - 1];
      cand[pick].ref = open.child0;
      std::memcpy(cand[pick].lo, open.lo0, 12), std::memcpy(cand[pick].hi, open.hi0, 12);
      cand[pick + 1].ref = open.child1;
      std::memcpy(cand[pick + 1].lo, open.lo1, 12), std::memcpy(cand[pick + 1].hi, open.hi1, 12);
      ++n;
    }
    const uint32_t mine = (uint32_t)out.size();
    rtflat_node node;
    quantiseNode(cand, n, node);
    out.push_back(node);
    if (p.parentOut == RTFLAT_INVALID)
      rootOut = mine;
    else
      out[p.parentOut].child[p.slot] = mine;
    for (uint32_t k = n; k-- > 0;) // pushed last = emitted first: child 0's subtree follows its parent
      if (!rtflat_is_leaf(cand[k].ref))
        stack.push_back({cand[k].ref, mine, k, p.depth + 1});
  }
  return rootOut;
}
} // namespace

FlatScene flattenScene(const ScenePoolsView& pools, const FlattenOptions& opt)
{
  const auto t0 = std::chrono::steady_clock::now();
  FlatScene flat;
  const std::vector<ModelReference>& refs = *pools.refs;
  const size_t numModels = refs.size();
  const uint32_t totalTris = (uint32_t)pools.indicesPool->size();
  const uint32_t maxLeaf = std::max(1u, std::min(opt.maxLeafTriangles, (uint32_t)RTFLAT_MAX_LEAF_TRIS));
  const uint32_t bins = std::max(2u, std::min(opt.sahBins, BlasBuilder::kMaxBins));

  // ---- scene bounds and padding
  AABB sceneBox;
  for (const AABB& b : *pools.modelBoxes)
    sceneBox = AABB::combine(sceneBox, b);
  // model boxes can be tighter than the "effective" triangles when the reference's edge indexing pairs a
  // triangle with another triangle's edges, so take the vertices into account as well
  for (const vec3& p : *pools.positionPool)
    sceneBox.adjust(p);
  const vec3 ext = sceneBox.max - sceneBox.min;
  const float diag = std::sqrt(ext.x * ext.x + ext.y * ext.y + ext.z * ext.z);
  const vec3 centre = (sceneBox.min + sceneBox.max) * 0.5f;
  const float originBound = opt.originBoundScale * diag;
  const float maxCentre = std::max(std::fabs(centre.x), std::max(std::fabs(centre.y), std::fabs(centre.z)));
  const float reach = maxCentre + originBound + diag; // no ray origin or vertex coordinate exceeds this magnitude
  const float padAbs = opt.padUlps * 5.9604645e-8f * reach;

  // ---- model tree (pre-order) with DFS leaf ranks
  std::vector<PendingModelNode> paired = pairModels(*pools.modelBoxes);
  std::vector<uint32_t> orderBase(numModels, 0), recordBase(numModels, 0);
  {
    uint32_t run = 0;
    for (size_t m = 0; m < numModels; ++m)
    {
      recordBase[m] = run;
      run += refs[m].numIndices;
    }
  }
  if (!paired.empty())
  {
    struct Item
    {
      int scratch;
      uint32_t parentOut;
      int slot;
    };
    std::vector<Item> stack{{(int)paired.size() - 1, RTFLAT_INVALID, 0}};
    uint32_t rank = 0;
    while (!stack.empty())
    {
      const Item it = stack.back();
      stack.pop_back();
      const PendingModelNode& src = paired[it.scratch];
      const uint32_t mine = (uint32_t)flat.modelNodes.size();
      rtflat_model_node mn{};
      mn.lo[0] = src.box.min.x, mn.lo[1] = src.box.min.y, mn.lo[2] = src.box.min.z;
      mn.hi[0] = src.box.max.x, mn.hi[1] = src.box.max.y, mn.hi[2] = src.box.max.z;
      mn.left = mn.right = RTFLAT_INVALID;
      mn.blas = RTFLAT_INVALID;
      if (src.model >= 0)
      {
        mn.triCount = refs[src.model].numIndices;
        mn.indicesOffset = refs[src.model].indicesOffset;
        mn.orderBase = rank;
        mn.modelIndex = (uint32_t)src.model;
        orderBase[src.model] = rank;
        rank += refs[src.model].numIndices;
      }
      flat.modelNodes.push_back(mn);
      if (it.parentOut != RTFLAT_INVALID)
      {
        if (it.slot == 0)
          flat.modelNodes[it.parentOut].left = mine;
        else
          flat.modelNodes[it.parentOut].right = mine;
      }
      if (src.model < 0)
      {
        stack.push_back({src.right, mine, 1});
        stack.push_back({src.left, mine, 0}); // left is popped (visited) first: DFS left-to-right ranks
      }
    }
  }

  // subtree sizes (pre-order: children follow their parent), stored in subtreeSize so a traversal can skip a subtree
  for (size_t k = flat.modelNodes.size(); k-- > 0;)
  {
    rtflat_model_node& mn = flat.modelNodes[k];
    mn.subtreeSize = 1;
    if (mn.left != RTFLAT_INVALID)
      mn.subtreeSize += flat.modelNodes[mn.left].subtreeSize + flat.modelNodes[mn.right].subtreeSize;
  }

  // ---- per-triangle operands and padded boxes
  flat.tris.resize(totalTris);
  std::vector<PrimBox> prims(totalTris);
  std::vector<uint32_t> order(totalTris);
  uint32_t numDegenerate = 0;
  for (size_t m = 0; m < numModels; ++m)
  {
    const ModelReference& r = refs[m];
    const size_t edgeBase = opt.referenceEdgeIndexing ? (size_t)r.indicesOffset : (size_t)2 * r.indicesOffset;
    for (uint32_t i = 0; i < r.numIndices; ++i)
    {
      const uint32_t tri = r.indicesOffset + i;
      const vec3 p0 = (*pools.positionPool)[r.positionOffset + (*pools.indicesPool)[tri].x];
      const vec3 e0 = (*pools.edgesPool)[edgeBase + 2 * (size_t)i];
      const vec3 e1 = (*pools.edgesPool)[edgeBase + 2 * (size_t)i + 1];
      const vec3 p1 = p0 + e0, p2 = p0 + e1;
      const vec3 nrm = cross(e0, e1);
      const float area2 = std::sqrt(nrm.x * nrm.x + nrm.y * nrm.y + nrm.z * nrm.z);
      if (nrm.x == 0.0f && nrm.y == 0.0f && nrm.z == 0.0f)
        numDegenerate++;
      // 1/sin of the sharpest corner: the reference's barycentric test is fuzzier for slivers
      const vec3 e2 = e1 - e0;
      float l0 = length(e0), l1 = length(e1), l2 = length(e2);
      if (l0 < l1)
        std::swap(l0, l1);
      if (l1 < l2)
        std::swap(l1, l2);
      if (l0 < l1)
        std::swap(l0, l1); // l0 >= l1 are the two longest edges
      float sliver = (area2 > 0.0f) ? (l0 * l1) / area2 : 64.0f;
      sliver = std::min(64.0f, std::max(1.0f, sliver));
      const float pad = padAbs * sliver;
      PrimBox& pb = prims[tri];
      pb.lo[0] = std::min(p0.x, std::min(p1.x, p2.x)) - pad;
      pb.lo[1] = std::min(p0.y, std::min(p1.y, p2.y)) - pad;
      pb.lo[2] = std::min(p0.z, std::min(p1.z, p2.z)) - pad;
      pb.hi[0] = std::max(p0.x, std::max(p1.x, p2.x)) + pad;
      pb.hi[1] = std::max(p0.y, std::max(p1.y, p2.y)) + pad;
      pb.hi[2] = std::max(p0.z, std::max(p1.z, p2.z)) + pad;
      order[tri] = i; // model-local
    }
  }

  // ---- build one BVH per model into a scratch pool, then emit pre-order
  std::vector<BinaryNode> scratch(opt.deviceBuild ? 1u : std::max(1u, totalTris));
  std::atomic<uint32_t> nextNode{0};
  uint32_t threads = opt.threads ? opt.threads : std::thread::hardware_concurrency();
  threads = std::max(1u, std::min(threads, 16u));
  uint32_t deepest = 0;
  std::vector<uint32_t> blasRoot(numModels, RTFLAT_INVALID);
  for (size_t m = 0; m < numModels && !opt.deviceBuild; ++m)
  {
    const ModelReference& r = refs[m];
    if (r.numIndices == 0)
      continue;
    BlasBuilder bb;
    bb.prims = prims.data() + r.indicesOffset;
    bb.order = order.data() + r.indicesOffset;
    bb.scratch = scratch.data();
    bb.nextNode = &nextNode;
    bb.recordBase = recordBase[m];
    bb.maxLeaf = maxLeaf;
    bb.traversalCost = opt.traversalCost;
    bb.bins = bins;
    bb.spareThreads = (int)threads - 1;
    float lo[3], hi[3];
    blasRoot[m] = bb.build(0, r.numIndices, lo, hi, 0);
  }
  flat.nodes.reserve(nextNode.load() / 2 + 1);
  for (size_t m = 0; m < numModels; ++m)
    if (blasRoot[m] != RTFLAT_INVALID)
      blasRoot[m] = emitWide(scratch.data(), blasRoot[m], flat.nodes, deepest);
  if (flat.nodes.empty())
  {
    // keep the array non-empty for the uploader: one node without children
    rtflat_node none{};
    quantiseNode(nullptr, 0, none);
    flat.nodes.push_back(none);
  }
  for (rtflat_model_node& mn : flat.modelNodes)
    if (mn.triCount > 0)
      mn.blas = blasRoot[mn.modelIndex];

  // ---- triangle records in leaf order
  for (size_t m = 0; m < numModels; ++m)
  {
    const ModelReference& r = refs[m];
    const size_t edgeBase = opt.referenceEdgeIndexing ? (size_t)r.indicesOffset : (size_t)2 * r.indicesOffset;
    for (uint32_t k = 0; k < r.numIndices; ++k)
    {
      const uint32_t i = order[r.indicesOffset + k];
      const uint32_t tri = r.indicesOffset + i;
      const vec3 p0 = (*pools.positionPool)[r.positionOffset + (*pools.indicesPool)[tri].x];
      const vec3 e0 = (*pools.edgesPool)[edgeBase + 2 * (size_t)i];
      const vec3 e1 = (*pools.edgesPool)[edgeBase + 2 * (size_t)i + 1];
      const vec3 fn = (*pools.faceNormalsPool)[tri];
      rtflat_tri& rec = flat.tris[recordBase[m] + k];
      rec.p0[0] = p0.x, rec.p0[1] = p0.y, rec.p0[2] = p0.z;
      rec.e0[0] = e0.x, rec.e0[1] = e0.y, rec.e0[2] = e0.z;
      rec.e1[0] = e1.x, rec.e1[1] = e1.y, rec.e1[2] = e1.z;
      rec.n[0] = fn.x, rec.n[1] = fn.y, rec.n[2] = fn.z;
      rec.triId = tri;
      rec.orderKey = orderBase[m] + i;
      rec.pad[0] = rec.pad[1] = 0;
    }
  }
  if (flat.tris.empty())
    flat.tris.push_back(rtflat_tri{});

  rtflat_info& info = flat.info;
  info.numModelNodes = (uint32_t)flat.modelNodes.size();
  info.numNodes = (uint32_t)flat.nodes.size();
  info.numTris = totalTris;
  info.rootModelNode = 0;
  info.sceneLo[0] = sceneBox.min.x, info.sceneLo[1] = sceneBox.min.y, info.sceneLo[2] = sceneBox.min.z;
  info.sceneHi[0] = sceneBox.max.x, info.sceneHi[1] = sceneBox.max.y, info.sceneHi[2] = sceneBox.max.z;
  info.padAbs = padAbs;
  info.originBound = originBound;
  info.cullRel = opt.cullRel;
  info.cullAbs = diag * (1.0f / 1048576.0f);
  info.maxDepthBlas = deepest;
  info.numDegenerate = numDegenerate;
  flat.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return flat;
}
} // namespace rt
