// rt_lbvh.h: device-side build of the triangle-level BVH below a model leaf -- part of the gfx950 kernel
// library.  The reference has nothing here (it scans a model's triangles linearly, src/cpu/CPUScene.cpp:143,191); the host
// flattener's binned-SAH builder (src/scene/BvhFlattener.cpp) stays as the quality baseline, this is the fast path:
//
//   k_lbvh_prims     padded triangle boxes (same rule as the flattener: padAbs x sliver factor) + 63-bit Morton codes of
//                    their centres; the code's bits are dealt to the axes longest-extent-first (a terrain two units wide and a
//                    third of a unit high gets its first height bit only after x and z have been halved twice or three times)
//   rocprim          radix sort of (code, triangle) pairs
//   binary tree, one of
//    quality 0 (LBVH):  k_lbvh_topology  binary radix tree over the sorted codes (Karras 2012), one thread per inner node
//                       k_lbvh_refit     boxes bottom-up, one launch per tree level (a node is finished once both children were
//                                        finished by an EARLIER launch: kernel boundaries order the box writes); on the way a node
//                                        re-pairs its grandchildren when that gives its children smaller boxes (tree rotations;
//                                        the refit runs twice for two rounds of them)
//    quality 1 (PLOC, Meister & Bittner 2018: parallel locally-ordered clustering; round 3, the default):
//                       k_ploc_nearest   every cluster of the Morton-ordered cluster array finds, among its `radius` neighbours
//                                        on either side, the one whose union with it has the smallest surface area
//                       k_ploc_merge     mutual nearest neighbours become one inner node (box = the union: no refit pass)
//                       rocprim scan + k_ploc_compact   the surviving clusters close ranks, order kept
//                                        ... repeated until one cluster is left (about 35 rounds for 10 M triangles)
//   k_lbvh_emit      collapse to 4-wide nodes level by level from the root (the child with the largest box is opened until
//                    the node has four children or only leaves), child boxes quantised outward onto the node's 8-bit grid:
//                    the same rtflat_node records the host builder writes, in breadth-first order; triangle ranges are dealt
//                    top-down (a node's triangles are a contiguous run of the final record order) and every leaf writes which
//                    triangles it holds
//   k_lbvh_gather    triangle records into that (leaf) order
//
// Hit results do not depend on the tree (closest hits are an arg-min with the orderKey tie rule, any-hits a disjunction), so
// a scene renders bit-identically through either builder.
#pragma once
#include "rt_kernel_common.h"

namespace rtk
{
struct LbvhBox
{
  float lo[3], hi[3];
};

struct LbvhNode // binary tree node (radix tree or PLOC)
{
  uint32_t left, right; // bit 31 set: a leaf = sorted triangle index in the low bits
  uint32_t count;       // triangles below it
  uint32_t pad;
};

// where each bit of the three 21-bit quantised coordinates goes in the 63-bit code (bit 20 of an axis first)
struct MortonLayout
{
  uint8_t pos[3][21];
};

// the padded box of a triangle record: same rule as the flattener (padAbs x sliver factor, see BvhFlattener.cpp)
__device__ __forceinline__ LbvhBox lbvhTriangleBox(const rtflat_tri& t, float padAbs)
{
  const vec3 p0{t.p0[0], t.p0[1], t.p0[2]}, e0{t.e0[0], t.e0[1], t.e0[2]}, e1{t.e1[0], t.e1[1], t.e1[2]};
  const vec3 p1 = p0 + e0, p2 = p0 + e1;
  const vec3 nrm = rt::cross(e0, e1);
  const float area2 = rt::rt_sqrt(nrm.x * nrm.x + nrm.y * nrm.y + nrm.z * nrm.z);
  // 1/sin of the sharpest corner: the reference's barycentric test is fuzzier for slivers (see BvhFlattener.cpp)
  float l0 = rt::length(e0), l1 = rt::length(e1), l2 = rt::length(e1 - e0);
  if (l0 < l1)
  {
    const float s = l0;
    l0 = l1, l1 = s;
  }
  if (l1 < l2)
  {
    const float s = l1;
    l1 = l2, l2 = s;
  }
  if (l0 < l1)
  {
    const float s = l0;
    l0 = l1, l1 = s;
  }
  float sliver = (area2 > 0.0f) ? (l0 * l1) / area2 : 64.0f;
  sliver = fminf(64.0f, fmaxf(1.0f, sliver));
  const float pad = padAbs * sliver;
  LbvhBox b;
  b.lo[0] = fminf(p0.x, fminf(p1.x, p2.x)) - pad, b.lo[1] = fminf(p0.y, fminf(p1.y, p2.y)) - pad, b.lo[2] = fminf(p0.z, fminf(p1.z, p2.z)) - pad;
  b.hi[0] = fmaxf(p0.x, fmaxf(p1.x, p2.x)) + pad, b.hi[1] = fmaxf(p0.y, fmaxf(p1.y, p2.y)) + pad, b.hi[2] = fmaxf(p0.z, fmaxf(p1.z, p2.z)) + pad;
  return b;
}

// tris: the model's records in model order; boxes/codes/ids: n entries
__global__ __launch_bounds__(kBlock) void k_lbvh_prims(const rtflat_tri* tris, uint32_t n, float padAbs, float3 sceneLo, float3 sceneInvExt, MortonLayout layout,
                                                       LbvhBox* boxes, uint64_t* codes, uint32_t* ids)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n)
    return;
  const LbvhBox b = lbvhTriangleBox(tris[i], padAbs);
  boxes[i] = b;
  const float cx = (0.5f * (b.lo[0] + b.hi[0]) - sceneLo.x) * sceneInvExt.x, cy = (0.5f * (b.lo[1] + b.hi[1]) - sceneLo.y) * sceneInvExt.y,
              cz = (0.5f * (b.lo[2] + b.hi[2]) - sceneLo.z) * sceneInvExt.z;
  const uint32_t qx = (uint32_t)(fminf(fmaxf(cx, 0.0f), 1.0f) * 2097151.0f), qy = (uint32_t)(fminf(fmaxf(cy, 0.0f), 1.0f) * 2097151.0f),
                 qz = (uint32_t)(fminf(fmaxf(cz, 0.0f), 1.0f) * 2097151.0f);
  uint64_t code = 0;
#pragma unroll
  for (int k = 0; k < 21; ++k)
  {
    code |= (uint64_t)((qx >> (20 - k)) & 1u) << layout.pos[0][k];
    code |= (uint64_t)((qy >> (20 - k)) & 1u) << layout.pos[1][k];
    code |= (uint64_t)((qz >> (20 - k)) & 1u) << layout.pos[2][k];
  }
  codes[i] = code;
  ids[i] = i;
}

// common prefix length of the codes at sorted positions i and j (equal codes are told apart by their positions), -1 outside
__device__ __forceinline__ int lbvhDelta(const uint64_t* codes, int n, int i, int j)
{
  if (j < 0 || j >= n)
    return -1;
  const uint64_t a = codes[i], b = codes[j];
  if (a == b)
    return 64 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clzll((long long)(a ^ b));
}

// one thread per inner node of the binary radix tree (n - 1 of them); node 0 is the root
__global__ __launch_bounds__(kBlock) void k_lbvh_topology(const uint64_t* codes, uint32_t n, LbvhNode* nodes)
{
  const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
  const int N = (int)n;
  if (i >= N - 1)
    return;
  const int d = (lbvhDelta(codes, N, i, i + 1) - lbvhDelta(codes, N, i, i - 1)) >= 0 ? 1 : -1;
  const int deltaMin = lbvhDelta(codes, N, i, i - d);
  int lmax = 2;
  while (lbvhDelta(codes, N, i, i + lmax * d) > deltaMin)
    lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (lbvhDelta(codes, N, i, i + (l + t) * d) > deltaMin)
      l += t;
  const int j = i + l * d;
  const int deltaNode = lbvhDelta(codes, N, i, j);
  int s = 0, t = l;
  do
  {
    t = (t + 1) >> 1;
    if (lbvhDelta(codes, N, i, i + (s + t) * d) > deltaNode)
      s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? d : 0);
  const int first = i < j ? i : j, last = i < j ? j : i;
  LbvhNode nd;
  nd.left = (first == gamma) ? (0x80000000u | (uint32_t)gamma) : (uint32_t)gamma;
  nd.right = (last == gamma + 1) ? (0x80000000u | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
  nd.count = (uint32_t)(last - first + 1), nd.pad = 0;
  nodes[i] = nd;
}

__device__ __forceinline__ LbvhBox lbvhUnion(const LbvhBox& a, const LbvhBox& b)
{
  LbvhBox r;
#pragma unroll
  for (int k = 0; k < 3; ++k)
  {
    r.lo[k] = fminf(a.lo[k], b.lo[k]);
    r.hi[k] = fmaxf(a.hi[k], b.hi[k]);
  }
  return r;
}

// One level of the bottom-up refit.  done[i] = the pass (>= 1) in which inner node i got its box; a child counts as finished
// only if an EARLIER pass finished it, so every box this launch reads was written before the launch began.
// ids: sorted position -> triangle (leaf boxes are the triangles' own)
__device__ __forceinline__ float lbvhHalfArea(const LbvhBox& b)
{
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return dx * dy + dy * dz + dz * dx;
}

// rotate != 0: before a node takes its box, its (up to four) grandchildren are re-paired if another pairing gives the two children
// smaller boxes (sum of their surface areas; a leaf child may trade places with a grandchild).  The node owns its children --
// both were finished by earlier launches and have no other parent -- so rewriting them races with nothing; its own box is the
// union of the same triangles either way.  (Bottom-up tree rotations, Kensler 2008, done in the refit's own passes.)
__global__ __launch_bounds__(kBlock) void k_lbvh_refit(LbvhNode* nodes, uint32_t numInner, const LbvhBox* primBoxes, const uint32_t* ids, LbvhBox* innerBoxes,
                                                       uint8_t* done, uint32_t pass, uint32_t rotate)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= numInner || done[i])
    return;
  const LbvhNode nd = nodes[i];
  const bool lLeaf = (nd.left & 0x80000000u) != 0, rLeaf = (nd.right & 0x80000000u) != 0;
  const uint32_t li = nd.left & 0x7fffffffu, ri = nd.right & 0x7fffffffu;
  if (!lLeaf && !(done[li] && done[li] < pass))
    return;
  if (!rLeaf && !(done[ri] && done[ri] < pass))
    return;
  auto boxOf = [&](uint32_t ref) { return (ref & 0x80000000u) ? primBoxes[ids[ref & 0x7fffffffu]] : innerBoxes[ref]; };
  auto countOf = [&](uint32_t ref) { return (ref & 0x80000000u) ? 1u : nodes[ref].count; };
  LbvhBox a = lLeaf ? primBoxes[ids[li]] : innerBoxes[li];
  LbvhBox b = rLeaf ? primBoxes[ids[ri]] : innerBoxes[ri];
  if (rotate && !(lLeaf && rLeaf))
  {
    if (!lLeaf && !rLeaf)
    {
      // grandchildren (l1 l2 | r1 r2): the other two pairings
      const LbvhNode L = nodes[li], R = nodes[ri];
      const LbvhBox l1 = boxOf(L.left), l2 = boxOf(L.right), r1 = boxOf(R.left), r2 = boxOf(R.right);
      const LbvhBox p11 = lbvhUnion(l1, r1), p22 = lbvhUnion(l2, r2), p12 = lbvhUnion(l1, r2), p21 = lbvhUnion(l2, r1);
      const float keep = lbvhHalfArea(a) + lbvhHalfArea(b);
      const float alt1 = lbvhHalfArea(p11) + lbvhHalfArea(p22), alt2 = lbvhHalfArea(p12) + lbvhHalfArea(p21);
      if (alt1 < keep && alt1 <= alt2)
      {
        nodes[li] = LbvhNode{L.left, R.left, countOf(L.left) + countOf(R.left), 0u};
        nodes[ri] = LbvhNode{L.right, R.right, countOf(L.right) + countOf(R.right), 0u};
        innerBoxes[li] = a = p11, innerBoxes[ri] = b = p22;
      }
      else if (alt2 < keep)
      {
        nodes[li] = LbvhNode{L.left, R.right, countOf(L.left) + countOf(R.right), 0u};
        nodes[ri] = LbvhNode{L.right, R.left, countOf(L.right) + countOf(R.left), 0u};
        innerBoxes[li] = a = p12, innerBoxes[ri] = b = p21;
      }
    }
    else
    {
      // one child is a triangle: it may trade places with one of the other child's children
      const uint32_t inner = lLeaf ? ri : li, leafRef = lLeaf ? nd.left : nd.right;
      const LbvhNode I = nodes[inner];
      const LbvhBox t = lLeaf ? a : b, inBox = lLeaf ? b : a;
      const LbvhBox c1 = boxOf(I.left), c2 = boxOf(I.right);
      const float keep = lbvhHalfArea(inBox); // (the triangle's own box counts the same in every arrangement: one of the three is always alone)
      const LbvhBox w1 = lbvhUnion(t, c2), w2 = lbvhUnion(c1, t); // I.left alone / I.right alone
      const float alt1 = lbvhHalfArea(w1) + lbvhHalfArea(c1) - lbvhHalfArea(t), alt2 = lbvhHalfArea(w2) + lbvhHalfArea(c2) - lbvhHalfArea(t);
      uint32_t alone = 0xffffffffu;
      if (alt1 < keep && alt1 <= alt2)
      {
        alone = I.left;
        nodes[inner] = LbvhNode{leafRef, I.right, 1u + countOf(I.right), 0u};
        innerBoxes[inner] = w1;
      }
      else if (alt2 < keep)
      {
        alone = I.right;
        nodes[inner] = LbvhNode{I.left, leafRef, countOf(I.left) + 1u, 0u};
        innerBoxes[inner] = w2;
      }
      if (alone != 0xffffffffu)
      {
        // the node's children: the grandchild that now stands alone, and the rebuilt inner node
        const LbvhBox aloneBox = boxOf(alone), rebuilt = innerBoxes[inner];
        nodes[i] = lLeaf ? LbvhNode{alone, inner, nd.count, 0u} : LbvhNode{inner, alone, nd.count, 0u};
        a = lLeaf ? aloneBox : rebuilt;
        b = lLeaf ? rebuilt : aloneBox;
        // (an inner `alone` was finished by an earlier pass, like `inner`: the done[] bookkeeping stands)
      }
    }
  }
  innerBoxes[i] = lbvhUnion(a, b);
  done[i] = (uint8_t)pass;
}

// ------------------------------------------------------------------------------------------------ PLOC
constexpr uint32_t kPlocMaxRadius = 32;

// nearest[i] = the cluster within `radius` positions of i whose union with cluster i has the smallest surface area (the
// lower position on a tie).  A block stages its clusters' boxes and the radius on either side in LDS.
__global__ __launch_bounds__(kBlock) void k_ploc_nearest(const LbvhBox* boxes, uint32_t count, uint32_t radius, uint32_t* nearest)
{
  __shared__ LbvhBox sBox[kBlock + 2 * kPlocMaxRadius];
  const int base = (int)(blockIdx.x * kBlock) - (int)radius;
  for (uint32_t k = threadIdx.x; k < kBlock + 2 * radius; k += kBlock)
  {
    const int src = base + (int)k;
    if (src >= 0 && src < (int)count)
      sBox[k] = boxes[src];
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count)
    return;
  const LbvhBox mine = sBox[threadIdx.x + radius];
  float best = FLT_MAX;
  uint32_t bestJ = i; // (count >= 2 whenever this runs, so a neighbour exists)
  const int lo = max(0, (int)i - (int)radius), hi = min((int)count - 1, (int)i + (int)radius);
  for (int j = lo; j <= hi; ++j)
  {
    if (j == (int)i)
      continue;
    const float a = lbvhHalfArea(lbvhUnion(mine, sBox[j - base]));
    if (a < best)
      best = a, bestJ = (uint32_t)j;
  }
  nearest[i] = bestJ;
}

// mutual nearest neighbours (i, j), i < j, merge into a new inner node that takes i's place; j's place is flagged empty.
// Everything else stays.  In place: a pair is handled by the thread of its lower position, which alone reads and writes both.
__global__ __launch_bounds__(kBlock) void k_ploc_merge(uint32_t* refs, LbvhBox* boxes, uint32_t count, const uint32_t* nearest, uint32_t* nodeCounter, LbvhNode* nodes,
                                                       LbvhBox* innerBoxes, uint32_t* flags)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count)
    return;
  const uint32_t j = nearest[i];
  uint32_t keep = 1;
  if (j != i && nearest[j] == i)
  {
    if (i < j)
    {
      const uint32_t ri = refs[i], rj = refs[j];
      const LbvhBox u = lbvhUnion(boxes[i], boxes[j]);
      const uint32_t idx = atomicAdd(nodeCounter, 1u);
      LbvhNode nd;
      nd.left = ri, nd.right = rj;
      nd.count = ((ri & 0x80000000u) ? 1u : nodes[ri].count) + ((rj & 0x80000000u) ? 1u : nodes[rj].count);
      nd.pad = 0;
      nodes[idx] = nd;
      innerBoxes[idx] = u;
      refs[i] = idx;
      boxes[i] = u;
    }
    else
      keep = 0;
  }
  flags[i] = keep;
}

// the surviving clusters close ranks (order kept); the last thread publishes the new count
__global__ __launch_bounds__(kBlock) void k_ploc_compact(const uint32_t* refs, const LbvhBox* boxes, const uint32_t* flags, const uint32_t* offsets, uint32_t count,
                                                         uint32_t* refsOut, LbvhBox* boxesOut, uint32_t* countOut)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count)
    return;
  if (flags[i])
  {
    refsOut[offsets[i]] = refs[i];
    boxesOut[offsets[i]] = boxes[i];
  }
  if (i == count - 1)
    *countOut = offsets[i] + flags[i];
}

// the initial cluster array: one cluster per triangle, in Morton order
__global__ __launch_bounds__(kBlock) void k_ploc_init(const LbvhBox* primBoxes, const uint32_t* ids, uint32_t n, uint32_t* refs, LbvhBox* boxes)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n)
    return;
  refs[i] = 0x80000000u | i;
  boxes[i] = primBoxes[ids[i]];
}

struct LbvhEmitArgs
{
  const LbvhNode* nodes;
  const LbvhBox* innerBoxes;
  const LbvhBox* primBoxes;
  const uint32_t* ids;
  rtflat_node* out;       // the scene's node array
  uint32_t outCapacity;   // entries of `out`
  uint32_t* nodeCounter;  // next free entry of `out`
  uint32_t recordBase;    // first rtflat_tri record of this model (leaf references are absolute)
  uint32_t maxLeaf;       // a subtree of at most this many triangles may become a leaf ...
  float traversalCost;    // ... and does when the surface-area heuristic prefers it: cost of visiting a node relative to a triangle test
  uint32_t* order;        // [first + k] = the (model-order) triangle that ends up as record recordBase + first + k
  const uint4* frontierIn;  // (binary inner node, its entry in `out`, first record of its triangle range, -)
  uint4* frontierOut;
  const uint32_t* countIn;
  uint32_t* countOut;
  uint32_t* overflow;
};

struct LbvhCandidate
{
  uint32_t ref; // binary child reference (bit 31: single sorted triangle)
  LbvhBox box;
  uint32_t count;
};

__device__ __forceinline__ LbvhCandidate lbvhCandidate(const LbvhEmitArgs& a, uint32_t ref)
{
  LbvhCandidate c;
  c.ref = ref;
  if (ref & 0x80000000u)
  {
    const uint32_t k = ref & 0x7fffffffu;
    c.box = a.primBoxes[a.ids[k]];
    c.count = 1;
  }
  else
  {
    c.box = a.innerBoxes[ref];
    c.count = a.nodes[ref].count;
  }
  return c;
}

// Does this subtree end here?  A single triangle does; a subtree of at most maxLeaf triangles does when testing all of them on
// every visit is no dearer, by the surface-area heuristic, than one more node visit plus its two halves as leaves -- the host
// builder's rule (BvhFlattener.cpp: n * A <= traversalCost * A + sum(n_child * A_child)).  Two triangles of one quad stay
// together (their boxes nearly coincide); four scattered soup triangles do not (round 3: taking every subtree of <= 4 triangles
// as a leaf made the 1 M soup test 22 triangles per ray against the host tree's 5.6).
__device__ __forceinline__ bool lbvhEndsHere(const LbvhEmitArgs& a, const LbvhCandidate& c)
{
  if (c.ref & 0x80000000u)
    return true;
  if (c.count > a.maxLeaf)
    return false;
  const LbvhNode nd = a.nodes[c.ref];
  const LbvhCandidate l = lbvhCandidate(a, nd.left), r = lbvhCandidate(a, nd.right);
  const float area = lbvhHalfArea(c.box);
  return (float)c.count * area <= a.traversalCost * area + ((float)l.count * lbvhHalfArea(l.box) + (float)r.count * lbvhHalfArea(r.box));
}

// quantises the child boxes outward onto the node's 8-bit grid (the device twin of quantiseNode in BvhFlattener.cpp)
__device__ __forceinline__ void lbvhQuantise(const LbvhCandidate* c, uint32_t n, const uint32_t* childRef, rtflat_node& node)
{
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t k = 0; k < n; ++k)
    for (int a = 0; a < 3; ++a)
    {
      lo[a] = fminf(lo[a], c[k].box.lo[a]);
      hi[a] = fmaxf(hi[a], c[k].box.hi[a]);
    }
  node.numChildren = (uint8_t)n;
  node.pad[0] = node.pad[1] = 0;
  for (int a = 0; a < 3; ++a)
  {
    node.origin[a] = lo[a];
    const double ext = (double)hi[a] - (double)lo[a];
    int e = -100;
    if (ext > 0.0)
    {
      e = (int)ceil(log2(ext / 255.0));
      while (ceil(ext / ldexp(1.0, e)) > 255.0)
        ++e;
    }
    e = max(-126, min(e, 126));
    node.exp[a] = (uint8_t)(e + 127);
    const double step = ldexp(1.0, e);
    for (uint32_t k = 0; k < 4; ++k)
    {
      if (k >= n)
      {
        node.qlo[a][k] = 255; // inverted box: never hit
        node.qhi[a][k] = 0;
        continue;
      }
      double ql = floor(((double)c[k].box.lo[a] - (double)lo[a]) / step);
      double qh = ceil(((double)c[k].box.hi[a] - (double)lo[a]) / step);
      ql = fmax(0.0, fmin(ql, 255.0));
      qh = fmax(0.0, fmin(qh, 255.0));
      while (ql > 0.0 && (double)lo[a] + ql * step > (double)c[k].box.lo[a])
        ql -= 1.0;
      while (qh < 255.0 && (double)lo[a] + qh * step < (double)c[k].box.hi[a])
        qh += 1.0;
      node.qlo[a][k] = (uint8_t)ql;
      node.qhi[a][k] = (uint8_t)qh;
    }
  }
  for (uint32_t k = 0; k < 4; ++k)
    node.child[k] = k < n ? childRef[k] : RTFLAT_INVALID;
}

// one level of the top-down collapse: every frontier entry becomes one 4-wide node; its inner children get entries of their
// own (allocated after their parent's: a child's index is always larger) and form the next frontier
__global__ __launch_bounds__(kBlock) void k_lbvh_emit(LbvhEmitArgs a)
{
  const uint32_t total = *a.countIn;
  for (uint32_t item = blockIdx.x * kBlock + threadIdx.x; item < total; item += gridDim.x * kBlock)
  {
    const uint4 fr = a.frontierIn[item];
    const LbvhNode nd = a.nodes[fr.x];
    LbvhCandidate cand[4];
    uint32_t n = 2;
    cand[0] = lbvhCandidate(a, nd.left);
    cand[1] = lbvhCandidate(a, nd.right);
    while (n < 4)
    {
      int pick = -1;
      float bestArea = -1.0f;
      for (uint32_t k = 0; k < n; ++k)
        if (!lbvhEndsHere(a, cand[k]))
        {
          const float dx = cand[k].box.hi[0] - cand[k].box.lo[0], dy = cand[k].box.hi[1] - cand[k].box.lo[1], dz = cand[k].box.hi[2] - cand[k].box.lo[2];
          const float area = dx * dy + dy * dz + dz * dx;
          if (area > bestArea)
          {
            bestArea = area;
            pick = (int)k;
          }
        }
      if (pick < 0)
        break;
      const LbvhNode open = a.nodes[cand[pick].ref];
      for (uint32_t k = n; k > (uint32_t)pick + 1; --k) // keep the children of the opened node next to each other
        cand[k] = cand[k - 1];
      cand[pick] = lbvhCandidate(a, open.left);
      cand[pick + 1] = lbvhCandidate(a, open.right);
      ++n;
    }
    uint32_t childRef[4] = {RTFLAT_INVALID, RTFLAT_INVALID, RTFLAT_INVALID, RTFLAT_INVALID};
    // the node's triangles are the run [fr.z, fr.z + count) of the final record order, dealt to the children in their order
    uint32_t first = fr.z;
    for (uint32_t k = 0; k < n; first += cand[k].count, ++k)
    {
      if (lbvhEndsHere(a, cand[k]))
      {
        childRef[k] = rtflat_make_leaf(a.recordBase + first, cand[k].count);
        // which triangles: the leaves of the (at most maxLeaf-triangle) subtree, left to right
        uint32_t stack[RTFLAT_MAX_LEAF_TRIS];
        uint32_t sp = 0, w = first, cur = cand[k].ref;
        for (;;)
        {
          if (cur & 0x80000000u)
          {
            a.order[w++] = a.ids[cur & 0x7fffffffu];
            if (sp == 0)
              break;
            cur = stack[--sp];
          }
          else
          {
            const LbvhNode sub = a.nodes[cur];
            if (sp < RTFLAT_MAX_LEAF_TRIS)
              stack[sp++] = sub.right;
            cur = sub.left;
          }
        }
      }
      else
      {
        const uint32_t idx = atomicAdd(a.nodeCounter, 1u);
        if (idx >= a.outCapacity)
        {
          atomicExch(a.overflow, 1u);
          childRef[k] = rtflat_make_leaf(a.recordBase + first, 1); // keeps the arrays well-formed; the build is rejected
          continue;
        }
        childRef[k] = idx;
        a.frontierOut[atomicAdd(a.countOut, 1u)] = make_uint4(cand[k].ref, idx, first, 0u);
      }
    }
    rtflat_node node;
    lbvhQuantise(cand, n, childRef, node);
    a.out[fr.y] = node;
  }
}

// ---- depth-first node order.  The collapse emits the wide tree level by level (breadth-first): a subtree's nodes end up
// scattered over the whole array, and an incoherent ray's walk down a subtree touches a new memory region at every level.
// The host builder writes depth-first pre-order, where a subtree is one contiguous run; these three kernels put the device's
// tree into the same order (sizes bottom-up, offsets top-down, one launch per level each, then one permutation).
__global__ __launch_bounds__(kBlock) void k_dfs_sizes(const rtflat_node* nodes, uint32_t levelBegin, uint32_t levelCount, uint32_t* sizes)
{
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= levelCount)
    return;
  const uint32_t i = levelBegin + k;
  const rtflat_node& n = nodes[i];
  uint32_t total = 1;
  for (uint32_t c = 0; c < n.numChildren && c < 4; ++c)
    if (!rtflat_is_leaf(n.child[c]) && n.child[c] != RTFLAT_INVALID)
      total += sizes[n.child[c]]; // a child sits on a deeper level: written by an earlier launch
  sizes[i] = total;
}
__global__ __launch_bounds__(kBlock) void k_dfs_offsets(const rtflat_node* nodes, uint32_t levelBegin, uint32_t levelCount, const uint32_t* sizes, uint32_t* newIndex)
{
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= levelCount)
    return;
  const uint32_t i = levelBegin + k;
  const rtflat_node& n = nodes[i];
  uint32_t next = newIndex[i] + 1; // (this node's own place was set by its parent's level, the root's by the host)
  for (uint32_t c = 0; c < n.numChildren && c < 4; ++c)
    if (!rtflat_is_leaf(n.child[c]) && n.child[c] != RTFLAT_INVALID)
    {
      newIndex[n.child[c]] = next;
      next += sizes[n.child[c]];
    }
}
__global__ __launch_bounds__(kBlock) void k_dfs_permute(const rtflat_node* in, uint32_t first, uint32_t count, const uint32_t* newIndex, rtflat_node* out)
{
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= count)
    return;
  rtflat_node n = in[first + k];
  for (uint32_t c = 0; c < 4; ++c)
    if (n.child[c] != RTFLAT_INVALID && !rtflat_is_leaf(n.child[c]))
      n.child[c] = newIndex[n.child[c]];
  out[newIndex[first + k]] = n;
}

// triangle records into leaf order (order[i] = the model-order triangle that becomes record i)
__global__ __launch_bounds__(kBlock) void k_lbvh_gather(const rtflat_tri* in, const uint32_t* ids, uint32_t n, rtflat_tri* out)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n)
    return;
  const float4* src = reinterpret_cast<const float4*>(in + ids[i]);
  float4* dst = reinterpret_cast<float4*>(out + i);
  dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
}

// ------------------------------------------------------------------------------------------------ refit (rthip_refit)
// new geometry into the records of an existing tree: record r keeps its place, its triId and its orderKey and takes the
// operands of triangle triId from the caller's model-order array; triangles that leave the box the scene was flattened for
// raise `outside` (the padding of every box was sized for that box)
__global__ __launch_bounds__(kBlock) void k_refit_tris(rtflat_tri* tris, const rtflat_tri* modelOrder, uint32_t n, float padAbs, float3 lo, float3 hi, uint32_t* outside)
{
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n)
    return;
  rtflat_tri t = tris[r];
  const rtflat_tri src = modelOrder[t.triId];
  for (int k = 0; k < 3; ++k)
    t.p0[k] = src.p0[k], t.e0[k] = src.e0[k], t.e1[k] = src.e1[k], t.n[k] = src.n[k];
  tris[r] = t;
  const LbvhBox b = lbvhTriangleBox(t, padAbs);
  if (!(b.lo[0] >= lo.x && b.lo[1] >= lo.y && b.lo[2] >= lo.z && b.hi[0] <= hi.x && b.hi[1] <= hi.y && b.hi[2] <= hi.z))
    atomicOr(outside, 1u);
}

// One level of the bottom-up refit of the 4-wide tree: a node whose inner children all got their boxes in EARLIER passes
// recomputes its children's boxes (leaf children from their triangles, with the flattener's padding; inner children from
// nodeBoxes) and quantises them afresh onto a grid of its own.  Topology, child order and leaf contents stay.
__global__ __launch_bounds__(kBlock) void k_refit_nodes(rtflat_node* nodes, uint32_t numNodes, const rtflat_tri* tris, float padAbs, LbvhBox* nodeBoxes, uint8_t* done,
                                                        uint32_t pass)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= numNodes || done[i])
    return;
  rtflat_node node = nodes[i];
  const uint32_t n = node.numChildren;
  LbvhCandidate cand[4];
  uint32_t childRef[4];
  for (uint32_t k = 0; k < 4; ++k)
    childRef[k] = node.child[k];
  for (uint32_t k = 0; k < n && k < 4; ++k)
  {
    const uint32_t ref = node.child[k];
    cand[k].ref = ref, cand[k].count = 0;
    if (rtflat_is_leaf(ref))
    {
      const uint32_t first = rtflat_leaf_first(ref), count = rtflat_leaf_count(ref);
      LbvhBox b = lbvhTriangleBox(tris[first], padAbs);
      for (uint32_t t = 1; t < count; ++t)
        b = lbvhUnion(b, lbvhTriangleBox(tris[first + t], padAbs));
      cand[k].box = b;
    }
    else
    {
      if (ref >= numNodes || !(done[ref] && done[ref] < pass))
        return; // not yet: a later pass
      cand[k].box = nodeBoxes[ref];
    }
  }
  if (n == 0)
  {
    done[i] = (uint8_t)pass; // the placeholder node of a scene whose models are all single leaves
    return;
  }
  LbvhBox all = cand[0].box;
  for (uint32_t k = 1; k < n; ++k)
    all = lbvhUnion(all, cand[k].box);
  lbvhQuantise(cand, n, childRef, node);
  nodes[i] = node;
  nodeBoxes[i] = all;
  done[i] = (uint8_t)pass;
}

// the packet kernel's copy of the nodes (FatNode, rt_kernel_common.h): every child plane dequantised in double and rounded
// outward to float, so a fat box contains the quantised box it stands for
__global__ __launch_bounds__(kBlock) void k_fatten(const rtflat_node* nodes, uint32_t numNodes, FatNode* fat)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= numNodes)
    return;
  const rtflat_node n = nodes[i];
  FatNode f;
  for (int a = 0; a < 3; ++a)
  {
    const double step = ldexp(1.0, (int)n.exp[a] - 127);
    for (int k = 0; k < 4; ++k)
    {
      const bool used = k < (int)n.numChildren && n.child[k] != RTFLAT_INVALID;
      f.plane[2 * a][k] = used ? __double2float_rd((double)n.origin[a] + (double)n.qlo[a][k] * step) : FLT_MAX;
      f.plane[2 * a + 1][k] = used ? __double2float_ru((double)n.origin[a] + (double)n.qhi[a][k] * step) : -FLT_MAX;
    }
  }
  for (int k = 0; k < 4; ++k)
    f.child[k] = n.child[k], f.pad[k] = 0;
  fat[i] = f;
}

// the packet kernel's per-light triangle operands (DeviceScene::shadowOperands): one thread per record, every directional light
__global__ __launch_bounds__(kBlock) void k_shadow_operands(const rtflat_tri* tris, uint32_t numTris, const Lights* lights, uint32_t numLights, float4* out)
{
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= numTris)
    return;
  const rtflat_tri t = tris[r];
  const vec3 e0{t.e0[0], t.e0[1], t.e0[2]}, e1{t.e1[0], t.e1[1], t.e1[2]};
  for (uint32_t l = 0; l < numLights; ++l)
  {
    const rthip_dir_light& L = lights->dir[l];
    const vec3 d{-L.direction[0], -L.direction[1], -L.direction[2]}; // as the shadow kernels aim: Ray(position, -d.direction), CPUScene.cpp:34
    const vec3 s1 = rt::cross(d, e1);                                   // CPUScene.cpp:163
    const float fraction = 1.0f / rt::dot(s1, e0);                      // :166
    out[(size_t)l * numTris + r] = make_float4(s1.x, s1.y, s1.z, fraction);
  }
}

// the model-level boxes are the reference's own (Model::transform's AABB, tested with its exact arithmetic): the caller brings them
__global__ void k_refit_model_boxes(rtflat_model_node* dst, const rtflat_model_node* src, uint32_t n)
{
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n)
    return;
  for (int a = 0; a < 3; ++a)
    dst[k].lo[a] = src[k].lo[a], dst[k].hi[a] = src[k].hi[a];
}
} // namespace rtk
