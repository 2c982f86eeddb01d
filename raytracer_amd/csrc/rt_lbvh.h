// rt_lbvh.h: device-side build of the triangle-level BVH below a model leaf -- part of the gfx950 kernel
// library.  The reference has nothing here (it scans a model's triangles linearly, src/cpu/CPUScene.cpp:143,191); the host
// flattener's binned-SAH builder (src/scene/BvhFlattener.cpp) stays as the quality baseline, this is the fast path:
//
//   k_lbvh_prims     padded triangle boxes (same rule as the flattener: padAbs x sliver factor) + 63-bit Morton codes of
//                    their centres
//   rocprim          radix sort of (code, triangle) pairs
//   k_lbvh_topology  binary radix tree over the sorted codes (Karras 2012), one thread per inner node
//   k_lbvh_refit     boxes bottom-up, one launch per tree level (a node is finished once both children were finished by an
//                    EARLIER launch: kernel boundaries order the box writes, no in-kernel hand-off between CUs)
//   k_lbvh_emit      collapse to 4-wide nodes level by level from the root (the child with the largest box is opened until
//                    the node has four children or only leaves), child boxes quantised outward onto the node's 8-bit grid:
//                    the same rtflat_node records the host builder writes, in breadth-first order
//   k_lbvh_gather    triangle records into leaf (= sorted) order
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

struct LbvhNode // binary radix-tree node
{
  uint32_t left, right; // bit 31 set: a leaf = sorted triangle index in the low bits
  uint32_t first, last; // range of sorted triangles it covers (inclusive)
};

__device__ __forceinline__ uint64_t spread21(uint32_t v)
{
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// tris: the model's records in model order; boxes/codes/ids: n entries
__global__ __launch_bounds__(kBlock) void k_lbvh_prims(const rtflat_tri* tris, uint32_t n, float padAbs, float3 sceneLo, float3 sceneInvExt, LbvhBox* boxes,
                                                       uint64_t* codes, uint32_t* ids)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n)
    return;
  const rtflat_tri& t = tris[i];
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
  boxes[i] = b;
  const float cx = (0.5f * (b.lo[0] + b.hi[0]) - sceneLo.x) * sceneInvExt.x, cy = (0.5f * (b.lo[1] + b.hi[1]) - sceneLo.y) * sceneInvExt.y,
              cz = (0.5f * (b.lo[2] + b.hi[2]) - sceneLo.z) * sceneInvExt.z;
  const uint32_t qx = (uint32_t)(fminf(fmaxf(cx, 0.0f), 1.0f) * 2097151.0f), qy = (uint32_t)(fminf(fmaxf(cy, 0.0f), 1.0f) * 2097151.0f),
                 qz = (uint32_t)(fminf(fmaxf(cz, 0.0f), 1.0f) * 2097151.0f);
  codes[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
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
  nd.first = (uint32_t)first, nd.last = (uint32_t)last;
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
__global__ __launch_bounds__(kBlock) void k_lbvh_refit(const LbvhNode* nodes, uint32_t numInner, const LbvhBox* primBoxes, const uint32_t* ids, LbvhBox* innerBoxes,
                                                       uint8_t* done, uint32_t pass)
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
  const LbvhBox a = lLeaf ? primBoxes[ids[li]] : innerBoxes[li];
  const LbvhBox b = rLeaf ? primBoxes[ids[ri]] : innerBoxes[ri];
  innerBoxes[i] = lbvhUnion(a, b);
  done[i] = (uint8_t)pass;
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
  uint32_t maxLeaf;       // a subtree of at most this many triangles becomes a leaf
  const uint2* frontierIn;  // (binary inner node, its entry in `out`)
  uint2* frontierOut;
  const uint32_t* countIn;
  uint32_t* countOut;
  uint32_t* overflow;
};

struct LbvhCandidate
{
  uint32_t ref; // binary child reference (bit 31: single sorted triangle)
  LbvhBox box;
  uint32_t first, count;
};

__device__ __forceinline__ LbvhCandidate lbvhCandidate(const LbvhEmitArgs& a, uint32_t ref)
{
  LbvhCandidate c;
  c.ref = ref;
  if (ref & 0x80000000u)
  {
    const uint32_t k = ref & 0x7fffffffu;
    c.box = a.primBoxes[a.ids[k]];
    c.first = k, c.count = 1;
  }
  else
  {
    c.box = a.innerBoxes[ref];
    const LbvhNode nd = a.nodes[ref];
    c.first = nd.first, c.count = nd.last - nd.first + 1;
  }
  return c;
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
    const uint2 fr = a.frontierIn[item];
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
        if (!(cand[k].ref & 0x80000000u) && cand[k].count > a.maxLeaf)
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
    for (uint32_t k = 0; k < n; ++k)
    {
      if ((cand[k].ref & 0x80000000u) || cand[k].count <= a.maxLeaf)
        childRef[k] = rtflat_make_leaf(a.recordBase + cand[k].first, cand[k].count);
      else
      {
        const uint32_t idx = atomicAdd(a.nodeCounter, 1u);
        if (idx >= a.outCapacity)
        {
          atomicExch(a.overflow, 1u);
          childRef[k] = rtflat_make_leaf(a.recordBase + cand[k].first, 1); // keeps the arrays well-formed; the build is rejected
          continue;
        }
        childRef[k] = idx;
        a.frontierOut[atomicAdd(a.countOut, 1u)] = make_uint2(cand[k].ref, idx);
      }
    }
    rtflat_node node;
    lbvhQuantise(cand, n, childRef, node);
    a.out[fr.y] = node;
  }
}

// triangle records into leaf (= sorted) order
__global__ __launch_bounds__(kBlock) void k_lbvh_gather(const rtflat_tri* in, const uint32_t* ids, uint32_t n, rtflat_tri* out)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n)
    return;
  const float4* src = reinterpret_cast<const float4*>(in + ids[i]);
  float4* dst = reinterpret_cast<float4*>(out + i);
  dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
}
} // namespace rtk
