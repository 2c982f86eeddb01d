// rthip.hip -- hand-written gfx950 (CDNA4, wave64) kernels and the C ABI of include/rthip.h.
//
// Pipeline of one launch batch (S sample passes x P pixels = S*P paths, slot == path index, nothing moves):
//   k_trace<PRIMARY / PRIMARY_GROUP>  depth 0: ray generation fused in (src/cpu/CPURenderer.cpp:158-192; k_generate +
//                                     k_trace<CLOSEST> when fusion is switched off), closest hit
//   k_shade(0)                        hit set-up, bounce ray                 src/cpu/CPUScene.cpp:10-62
//   for depth d = 1 .. maxDepth-1
//     k_trace<SHADOW>                 any-hit toward every directional light from the hit found at depth d-1
//                                                                          src/cpu/CPUScene.cpp:31-38,107-121,139-186
//     k_trace<CLOSEST>                closest hit of the depth-d ray (side by side with the shadow launch, two streams)
//                                                                          src/cpu/CPUScene.cpp:8,123-137,187-252
//     k_shade(d)                      fold the lights of depth d-1, roulette, hit set-up, bounce ray
//   k_trace<SHADOW> (d = maxDepth)    shadow rays of the last hits, lights folded in the same launch (foldLights);
//                                     k_shade(final) does the fold when there are no directional lights
//   k_accumulate                      accumulator += radiance / N in sample order   src/cpu/CPURenderer.cpp:196-197
//   k_fold_counters                   batch tallies -> frame totals (device memory; a frame needs no host round trip)
//
// Traversal kernels (one body, five template modes) run on a persistent grid: a wave pulls chunks of slots from one of eight
// work counters, scans the chunk's liveness bytes, compacts the live slots into a per-wave LDS list with ballot + mbcnt, and
// keeps its 64 lanes busy by handing the next list entries to lanes whose ray has finished.  Each lane walks the quantised
// 4-wide triangle BVH with a short stack in LDS (slot-major, conflict free) that spills to global memory when deeper; leaves
// are parked and tested in separate wave steps.  With a depth cap of 1 two batches are in flight on two sets of buffers
// (rthip_ctx::Slot).  DESIGN.md section 5 has the measurements behind each of these choices.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "rt_device.h"

using rt::vec3;
using rtd::CameraConsts;

#define HIPCHK(x)                                                                                                                \
  do                                                                                                                            \
  {                                                                                                                             \
    hipError_t e_ = (x);                                                                                                        \
    if (e_ != hipSuccess)                                                                                                       \
      return (int)e_;                                                                                                           \
  } while (0)

namespace
{
constexpr uint32_t kBlock = 256;       // 4 waves
constexpr uint32_t kWavesPerBlock = kBlock / 64;
constexpr uint32_t kChunk = 512;       // work items a wave scans and compacts at a time
constexpr uint32_t kStackLds = 16;     // stack entries per lane kept in LDS
constexpr uint32_t kMaxStack = 96;     // deepest triangle BVH accepted (LDS part + spill)
constexpr uint32_t kLeafTrigger = 4;    // lanes that must be waiting on a leaf before a wave takes a leaf step (tuned on the 1M soup)
constexpr uint32_t kDone = 0xffffffffu; // `cur`: the ray has no node, leaf or model left (== RTFLAT_INVALID)
constexpr uint32_t kNone = 0xfffffffeu; // `cur` / `postponed`: empty
constexpr uint32_t kRefillThreshold = 16; // refill once this many lanes are idle (or no lane is active)
constexpr uint32_t kMaxRun = 1024;     // most work items one wave is handed (static mode)
constexpr uint32_t kFetchShards = 8;   // one work counter per XCD label (blockIdx % 8): 8x the atomic rate of one word, L2-local ranges
constexpr uint32_t kFetchStride = 64;  // counters 256 B apart
constexpr uint32_t kMaxTraceLaunches = 2 * (64 + 2); // two traversal launches per depth, depth <= kMaxDepthSlots
constexpr uint8_t kDead = 0xff;        // path finished, radiance in s0
constexpr uint8_t kDeadZero = 0xfe;    // path finished with exactly zero radiance (primary miss) or pixel outside the image
constexpr uint32_t kMaxLights = 8;
constexpr size_t kMaxTimedLaunches = 16384; // event pairs kept by the kernel timer between two reads
constexpr uint32_t kMaxDepthSlots = 64;

enum TraceMode
{
  CLOSEST = 0,  // closest hit of the stored ray (depth >= 1, or depth 0 when primary rays are stored)
  SHADOW = 1,   // any-hit toward every directional light from the stored origin
  ANYPROBE = 2, // any-hit of the stored ray (probe entry point)
  PRIMARY = 3,  // ray generation fused in (the primary ray is computed in registers, never stored); dynamic work fetch
  PRIMARY_GROUP = 4 // the same for tiny scenes: static runs, one 8x8 pixel block at a time, no refilling
};
constexpr bool isPrimary(int mode) { return mode == PRIMARY || mode == PRIMARY_GROUP; }
constexpr bool isClosest(int mode) { return mode == CLOSEST || isPrimary(mode); }

struct DeviceScene
{
  const rtflat_model_node* modelNodes;
  const rtflat_node* nodes;
  const rtflat_tri* tris;
  uint32_t numModelNodes;
  float cullRel, cullAbs;
};

struct Lights
{
  rthip_dir_light dir[kMaxLights];
  rthip_point_light point[kMaxLights];
  uint32_t numDir, numPoint;
};

// device-side tallies; one 64-bit word each, updated once per wave/block at kernel end
struct DeviceCounters
{
  unsigned long long extendPrimary, extendSecondary, shadow, nodesVisited, trisTested, nonFinite, nodesExtend, trisExtend;
  unsigned long long waveNodeSteps, waveLeafSteps; // work counting: node / leaf steps executed by waves (x64 = lane slots offered)
  unsigned long long hits[16];
  unsigned long long alive[kMaxDepthSlots + 2]; // paths alive at each depth of the current batch
  uint32_t listCount[kMaxDepthSlots + 2];       // sparse mode: entries of the slot list of each depth
};

struct FrameGeom
{
  uint32_t width, height, numSamples;
  uint32_t tileSize, tilesX, numLocalTiles;
  uint32_t numLocalPixels;  // P = numLocalTiles * tileSize^2
  uint32_t tileShift;       // log2(tileSize); tiles are powers of two so that pixel decoding is shifts and masks
  uint32_t sampleMagic;     // ceil(2^32 / numLocalTiles) (0 when numLocalTiles == 1): slot -> sample index without a division
  const uint32_t* tileXY;   // device, numLocalTiles entries: x0 | y0 << 16, the tile's first pixel
};

struct PathBuffers
{
  // Between shade(d) and the light fold of depth d+1 a path needs its hit position, the hit's face normal and its
  // radiance; only a path that goes on bouncing also needs a direction and its throughput ("material").  The arrays are
  // split along that line, so the last shaded depth writes 36 bytes per hit instead of 64.
  float4* rayO;  // o.xyz, w = face normal x of the hit the ray leaves from
  float4* rayD;  // d.xyz, w = material.x                        (not written by the last shaded depth)
  uint2* hit;    // x = float bits of t, y = triangle record index or RTFLAT_INVALID
  float4* s0;    // radiance.xyz, w = face normal y of the previous hit
  float2* s1;    // material.y, material.z                       (not written by the last shaded depth)
  float* nz;     // face normal z of the previous hit
  uint8_t* live; // depth at which the path is waiting, kDead / kDeadZero when finished
  uint8_t* mask; // occlusion bits of the previous hit's shadow rays
};

__device__ __forceinline__ uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t rankIn(uint64_t m)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// MI355X deals consecutive workgroups round-robin to its 8 XCDs (each with a private L2).  This bijective remap
// hands every XCD one contiguous range of work instead, so neighbouring image regions -- which walk the same BVH
// subtrees -- share an L2.  Performance only: nothing depends on where a block runs.
__device__ __forceinline__ uint32_t xcdRemap(uint32_t b, uint32_t nb)
{
  const uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// slot -> (sample index within the batch, local pixel index)
__device__ __forceinline__ void sampleOf(const FrameGeom& g, uint32_t slot, uint32_t& s, uint32_t& i)
{
  const uint32_t t = slot >> (2u * g.tileShift); // global tile slot = s * numLocalTiles + local tile
  s = g.sampleMagic ? __umulhi(t, g.sampleMagic) : t;
  i = slot - s * g.numLocalPixels;
}

// local pixel index -> image pixel; 8x8 pixel blocks are contiguous so a wave covers a compact footprint
__device__ __forceinline__ bool pixelOf(const FrameGeom& g, uint32_t i, uint32_t& w, uint32_t& h)
{
  const uint32_t sh = g.tileShift;
  const uint32_t lt = i >> (2u * sh), k = i & ((1u << (2u * sh)) - 1u);
  const uint32_t xy = g.tileXY[lt];
  const uint32_t blk = k >> 6, l = k & 63u;
  const uint32_t by = blk >> (sh - 3u), bx = blk & ((1u << (sh - 3u)) - 1u);
  w = (xy & 0xffffu) + bx * 8u + (l & 7u);
  h = (xy >> 16) + by * 8u + (l >> 3);
  return w < g.width && h < g.height;
}

// Adds the direct light of the hit a path has just left (CPUScene.cpp:31-49, lights in that order): directional lights
// unless their shadow ray was occluded, then the point lights, unshadowed.  `o` is the hit position, nPrev its face normal.
// One function for k_shade and the shadow kernel's final fold, so both produce the same bits.
__device__ __forceinline__ vec3 foldLights(const Lights* lights, uint32_t occluded, vec3 o, vec3 nPrev, vec3 radiance)
{
  const vec3 albedo{0.0f, 1.0f, 0.0f}; // CPUScene.cpp:245
  for (uint32_t l = 0; l < lights->numDir; ++l)
  {
    const rthip_dir_light& L = lights->dir[l];
    if (!((occluded >> l) & 1u))
      radiance = radiance + rtd::brdfEvaluate(albedo, nPrev, vec3{-L.direction[0], -L.direction[1], -L.direction[2]},
                                              vec3{L.color[0], L.color[1], L.color[2]});
  }
  for (uint32_t l = 0; l < lights->numPoint; ++l)
  {
    const rthip_point_light& L = lights->point[l];
    const vec3 lightDir = vec3{L.position[0], L.position[1], L.position[2]} - o;
    const float dist = rt::length(lightDir);
    const float illuminance = rt::maxRef(1 - dist / L.attenuation, 0.0f);
    radiance = radiance + illuminance * rtd::brdfEvaluate(albedo, nPrev, lightDir, vec3{L.color[0], L.color[1], L.color[2]});
  }
  return radiance;
}

// ------------------------------------------------------------------------------------------------ set-up kernels
// also clears the frame's counter totals (one dispatch less in the frame prologue)
__global__ void k_camera(rthip_camera cam, CameraConsts* out, rthip_counters* totals)
{
  if (totals && blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < sizeof(rthip_counters) / sizeof(uint64_t); i += blockDim.x)
      reinterpret_cast<uint64_t*>(totals)[i] = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *out = rtd::cameraSetup(cam);
}

// stand-alone ray generation (used when primary-ray fusion is switched off, and as the A/B baseline for it)
__global__ __launch_bounds__(kBlock) void k_generate(FrameGeom g, const CameraConsts* camp, PathBuffers pb, uint32_t sampBegin,
                                                     uint32_t numPaths)
{
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= numPaths)
    return;
  uint32_t s, i, w, h;
  sampleOf(g, p, s, i);
  if (!pixelOf(g, i, w, h))
  {
    pb.live[p] = kDeadZero;
    return;
  }
  const CameraConsts k = *camp;
  const uint32_t samp = sampBegin + s;
  const vec3 rnd = rtd::rand01(w, h, samp);
  vec3 o, d;
  rtd::primaryRay(k, g.width, g.height, w, h, samp, rnd, o, d);
  pb.rayO[p] = make_float4(o.x, o.y, o.z, 0.0f);
  pb.rayD[p] = make_float4(d.x, d.y, d.z, 0.0f);
  pb.live[p] = 0;
}

// ------------------------------------------------------------------------------------------------ traversal
struct TraceArgs
{
  DeviceScene scene;
  PathBuffers pb;
  const Lights* lights;      // SHADOW only
  FrameGeom geom;            // PRIMARY only
  const CameraConsts* cam;   // PRIMARY only
  uint32_t sampBegin;        // PRIMARY only
  uint32_t numSlots;         // dense: S*P work items, item == slot
  const uint32_t* list;      // sparse: work item -> slot; nullptr in dense mode
  const uint32_t* listCount; // sparse: number of work items (device memory)
  uint32_t runLen;           // static mode: work items per wave (multiple of 64, <= kMaxRun)
  uint32_t chunkLen;         // dynamic mode: work items per fetch (multiple of 64)
  uint32_t refillAt;         // idle lanes that trigger a refill from the work list
  uint32_t leafTrigger;      // lanes waiting on a leaf that trigger a leaf step
  uint32_t stealSpans;       // spans a wave may pull chunks from: its own first, then its neighbours'
  uint32_t* fetch;           // dynamic mode: kFetchShards counters, kFetchStride words apart, zeroed before the launch
  uint32_t depth;            // live[] value this launch serves (dense mode filter)
  uint32_t* spill;           // [spillCap][gridLanes]
  DeviceCounters* counters;
  float tmin, tmax;
  uint8_t* anyOut;           // ANYPROBE
  uint32_t foldFinal;        // SHADOW: 1 = this is the last depth: fold the lights into the radiance and finish the path here
  uint32_t aovDepths;        // SHADOW with foldFinal: shadow-flag AOV
  uint8_t* aovShadow;
};

template <int MODE, bool COUNT>
__global__ __launch_bounds__(kBlock, (MODE == 4 ? 5 : 6)) void k_trace(TraceArgs a)
{
  __shared__ uint32_t sStack[kWavesPerBlock][kStackLds][64];
  __shared__ uint32_t sList[kWavesPerBlock][kChunk];

  const uint32_t lane = laneId();
  const uint32_t wv = threadIdx.x >> 6;
  uint32_t(*stack)[64] = sStack[wv];
  uint32_t* list = sList[wv];
  const uint32_t gridLanes = gridDim.x * kBlock;
  uint32_t* spill = a.spill + (blockIdx.x * kBlock + threadIdx.x);

  const rtflat_model_node* __restrict__ modelNodes = a.scene.modelNodes;
  const float4* __restrict__ nodes4 = reinterpret_cast<const float4*>(a.scene.nodes);
  const float4* __restrict__ tris4 = reinterpret_cast<const float4*>(a.scene.tris);
  const uint32_t numModelNodes = a.scene.numModelNodes;

  // work distribution
  //  static mode (chunkLen == 0): wave `wid` owns items [wid*runLen, (wid+1)*runLen) -- used for coherent primary rays
  //    over tiny scenes, where every ray costs the same and 64-slot groups need no refilling;
  //  dynamic mode: a persistent grid; the items are split into kFetchShards contiguous spans (one per XCD label, so that
  //    an XCD's L2 keeps seeing the same image region) and a wave pulls chunkLen items at a time from its span's counter.
  const bool sparse = a.list != nullptr;
  const uint32_t total = sparse ? *a.listCount : a.numSlots;
  constexpr bool dynamic = MODE != PRIMARY_GROUP;
  const uint32_t wid = xcdRemap(blockIdx.x, gridDim.x) * kWavesPerBlock + wv;
  uint32_t runNext = dynamic ? 0u : min(total, wid * a.runLen);
  uint32_t runEnd = dynamic ? 0u : min(total, runNext + a.runLen);
  const uint32_t shard = blockIdx.x % kFetchShards;
  const uint32_t chunksPerShard = dynamic ? ((total + kFetchShards - 1) / kFetchShards + a.chunkLen - 1) / a.chunkLen : 0u;
  bool exhausted = !dynamic; // dynamic mode: no more chunks anywhere
  uint32_t shardsDone = 0;   // spans found empty so far: the wave's own first, then its neighbours' (work stealing)
  // pulls the next chunk into [runNext, runEnd): from this wave's own span while it lasts, then from the other spans
  // in turn (one attempt per call), so that an XCD whose image region was cheap helps the others out instead of idling
  // through their tail.
  auto fetchChunk = [&]() {
    const uint32_t sh = (shard + shardsDone) % kFetchShards;
    uint32_t c = 0;
    if (lane == 0)
      c = atomicAdd(a.fetch + sh * kFetchStride, 1u);
    c = __builtin_amdgcn_readfirstlane(c);
    const uint32_t base = (sh * chunksPerShard + c) * a.chunkLen;
    if (c < chunksPerShard && base < total)
    {
      runNext = base;
      runEnd = min(total, base + a.chunkLen);
    }
    else if (++shardsDone >= a.stealSpans) // this span is empty: the next call tries the next one
      exhausted = true;
  };

  // wave-uniform work-list state
  uint32_t listCount = 0, listNext = 0;

  // per-lane ray state
  bool active = false;
  bool groupValid = false; // PRIMARY: this lane carries a camera sample of the current group
  uint32_t slot = 0;
  vec3 o{0, 0, 0}, d{0, 0, 1}, invD{0, 0, 0}, oi{0, 0, 0};
  float best = FLT_MAX, cull = FLT_MAX;
  uint32_t bestRec = RTFLAT_INVALID;
  uint32_t cur = kNone;          // inner node index, a second leaf waiting for the leaf step, kDone or kNone
  uint32_t mk = 0;               // model-tree cursor (pre-order index)
  uint32_t sp = 0, spLo = 0;     // stack entries [spLo, sp) are in LDS, [0, spLo) spilled
  uint32_t postponed = 0xfffffffeu; // a leaf reference parked while the lane keeps walking inner nodes (kNone = none)
  bool needPop = false;          // the lane needs a new `cur` from its stack / the model tree
  uint32_t lightIdx = 0;         // SHADOW: which directional light this lane is testing
  uint32_t occluded = 0;         // SHADOW: result bits
  float prevNx = 0.0f;           // SHADOW: rayO.w, the face normal x of the hit the shadow rays leave from
  uint32_t nRays = 0, nNodes = 0, nTris = 0, nNodeSteps = 0, nLeafSteps = 0;

  // SHADOW: every light of this lane's path has been tested.  Normally the occlusion bits go to mask[] and the next
  // shade launch folds the lights; at the last depth the fold happens right here (same function, same bits), which
  // saves that launch and its re-read of the path state.
  auto finishShadow = [&]() {
    if (a.foldFinal)
    {
      const float4 s0 = a.pb.s0[slot];
      const vec3 radiance = foldLights(a.lights, occluded, o, vec3{prevNx, s0.w, a.pb.nz[slot]}, vec3{s0.x, s0.y, s0.z});
      if (a.depth - 1 < a.aovDepths)
      {
        uint32_t sidx, i;
        sampleOf(a.geom, slot, sidx, i);
        a.aovShadow[((size_t)sidx * a.aovDepths + (a.depth - 1)) * a.geom.numLocalPixels + i] = (uint8_t)occluded;
      }
      a.pb.s0[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
      a.pb.live[slot] = kDead;
    }
    else
      a.pb.mask[slot] = (uint8_t)occluded;
  };

  for (;;)
  {
    if (MODE == PRIMARY_GROUP)
    {
      // ---------------- coherent primary rays over a tiny scene: take the run 64 slots (one 8x8 pixel block) at a time,
      // generate the rays in registers at full occupancy, traverse until the whole group is done, store the 64 hit
      // records together
      if (__ballot(active) == 0)
      {
        // a primary miss contributes exactly +0: its slot is marked finished here and costs the shade kernel nothing
        if (groupValid)
        {
          const bool hitSomething = bestRec != RTFLAT_INVALID;
          if (hitSomething)
            a.pb.hit[slot] = make_uint2(__float_as_uint(best), bestRec);
          a.pb.live[slot] = hitSomething ? (uint8_t)0 : kDeadZero;
        }
        groupValid = false;
        if (runNext >= runEnd)
          break;
        slot = runNext + lane;
        runNext += 64;
        uint32_t sidx, i, w, h;
        sampleOf(a.geom, slot, sidx, i);
        const bool inImage = pixelOf(a.geom, i, w, h);
        if (slot < runEnd && !inImage)
          a.pb.live[slot] = kDeadZero; // padding pixel of a tile that overhangs the image
        if (slot < runEnd && inImage)
        {
          const uint32_t samp = a.sampBegin + sidx;
          rtd::primaryRay(*a.cam, a.geom.width, a.geom.height, w, h, samp, rtd::rand01(w, h, samp), o, d);
          invD = rtd::padInvDir(d);
          oi = o * invD;
          best = FLT_MAX;
          cull = FLT_MAX;
          bestRec = RTFLAT_INVALID;
          cur = kNone;
          postponed = kNone;
          needPop = true;
          mk = 0;
          sp = spLo = 0;
          active = true;
          groupValid = true;
          nRays++;
        }
        continue;
      }
    }
    else if (MODE == PRIMARY)
    {
      // ---------------- primary rays over a real scene: neighbouring pixels still walk very different numbers of nodes,
      // so idle lanes are refilled with the next consecutive slots of the chunk once half the wave is idle
      const uint64_t idleMask = __ballot(!active);
      const uint32_t numIdle = __popcll(idleMask);
      if (numIdle >= 32)
      {
        if (runNext >= runEnd && !exhausted)
          fetchChunk();
        if (runNext < runEnd)
        {
          const uint32_t r = rankIn(idleMask);
          const uint32_t avail = runEnd - runNext;
          if (!active && r < avail)
          {
            slot = runNext + r;
            uint32_t sidx, i, w, h;
            sampleOf(a.geom, slot, sidx, i);
            if (!pixelOf(a.geom, i, w, h))
              a.pb.live[slot] = kDeadZero; // padding pixel of a tile that overhangs the image
            else
            {
              const uint32_t samp = a.sampBegin + sidx;
              rtd::primaryRay(*a.cam, a.geom.width, a.geom.height, w, h, samp, rtd::rand01(w, h, samp), o, d);
              invD = rtd::padInvDir(d);
              oi = o * invD;
              best = FLT_MAX;
              cull = FLT_MAX;
              bestRec = RTFLAT_INVALID;
              cur = kNone;
              postponed = kNone;
              needPop = true;
              mk = 0;
              sp = spLo = 0;
              active = true;
              nRays++;
            }
          }
          runNext += min(numIdle, avail);
        }
      }
      if (__ballot(active) == 0)
      {
        if (exhausted && runNext >= runEnd)
          break;
        continue;
      }
    }
    else
    {
      // ---------------- work list: scan + compact the next chunk of the run when the list has run dry
      const uint64_t idleMask = __ballot(!active);
      const uint32_t numIdle = __popcll(idleMask);
      if (listNext >= listCount && numIdle >= a.refillAt && runNext >= runEnd && !exhausted)
        fetchChunk();
      if (listNext >= listCount && runNext < runEnd && numIdle >= a.refillAt)
      {
        const uint32_t chunkEnd = min(runEnd, runNext + kChunk);
        uint32_t n = 0;
        for (uint32_t base = runNext; base < chunkEnd; base += 64)
        {
          const uint32_t item = base + lane;
          bool isLive = item < chunkEnd;
          uint32_t s = item;
          if (sparse)
          {
            if (isLive)
              s = a.list[item];
          }
          else
            isLive = isLive && (a.pb.live[item] == (uint8_t)a.depth);
          const uint64_t m = __ballot(isLive);
          if (isLive)
            list[n + rankIn(m)] = s;
          n += __popcll(m);
        }
        runNext = chunkEnd;
        listCount = n;
        listNext = 0;
        // the list is written and read by this wave only; LDS operations of one wave complete in order
      }
      // ---------------- hand list entries to idle lanes (ballot + mbcnt compaction)
      const uint32_t avail = listCount - listNext;
      if (avail > 0 && (numIdle >= a.refillAt || numIdle == 64))
      {
        const uint32_t r = rankIn(idleMask);
        const bool take = !active && r < avail;
        if (take)
        {
          slot = list[listNext + r];
          const float4 ro = a.pb.rayO[slot];
          o = vec3{ro.x, ro.y, ro.z};
          if (MODE == SHADOW)
          {
            lightIdx = 0;
            occluded = 0;
            prevNx = ro.w;
            const rthip_dir_light& L = a.lights->dir[0];
            d = vec3{-L.direction[0], -L.direction[1], -L.direction[2]}; // Ray(position, -d.direction), CPUScene.cpp:34
          }
          else
          {
            const float4 rd = a.pb.rayD[slot];
            d = vec3{rd.x, rd.y, rd.z};
          }
          invD = rtd::padInvDir(d);
          oi = o * invD;
          best = FLT_MAX;
          cull = FLT_MAX;
          bestRec = RTFLAT_INVALID;
          cur = kNone;
          postponed = kNone;
          needPop = true;
          mk = 0;
          sp = spLo = 0;
          active = true;
          nRays++;
        }
        listNext += min(numIdle, avail);
      }
      if (__ballot(active) == 0)
      {
        if (exhausted && runNext >= runEnd && listNext >= listCount)
          break;
        continue;
      }
    }

    const bool refillPossible = (MODE == PRIMARY_GROUP) ? false
                                : (MODE == PRIMARY)     ? (runNext < runEnd || !exhausted)
                                                        : (listNext < listCount || runNext < runEnd || !exhausted);
    const uint32_t refillAt = (MODE == PRIMARY) ? 32u : a.refillAt;
    // ---------------- traversal: a wave alternates NODE steps (every lane that holds an inner node tests its two child
    // boxes) with LEAF steps (every lane that holds a postponed leaf runs the reference's triangle test on it).  A lane
    // that reaches a leaf parks it in `postponed` and keeps walking inner nodes; the leaf step is taken only once
    // kLeafTrigger lanes cannot go on without it (they met a second leaf, or have nothing else left) or no lane can do a
    // node step.  Compared with a plain while-while loop, where every lane waits at its leaf until the slowest lane has
    // found one, this keeps most lanes busy in both kinds of step.
    for (;;)
    {
      // --- resolve: lanes that need a new `cur` pop their stack, or walk on to the next model leaf when it is empty
      while (active && needPop)
      {
        uint32_t got;
        if (sp != 0)
        {
          if (sp == spLo)
          {
            spLo--;
            stack[spLo % kStackLds][lane] = spill[(size_t)spLo * gridLanes];
          }
          sp--;
          got = stack[sp % kStackLds][lane];
        }
        else
        {
          // stack empty: next model leaf whose box chain passes the reference's test, in DFS order
          // (CPUScene.cpp:123-137 / 107-121: both children always, box test with the caller's tmin/tmax)
          got = kDone;
          while (mk < numModelNodes)
          {
            const rtflat_model_node* mn = modelNodes + mk;
            if (COUNT)
              nNodes++;
            // AABB.h:49 divides by the direction: invD is that quotient unless it was clamped (axis-parallel rays, rare),
            // so the three IEEE divisions are repeated only for a wave that holds such a ray
            vec3 invRef = invD;
            if (fabsf(invD.x) == 1e18f || fabsf(invD.y) == 1e18f || fabsf(invD.z) == 1e18f)
              invRef = rt::rcp3(d);
            if (!rtd::refSlab(mn->lo, mn->hi, o, invRef, a.tmin, a.tmax))
            {
              mk += mn->subtreeSize; // skip the subtree
              continue;
            }
            mk++;
            if (mn->triCount > 0)
            {
              got = mn->blas;
              break;
            }
          }
        }
        if (got != kDone && (int)got < 0 && postponed == kNone)
          postponed = got; // a leaf: park it and keep looking for inner-node work
        else
        {
          cur = got;
          needPop = false;
        }
      }
      // --- rays that have nothing left
      if (active && cur == kDone && postponed == kNone)
      {
        bool done = true;
        if (MODE == SHADOW)
        {
          // not occluded for this light; the next light restarts the walk
          lightIdx++;
          if (lightIdx < a.lights->numDir)
          {
            const rthip_dir_light& L = a.lights->dir[lightIdx];
            d = vec3{-L.direction[0], -L.direction[1], -L.direction[2]};
            invD = rtd::padInvDir(d);
            oi = o * invD;
            mk = 0;
            needPop = true;
            cur = kNone;
            nRays++;
            done = false;
          }
          else
            finishShadow();
        }
        else if (MODE == CLOSEST)
          a.pb.hit[slot] = make_uint2(__float_as_uint(best), bestRec);
        else if (MODE == PRIMARY)
        {
          // a primary miss contributes exactly +0: its slot is marked finished here and costs the shade kernel nothing
          const bool hitSomething = bestRec != RTFLAT_INVALID;
          if (hitSomething)
            a.pb.hit[slot] = make_uint2(__float_as_uint(best), bestRec);
          a.pb.live[slot] = hitSomething ? (uint8_t)0 : kDeadZero;
        }
        else if (MODE == ANYPROBE)
          a.anyOut[slot] = 0;
        // PRIMARY_GROUP: the group's hit records are stored together once every lane has finished
        if (done)
          active = false;
      }
      const bool atNode = active && !needPop && (int)cur >= 0;
      const bool stalled = active && !needPop && postponed != kNone && !((int)cur >= 0);
      const uint64_t nodeMask = __ballot(atNode || (active && needPop));
      const uint32_t numStalled = __popcll(__ballot(stalled));
      const bool leafStep = numStalled >= a.leafTrigger || (nodeMask == 0 && numStalled > 0);
      if (leafStep)
      {
        if (COUNT)
          nLeafSteps++;
        // --- LEAF step: the reference's triangle test on every record of every parked leaf
        if (active && postponed != kNone)
        {
          const uint32_t first = rtflat_leaf_first(postponed), count = rtflat_leaf_count(postponed);
          postponed = kNone;
          bool anyAccepted = false;
          for (uint32_t k = 0; k < count; ++k)
          {
            const float4* tp = tris4 + (size_t)(first + k) * 4;
            const float4 a0 = tp[0], a1 = tp[1], a2 = tp[2];
            if (COUNT)
              nTris++;
            float t;
            const int acc = rtd::triangleTest(vec3{a0.x, a0.y, a0.z}, vec3{a0.w, a1.x, a1.y}, vec3{a1.z, a1.w, a2.x}, o, d, a.tmin, a.tmax, t);
            if (acc == 2)
              atomicAdd(&a.counters->nonFinite, 1ull);
            if (acc != 1)
              continue;
            if (isClosest(MODE))
            {
              bool better = t < best;
              if (!better && t == best)
              {
                // tie: the triangle later in (DFS model order, index in model) wins (CPUScene.cpp:231, :136)
                const uint32_t keyNew = a.scene.tris[first + k].orderKey, keyOld = a.scene.tris[bestRec].orderKey;
                better = keyNew > keyOld;
              }
              if (better)
              {
                best = t;
                bestRec = first + k;
                cull = rt::fma(best, 1.0f + a.scene.cullRel, a.scene.cullAbs);
              }
            }
            else
            {
              anyAccepted = true;
              break;
            }
          }
          if (!isClosest(MODE) && anyAccepted)
          {
            // first accepted triangle ends an any-hit ray (testModel returns true, CPUScene.cpp:182)
            if (MODE == SHADOW)
            {
              occluded |= 1u << lightIdx;
              lightIdx++;
              if (lightIdx < a.lights->numDir)
              {
                const rthip_dir_light& L = a.lights->dir[lightIdx];
                d = vec3{-L.direction[0], -L.direction[1], -L.direction[2]};
                invD = rtd::padInvDir(d);
                oi = o * invD;
                mk = 0;
                sp = spLo = 0;
                cur = kNone;
                needPop = true;
                nRays++;
              }
              else
              {
                finishShadow();
                active = false;
              }
            }
            else
            {
              a.anyOut[slot] = 1;
              active = false;
            }
          }
          else if (!needPop && cur != kDone && (int)cur < 0)
          {
            // the lane was holding a second leaf: park it now and look for more work
            postponed = cur;
            cur = kNone;
            needPop = true;
          }
        }
        continue;
      }
      if (nodeMask == 0)
        break; // nothing left to traverse in this wave: back to the work list
      // leave for a refill as soon as enough lanes are idle and there is work to hand out
      if (refillPossible && (uint32_t)__popcll(__ballot(!active)) >= refillAt)
        break;
      // --- NODE step: one 64-byte node = four quantised child boxes
      if (COUNT)
        nNodeSteps++;
      if (atNode)
      {
        const float4* np = nodes4 + (size_t)cur * 4;
        const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
        if (COUNT)
          nNodes++;
        const rtd::WideNode wn = rtd::wideNodeSetup(w0, w2, w3, invD, oi);
        float en0, ex0, en1, ex1, en2, ex2, en3, ex3;
        rtd::wideSlabPair<0>(wn, en0, ex0, en1, ex1);
        rtd::wideSlabPair<2>(wn, en2, ex2, en3, ex3);
        constexpr bool kClosest = isClosest(MODE);
        // sort the (at most four) children that were hit by entry distance; unused slots have inverted boxes
        uint32_t k0 = ((en0 <= ex0) && (!kClosest || en0 <= cull)) ? __float_as_uint(en0) : 0xffffffffu;
        uint32_t k1 = ((en1 <= ex1) && (!kClosest || en1 <= cull)) ? __float_as_uint(en1) : 0xffffffffu;
        uint32_t k2 = ((en2 <= ex2) && (!kClosest || en2 <= cull)) ? __float_as_uint(en2) : 0xffffffffu;
        uint32_t k3 = ((en3 <= ex3) && (!kClosest || en3 <= cull)) ? __float_as_uint(en3) : 0xffffffffu;
        uint32_t r0 = __float_as_uint(w1.x), r1 = __float_as_uint(w1.y), r2 = __float_as_uint(w1.z), r3 = __float_as_uint(w1.w);
        rtd::orderPair(k0, r0, k1, r1);
        rtd::orderPair(k2, r2, k3, r3);
        rtd::orderPair(k0, r0, k2, r2);
        rtd::orderPair(k1, r1, k3, r3);
        rtd::orderPair(k1, r1, k2, r2);
        // push the farther ones, farthest first, so that the nearest of them is popped first.  Misses sort to the
        // end, so the valid flags are monotone (v1 >= v2 >= v3) and the three references can be written without
        // branches: r3 at sp, r2 at sp+v3, r1 at sp+v3+v2 -- a reference that is not valid is overwritten by the next
        // one or lands above the new top.  One spill check per step makes room for all three.
        if (sp + 3 - spLo > kStackLds)
        {
          do
          {
            spill[(size_t)spLo * gridLanes] = stack[spLo % kStackLds][lane];
            spLo++;
          } while (sp + 3 - spLo > kStackLds);
        }
        const uint32_t v3 = k3 != 0xffffffffu, v2 = k2 != 0xffffffffu, v1 = k1 != 0xffffffffu;
        stack[sp % kStackLds][lane] = r3;
        stack[(sp + v3) % kStackLds][lane] = r2;
        stack[(sp + v3 + v2) % kStackLds][lane] = r1;
        sp += v3 + v2 + v1;
        if (k0 == 0xffffffffu)
          needPop = true;
        else if ((int)r0 < 0 && postponed == kNone)
        {
          postponed = r0; // a leaf: park it, go on with the stack
          needPop = true;
        }
        else
          cur = r0; // an inner node, or a second leaf (the lane now waits for the leaf step)
        if (needPop)
          cur = kNone;
      }
    }
    // the wave is out of traversal work (or wants a refill): PRIMARY_GROUP and the list modes pick up from here
  }

  // ---------------- tallies: one atomic per wave and counter
  for (int off = 32; off > 0; off >>= 1)
  {
    nRays += __shfl_down(nRays, off);
    if (COUNT)
    {
      nNodes += __shfl_down(nNodes, off);
      nTris += __shfl_down(nTris, off);
    }
  }
  if (lane == 0)
  {
    if (nRays)
    {
      unsigned long long* dst =
          (MODE == SHADOW) ? &a.counters->shadow : ((isPrimary(MODE) || a.depth == 0) ? &a.counters->extendPrimary : &a.counters->extendSecondary);
      atomicAdd(dst, (unsigned long long)nRays);
    }
    if (COUNT)
    {
      atomicAdd(&a.counters->waveNodeSteps, (unsigned long long)nNodeSteps);
      atomicAdd(&a.counters->waveLeafSteps, (unsigned long long)nLeafSteps);
    }
    if (COUNT && nNodes)
    {
      atomicAdd(&a.counters->nodesVisited, (unsigned long long)nNodes);
      if (isClosest(MODE))
        atomicAdd(&a.counters->nodesExtend, (unsigned long long)nNodes);
    }
    if (COUNT && nTris)
    {
      atomicAdd(&a.counters->trisTested, (unsigned long long)nTris);
      if (isClosest(MODE))
        atomicAdd(&a.counters->trisExtend, (unsigned long long)nTris);
    }
  }
}

// ------------------------------------------------------------------------------------------------ shading
struct ShadeArgs
{
  FrameGeom g;
  DeviceScene scene;
  PathBuffers pb;
  const Lights* lights;
  const CameraConsts* cam;
  DeviceCounters* counters;
  uint32_t numSlots, depth, maxDepth, sampBegin;
  uint32_t finalFold;      // 1: only fold the pending lights and finish every path (depth == maxDepth)
  uint32_t primaryInRegs;  // 1: depth-0 rays were never stored; recompute them from (pixel, sample)
  uint32_t aovDepths;
  uint32_t* aovTri;        // [S][aovDepths][P]
  uint8_t* aovShadow;      // [S][aovDepths][P]
  const uint32_t* list;    // sparse mode: slots alive at `depth` (nullptr = dense)
  uint32_t* listNext;      // sparse mode: survivors are appended here (count in counters->listCount[depth+1])
};

template <bool SPARSE>
__global__ __launch_bounds__(kBlock) void k_shade(ShadeArgs a)
{
  __shared__ uint32_t sAlive[kWavesPerBlock], sHits[kWavesPerBlock];
  uint32_t nAlive = 0, nHits = 0;
  const uint8_t depthTag = (uint8_t)a.depth;
  const vec3 albedo{0.0f, 1.0f, 0.0f};    // CPUScene.cpp:245
  const vec3 emissive{0.0f, 0.0f, 0.0f};  // CPUScene.cpp:246
  const uint32_t total = SPARSE ? a.counters->listCount[a.depth] : a.numSlots;
  // every wave runs the same number of iterations so that the sparse append can use wave-wide ballots
  const uint32_t stride = gridDim.x * kBlock;
  const uint32_t iters = (total + stride - 1) / stride;
  for (uint32_t it = 0; it < iters; ++it)
  {
    const uint32_t item = it * stride + blockIdx.x * kBlock + threadIdx.x;
    bool mine = item < total;
    uint32_t p = item;
    if (SPARSE)
    {
      if (mine)
        p = a.list[item];
    }
    else if (mine)
      mine = a.pb.live[p] == depthTag; // (the fused primary kernels write the depth-0 bytes: 0 = hit, kDeadZero = miss)
    bool survives = false;
    if (mine)
    {
      uint32_t s, i;
      sampleOf(a.g, p, s, i);
      uint32_t w = 0, h = 0;
      const bool regen = a.depth == 0 && a.primaryInRegs;
      const bool folding = a.finalFold != 0;
      vec3 o{0, 0, 0}, d{0, 0, 1}, rnd{0, 0, 0};
      vec3 radiance{0.0f, 0.0f, 0.0f}, material{1.0f, 1.0f, 1.0f}; // Ray.h:7-8
      if (!regen)
      {
        const float4 ro = a.pb.rayO[p];
        o = vec3{ro.x, ro.y, ro.z};
        if (!folding)
        {
          const float4 rd = a.pb.rayD[p];
          d = vec3{rd.x, rd.y, rd.z};
          if (a.depth > 0)
            material.x = rd.w;
        }
        if (a.depth > 0)
        {
          const float4 s0 = a.pb.s0[p];
          radiance = vec3{s0.x, s0.y, s0.z};
          if (!folding)
          {
            const float2 s1 = a.pb.s1[p];
            material.y = s1.x, material.z = s1.y;
          }
          // lights of the previous hit, whose position is this ray's origin (CPUScene.cpp:31-49)
          const uint32_t occl = a.lights->numDir ? a.pb.mask[p] : 0u;
          radiance = foldLights(a.lights, occl, o, vec3{ro.w, s0.w, a.pb.nz[p]}, radiance);
          if (a.depth - 1 < a.aovDepths)
            a.aovShadow[((size_t)s * a.aovDepths + (a.depth - 1)) * a.g.numLocalPixels + i] = (uint8_t)occl;
        }
      }
      bool finished = folding;
      uint32_t rec = RTFLAT_INVALID;
      float t = FLT_MAX;
      if (!finished)
      {
        const uint2 hv = a.pb.hit[p];
        t = __uint_as_float(hv.x);
        rec = hv.y;
        if (!(t < FLT_MAX)) // CPUScene.cpp:10
          finished = true;
        else
        {
          nHits++;
          pixelOf(a.g, i, w, h);
          rnd = rtd::rand01(w, h, a.sampBegin + s);
          if (regen) // the primary ray was never stored: same function, same bits (only paths that hit need it)
            rtd::primaryRay(*a.cam, a.g.width, a.g.height, w, h, a.sampBegin + s, rnd, o, d);
          const float pr = rt::maxRef(rt::maxRef(albedo.x, albedo.y), albedo.z); // :13
          if (a.depth > 5)                                                      // :19 (depth >= maxDepth never reaches here)
          {
            if (rnd.z >= pr)
              finished = true;
            else
              material = material / pr;
          }
        }
      }
      if (finished)
      {
        if (a.depth == 0)
          a.pb.live[p] = kDeadZero; // a primary miss adds exactly +0 to the accumulator: nothing to store
        else
        {
          a.pb.s0[p] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
          a.pb.live[p] = kDead;
        }
      }
      else
      {
        const float4* tp = reinterpret_cast<const float4*>(a.scene.tris + rec);
        const float4 a2 = tp[2], a3 = tp[3];
        const vec3 n{a2.y, a2.z, a2.w};
        if (a.depth < a.aovDepths)
          a.aovTri[((size_t)s * a.aovDepths + a.depth) * a.g.numLocalPixels + i] = __float_as_uint(a3.x);
        const float payloadEmissive = a.depth == 0 ? 1.0f : 0.0f;                  // Ray.h:10, CPUScene.cpp:61
        radiance = radiance + material * emissive * payloadEmissive;              // :27
        material = material * albedo;                                             // :28
        const vec3 position = o + d * t;                                          // :238
        a.pb.rayO[p] = make_float4(position.x, position.y, position.z, n.x);
        a.pb.s0[p] = make_float4(radiance.x, radiance.y, radiance.z, n.y);
        a.pb.nz[p] = n.z;
        if (a.depth + 1 < a.maxDepth) // the last shaded depth traces no bounce ray: neither direction nor throughput is kept
        {
          const vec3 normalLight = rt::dot(n, d) < 0 ? n : -n;                    // :240
          const vec3 nd = rtd::bounceDirection(normalLight, rnd);                 // :54-60
          a.pb.rayD[p] = make_float4(nd.x, nd.y, nd.z, material.x);
          a.pb.s1[p] = make_float2(material.y, material.z);
        }
        a.pb.live[p] = (uint8_t)(a.depth + 1);
        nAlive++;
        survives = true;
      }
    }
    if (SPARSE)
    {
      // append the survivors to the next depth's slot list: one atomic per wave, ranks from the ballot
      const uint64_t m = __ballot(survives);
      if (m)
      {
        uint32_t base = 0;
        if (laneId() == (uint32_t)__ffsll((long long)m) - 1)
          base = atomicAdd(&a.counters->listCount[a.depth + 1], (uint32_t)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (survives)
          a.listNext[base + rankIn(m)] = p;
      }
    }
  }
  // block tallies: one atomic per block and counter
  for (int off = 32; off > 0; off >>= 1)
  {
    nAlive += __shfl_down(nAlive, off);
    nHits += __shfl_down(nHits, off);
  }
  const uint32_t wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0)
  {
    sAlive[wv] = nAlive;
    sHits[wv] = nHits;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t ta = 0, th = 0;
    for (uint32_t k = 0; k < kWavesPerBlock; ++k)
    {
      ta += sAlive[k];
      th += sHits[k];
    }
    if (ta)
      atomicAdd(&a.counters->alive[a.depth + 1], (unsigned long long)ta);
    if (th)
      atomicAdd(&a.counters->hits[a.depth < 15 ? a.depth : 15], (unsigned long long)th);
  }
}

// dense -> sparse hand-over: collects the slots alive at `depth` into a list (order: ascending slot within a
// block's 4096-slot span, spans in arrival order).  One atomic per block.
__global__ __launch_bounds__(kBlock) void k_compact(PathBuffers pb, uint32_t numSlots, uint32_t depth, uint32_t* list, uint32_t capacity,
                                                    DeviceCounters* counters)
{
  __shared__ uint32_t sWave[kWavesPerBlock];
  __shared__ uint32_t sBase;
  const uint32_t lane = laneId(), wv = threadIdx.x >> 6;
  const uint32_t spanBase = blockIdx.x * (kBlock * 16) + threadIdx.x * 16;
  uint32_t bits = 0, cnt = 0;
  if (spanBase < numSlots) // numSlots is a multiple of 64, the buffer is padded: the 16-byte load stays in bounds
  {
    const uint4 v = *reinterpret_cast<const uint4*>(pb.live + spanBase);
    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (spanBase + k < numSlots && ((wds[k >> 2] >> ((k & 3) * 8)) & 0xffu) == depth)
      {
        bits |= 1u << k;
        cnt++;
      }
  }
  // exclusive prefix of cnt over the wave, then over the block
  uint32_t incl = cnt;
  for (int off = 1; off < 64; off <<= 1)
  {
    const uint32_t up = __shfl_up(incl, off);
    if ((int)lane >= off)
      incl += up;
  }
  if (lane == 63)
    sWave[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t tot = 0;
    for (uint32_t k = 0; k < kWavesPerBlock; ++k)
    {
      const uint32_t c = sWave[k];
      sWave[k] = tot;
      tot += c;
    }
    sBase = tot ? atomicAdd(&counters->listCount[depth], tot) : 0u;
  }
  __syncthreads();
  uint32_t dst = sBase + sWave[wv] + (incl - cnt);
  for (int k = 0; k < 16; ++k)
    if ((bits >> k) & 1u)
    {
      if (dst < capacity)
        list[dst] = spanBase + k;
      dst++;
    }
}

// accumulator[pix] += radiance / float(N), sample passes in ascending order (CPURenderer.cpp:197)
// `fresh` = first batch of a frame: the accumulator is taken as zero instead of being loaded (saves clearing it beforehand)
__global__ __launch_bounds__(kBlock) void k_accumulate(FrameGeom g, PathBuffers pb, float* acc, uint32_t numBatchSamples, uint32_t fresh)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= g.numLocalPixels)
    return;
  float r = 0, gg = 0, b = 0;
  bool loaded = fresh != 0; // 0 + x == x and 0 + 0 == +0: starting from zero is what the cleared accumulator would give
  const float n = float(g.numSamples);
  for (uint32_t s = 0; s < numBatchSamples; ++s)
  {
    const size_t p = (size_t)s * g.numLocalPixels + i;
    // kDeadZero: the sample's radiance is exactly +0 (primary miss) -> adding 0/N leaves the sum unchanged
    if (pb.live[p] != kDead)
      continue;
    if (!loaded)
    {
      r = acc[3 * (size_t)i], gg = acc[3 * (size_t)i + 1], b = acc[3 * (size_t)i + 2];
      loaded = true;
    }
    const float4 v = pb.s0[p];
    r += v.x / n;
    gg += v.y / n;
    b += v.z / n;
  }
  if (loaded)
  {
    acc[3 * (size_t)i] = r;
    acc[3 * (size_t)i + 1] = gg;
    acc[3 * (size_t)i + 2] = b;
  }
}

// image = pow(max(accumulator * resolver, 0), 0.45)  (CPURenderer.cpp:196,198); display only
__global__ __launch_bounds__(kBlock) void k_tonemap(const float* acc, float* img, uint32_t numFloats, float resolver)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < numFloats)
    img[i] = powf(rt::maxRef(acc[i] * resolver, 0.0f), 0.45f);
}

// tile-major -> row-major scatter (multi-GPU assembly and read-back)
__global__ __launch_bounds__(kBlock) void k_untile(FrameGeom g, const float* tileMajor, float* rowMajor, uint32_t numPixels)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= numPixels)
    return;
  uint32_t w, h;
  if (!pixelOf(g, i, w, h))
    return;
  const size_t dst = 3 * ((size_t)h * g.width + w);
  rowMajor[dst] = tileMajor[3 * (size_t)i];
  rowMajor[dst + 1] = tileMajor[3 * (size_t)i + 1];
  rowMajor[dst + 2] = tileMajor[3 * (size_t)i + 2];
}

// ------------------------------------------------------------------------------------------------ probes
// adds one batch's tallies to the frame totals (device memory), so a frame needs no host round trip between batches
__global__ void k_fold_counters(const DeviceCounters* batch, rthip_counters* totals)
{
  if (threadIdx.x != 0 || blockIdx.x != 0)
    return;
  // atomics: two batches in flight on two streams may fold at the same time
  auto add = [](uint64_t* dst, unsigned long long v) {
    if (v)
      atomicAdd(reinterpret_cast<unsigned long long*>(dst), v);
  };
  add(&totals->extendPrimary, batch->extendPrimary);
  add(&totals->extendSecondary, batch->extendSecondary);
  add(&totals->shadow, batch->shadow);
  add(&totals->nodesVisited, batch->nodesVisited);
  add(&totals->trisTested, batch->trisTested);
  add(&totals->nonFiniteAccepts, batch->nonFinite);
  add(&totals->nodesVisitedExtend, batch->nodesExtend);
  add(&totals->trisTestedExtend, batch->trisExtend);
  add(&totals->waveNodeSteps, batch->waveNodeSteps);
  add(&totals->waveLeafSteps, batch->waveLeafSteps);
  for (int i = 0; i < 16; ++i)
    add(&totals->hits[i], batch->hits[i]);
}

__global__ void k_probe_rand01(const uint32_t* xyz, uint32_t n, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const vec3 r = rtd::rand01(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
}
__global__ void k_probe_sincos(const float* x, uint32_t n, float* s, float* c)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    rtd::sincosDet(x[i], s[i], c[i]);
}
__global__ void k_probe_rays(const CameraConsts* camp, uint32_t width, uint32_t height, const uint32_t* whs, uint32_t n, float* o3, float* d3)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const CameraConsts k = *camp;
  const uint32_t w = whs[3 * i], h = whs[3 * i + 1], s = whs[3 * i + 2];
  vec3 o, d;
  rtd::primaryRay(k, width, height, w, h, s, rtd::rand01(w, h, s), o, d);
  o3[3 * i] = o.x, o3[3 * i + 1] = o.y, o3[3 * i + 2] = o.z;
  d3[3 * i] = d.x, d3[3 * i + 1] = d.y, d3[3 * i + 2] = d.z;
}
__global__ void k_probe_bounce(const float* n3, const float* r3, uint32_t n, float* d3)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const vec3 d = rtd::bounceDirection(vec3{n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]}, vec3{r3[3 * i], r3[3 * i + 1], r3[3 * i + 2]});
  d3[3 * i] = d.x, d3[3 * i + 1] = d.y, d3[3 * i + 2] = d.z;
}
__global__ void k_probe_pack(const float* o3, const float* d3, uint32_t n, float4* ro, float4* rd, uint8_t* live)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  ro[i] = make_float4(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2], 0.0f);
  rd[i] = make_float4(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2], 0.0f);
  live[i] = 0;
}
__global__ void k_probe_unpack(const DeviceScene scene, const uint2* hit, uint32_t n, uint32_t* triId, float* t)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const uint2 h = hit[i];
  t[i] = __uint_as_float(h.x);
  triId[i] = (h.y == RTFLAT_INVALID) ? RTFLAT_INVALID : scene.tris[h.y].triId;
}

template <typename T>
struct DevBuf
{
  T* p = nullptr;
  size_t n = 0;
  int ensure(size_t count)
  {
    if (count <= n && p)
      return 0;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
    HIPCHK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
    n = count;
    return 0;
  }
  void release()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};
} // namespace

struct rthip_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr; // second stream for the shadow launch of a depth (see renderBatch)
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  hipEvent_t evA = nullptr, evB = nullptr, evK0 = nullptr, evK1 = nullptr;
  int numCUs = 256;
  uint32_t traceBlocks = 2048; // persistent grid of the dynamic mode
  uint32_t fetchCursor = 0;    // next unused counter set of the current batch
  // scene
  bool haveScene = false;
  rtflat_info info{};
  DevBuf<rtflat_model_node> modelNodes;
  DevBuf<rtflat_node> nodes;
  DevBuf<rtflat_tri> tris;
  DevBuf<Lights> lights;
  Lights hostLights{};
  // frame
  bool haveFrame = false;
  rthip_camera camera{};
  FrameGeom geom{};
  uint32_t maxDepth = 12, aovDepths = 0;
  uint32_t batchCap = 1; // sample passes per launch batch
  std::vector<uint32_t> hostTileIds, hostTileXY;
  DevBuf<uint32_t> tileIds;
  DevBuf<CameraConsts> camConsts;
  DevBuf<float4> rayO, rayD, s0;
  DevBuf<float2> s1;
  DevBuf<float> nz;
  DevBuf<uint2> hit;
  DevBuf<uint8_t> live, mask, anyOut;
  DevBuf<float> acc, img, rowScratch;
  DevBuf<uint32_t> aovTri;
  DevBuf<uint8_t> aovShadow;
  DevBuf<uint32_t> spill;
  DevBuf<uint32_t> fetch;        // [batch counters (kCounterWords) | work-fetch counters]: everything a batch zeroes, one memset
  DevBuf<uint32_t> spill2;       // stack spill area of launches on stream2
  DevBuf<uint32_t> listA, listB; // sparse-mode slot lists (even / odd depth)
  uint32_t spillCap = 1;         // per-lane spill entries the uploaded BVH can need
  uint32_t sparseCap = 0;        // capacity of listA / listB
  uint32_t tuneBlocks = 6;
  uint32_t overlapShadow = 1;    // run a depth's shadow and closest-hit launches on two streams (their tails overlap)
  uint32_t tuneSteal = 1; // spans a wave may pull chunks from (1 = its own only; stealing measured +4% on the 1M soup, -4% on the bunny: off)
  uint32_t tuneRefill = kRefillThreshold, tuneLeaf = kLeafTrigger;
  bool fusedPrimary = true;      // generate rays inside the depth-0 traversal kernel instead of storing them
  uint32_t lastBatchBegin = 0, lastBatchCount = 0, lastSamp = 0;
  bool renderedAny = false;
  bool accFresh = false;         // the accumulator has not been written this frame: the first k_accumulate starts from zero
  bool countWork = false, kernelTiming = false;
  float kernelMs[6] = {};
  std::vector<hipEvent_t> timerEvents; // pairs, created on demand (see KernelTimer)
  std::vector<int> timerLog;           // kernel family of each recorded pair
  uint32_t kernelLaunches[6] = {};
  DevBuf<rthip_counters> totals; // frame totals, device memory (read back by rthip_get_counters)

  // Second set of per-batch resources.  With a depth cap of 1 a batch needs no host round trip, so two batches are kept in
  // flight on two streams: the VALU-bound extend launch of one overlaps the memory-bound shade / shadow / accumulate
  // launches of the other.  swapSlot() exchanges this set with the members above, so every helper keeps using `c->...`.
  struct Slot
  {
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr, evDone = nullptr;
    uint32_t fetchCursor = 0;
    DevBuf<float4> rayO, rayD, s0;
    DevBuf<float2> s1;
    DevBuf<float> nz;
    DevBuf<uint2> hit;
    DevBuf<uint8_t> live, mask;
    DevBuf<uint32_t> spill, spill2, fetch, listA, listB;
  } alt;
  hipEvent_t evDone = nullptr;   // this slot's "batch finished" event (recorded after every batch)
  hipEvent_t evFrame = nullptr;  // fork point: the second slot's stream waits for the frame prologue on the main stream
  hipEvent_t lastAcc = nullptr;  // evDone of the most recent batch, whichever slot ran it: the next accumulate waits for it
  hipStream_t lastAccStream = nullptr;
  bool altPending = false;       // work may be in flight on the second slot's stream that the main stream has not joined
  uint32_t frameParity = 0;      // camera constants and counter totals are double-buffered per frame (frames overlap)
  uint32_t pipelineBatches = 2;  // pieces a call's passes are split into when the second slot is used (depth cap 1, >= 2 passes); 0 = off
};

namespace
{
DeviceScene devScene(const rthip_ctx* c)
{
  DeviceScene s;
  s.modelNodes = c->modelNodes.p;
  s.nodes = c->nodes.p;
  s.tris = c->tris.p;
  s.numModelNodes = c->info.numModelNodes;
  s.cullRel = c->info.cullRel;
  s.cullAbs = c->info.cullAbs;
  return s;
}
constexpr size_t kCounterWords = ((sizeof(DeviceCounters) + 255) / 256) * 64; // DeviceCounters rounded up to 256 B, in words
DeviceCounters* batchCounters(const rthip_ctx* c) { return reinterpret_cast<DeviceCounters*>(c->fetch.p); }
uint32_t* fetchBase(const rthip_ctx* c) { return c->fetch.p + kCounterWords; }

CameraConsts* frameCamera(const rthip_ctx* c) { return c->camConsts.p + c->frameParity; }
rthip_counters* frameTotals(const rthip_ctx* c) { return c->totals.p + c->frameParity; }

// makes the main stream wait for whatever the second slot still has in flight (called before anything reads results)
int joinSlots(rthip_ctx* c)
{
  if (c->altPending)
  {
    HIPCHK(hipStreamWaitEvent(c->stream, c->alt.evDone, 0));
    c->altPending = false;
  }
  return 0;
}

void swapSlot(rthip_ctx* c)
{
  rthip_ctx::Slot& a = c->alt;
  std::swap(c->stream, a.stream), std::swap(c->stream2, a.stream2);
  std::swap(c->evFork, a.evFork), std::swap(c->evJoin, a.evJoin), std::swap(c->evDone, a.evDone);
  std::swap(c->fetchCursor, a.fetchCursor);
  std::swap(c->rayO, a.rayO), std::swap(c->rayD, a.rayD), std::swap(c->s0, a.s0), std::swap(c->s1, a.s1), std::swap(c->nz, a.nz);
  std::swap(c->hit, a.hit), std::swap(c->live, a.live), std::swap(c->mask, a.mask);
  std::swap(c->spill, a.spill), std::swap(c->spill2, a.spill2), std::swap(c->fetch, a.fetch);
  std::swap(c->listA, a.listA), std::swap(c->listB, a.listB);
}

PathBuffers pathBuffers(const rthip_ctx* c)
{
  PathBuffers pb;
  pb.rayO = c->rayO.p, pb.rayD = c->rayD.p, pb.hit = c->hit.p, pb.s0 = c->s0.p, pb.s1 = c->s1.p, pb.nz = c->nz.p, pb.live = c->live.p, pb.mask = c->mask.p;
  return pb;
}

// Per-kernel timing without stalling anything: an event pair is recorded around a launch on the stream it goes to and
// logged; the pairs are resolved when the times are read (rthip_get_kernel_times), after the streams have drained.  So
// batches and frames overlap exactly as they do untimed, and a kernel's time is what rocprofv3 reports for it.
struct KernelTimer
{
  rthip_ctx* c;
  int which;
  hipStream_t stream;
  int pair = -1;
  KernelTimer(rthip_ctx* ctx, int k, hipStream_t st = nullptr) : c(ctx), which(k), stream(st ? st : ctx->stream)
  {
    if (!c->kernelTiming || c->timerLog.size() >= kMaxTimedLaunches)
      return;
    pair = (int)c->timerLog.size();
    while (c->timerEvents.size() < 2 * (size_t)(pair + 1))
    {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess)
      {
        pair = -1;
        return;
      }
      c->timerEvents.push_back(e);
    }
    c->timerLog.push_back(which);
    (void)hipEventRecord(c->timerEvents[2 * (size_t)pair], stream);
  }
  ~KernelTimer()
  {
    c->kernelLaunches[which]++;
    if (pair >= 0)
      (void)hipEventRecord(c->timerEvents[2 * (size_t)pair + 1], stream);
  }
};

// static mode: one run of runLen items per wave, as many blocks as it takes
void staticGeometry(uint32_t total, uint32_t& blocks, uint32_t& runLen)
{
  runLen = kMaxRun;
  const uint32_t waves = (total + runLen - 1) / runLen;
  blocks = std::max(1u, (waves + kWavesPerBlock - 1) / kWavesPerBlock);
}
// dynamic mode: persistent grid, chunks sized so that every resident wave fetches ~8 times when there is that much work
void dynamicGeometry(const rthip_ctx* c, uint32_t total, uint32_t& blocks, uint32_t& chunkLen)
{
  const uint32_t waves = c->traceBlocks * kWavesPerBlock;
  uint32_t cl = (total + waves * 8u - 1) / (waves * 8u);
  cl = ((cl + 63u) / 64u) * 64u;
  chunkLen = std::max(64u, std::min(cl, 4u * kChunk)); // a fetched span is scanned kChunk items at a time
  const uint32_t wavesNeeded = (total + chunkLen - 1) / chunkLen;
  blocks = std::max(kFetchShards, std::min(c->traceBlocks, (wavesNeeded + kWavesPerBlock - 1) / kWavesPerBlock));
}

int ensurePathBuffers(rthip_ctx* c, size_t numSlots)
{
  int rc;
  if ((rc = c->rayO.ensure(numSlots)) || (rc = c->rayD.ensure(numSlots)) || (rc = c->s0.ensure(numSlots)) || (rc = c->s1.ensure(numSlots)) || (rc = c->nz.ensure(numSlots)) ||
      (rc = c->hit.ensure(numSlots)) || (rc = c->live.ensure(numSlots + 64)) || (rc = c->mask.ensure(numSlots)))
    return rc;
  uint32_t blocks, runLen;
  staticGeometry((uint32_t)numSlots, blocks, runLen);
  blocks = std::max(blocks, c->traceBlocks);
  if ((rc = c->spill.ensure((size_t)c->spillCap * blocks * kBlock)) || (rc = c->spill2.ensure((size_t)c->spillCap * blocks * kBlock)))
    return rc;
  if ((rc = c->fetch.ensure(kCounterWords + (size_t)kMaxTraceLaunches * kFetchShards * kFetchStride)))
    return rc;
  c->sparseCap = (uint32_t)std::max<size_t>(4096, numSlots / 32);
  if ((rc = c->listA.ensure(c->sparseCap)) || (rc = c->listB.ensure(c->sparseCap)))
    return rc;
  return 0;
}

// list == nullptr: dense launch over all numSlots slots; otherwise sparse launch over the device-side list whose
// length is at most `bound`
template <int MODE>
int launchTrace(rthip_ctx* c, uint32_t numSlots, uint32_t depth, uint32_t sampBegin, float tmin, float tmax, const uint32_t* list = nullptr,
                uint32_t bound = 0, bool foldFinal = false, bool sideStream = false)
{
  TraceArgs a;
  a.scene = devScene(c);
  a.pb = pathBuffers(c);
  a.lights = c->lights.p;
  a.geom = c->geom;
  a.cam = frameCamera(c);
  a.sampBegin = sampBegin;
  a.numSlots = numSlots;
  a.list = list;
  a.listCount = list ? &batchCounters(c)->listCount[depth] : nullptr;
  a.depth = depth;
  a.spill = sideStream ? c->spill2.p : c->spill.p;
  a.counters = batchCounters(c);
  a.tmin = tmin;
  a.tmax = tmax;
  a.anyOut = c->anyOut.p;
  a.refillAt = c->tuneRefill;
  a.leafTrigger = c->tuneLeaf;
  a.stealSpans = c->tuneSteal;
  a.foldFinal = foldFinal ? 1u : 0u;
  a.aovDepths = c->aovDepths;
  a.aovShadow = c->aovShadow.p;
  uint32_t blocks;
  constexpr bool staticMode = MODE == PRIMARY_GROUP;
  a.runLen = 0;
  a.chunkLen = 0;
  a.fetch = nullptr;
  if (staticMode)
    staticGeometry(numSlots, blocks, a.runLen);
  else
  {
    dynamicGeometry(c, list ? bound : numSlots, blocks, a.chunkLen);
    if (c->fetchCursor >= kMaxTraceLaunches)
      return RTHIP_E_LIMIT;
    a.fetch = fetchBase(c) + (size_t)(c->fetchCursor++) * kFetchShards * kFetchStride;
  }
  if ((size_t)c->spillCap * blocks * kBlock > c->spill.n)
    return RTHIP_E_LIMIT;
  hipStream_t stream = sideStream ? c->stream2 : c->stream;
  KernelTimer kt(c, isClosest(MODE) ? 1 : 3, stream);
  if (c->countWork)
    hipLaunchKernelGGL((k_trace<MODE, true>), dim3(blocks), dim3(kBlock), 0, stream, a);
  else
    hipLaunchKernelGGL((k_trace<MODE, false>), dim3(blocks), dim3(kBlock), 0, stream, a);
  return (int)hipGetLastError();
}

// one batch of `count` sample passes starting at sampBegin
int renderBatch(rthip_ctx* c, uint32_t sampBegin, uint32_t count, hipEvent_t accumulateAfter = nullptr)
{
  const uint32_t P = c->geom.numLocalPixels;
  const uint32_t numSlots = P * count;
  const uint32_t shadeBlocksDense = std::max(1u, std::min((numSlots + kBlock - 1) / kBlock, (uint32_t)c->numCUs * 8u));
  HIPCHK(hipMemsetAsync(c->fetch.p, 0, c->fetch.n * sizeof(uint32_t), c->stream)); // batch counters + work-fetch counters
  c->fetchCursor = 0;
  if (c->aovDepths)
  {
    HIPCHK(hipMemsetAsync(c->aovTri.p, 0xff, (size_t)count * c->aovDepths * P * sizeof(uint32_t), c->stream));
    HIPCHK(hipMemsetAsync(c->aovShadow.p, 0xff, (size_t)count * c->aovDepths * P, c->stream));
  }
  const bool fused = c->fusedPrimary;
  if (!fused)
  {
    KernelTimer kt(c, 0);
    hipLaunchKernelGGL(k_generate, dim3((numSlots + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->geom, frameCamera(c), pathBuffers(c),
                       sampBegin, numSlots);
    HIPCHK(hipGetLastError());
  }
  const bool haveDirLights = c->hostLights.numDir > 0;
  ShadeArgs sa;
  sa.g = c->geom;
  sa.scene = devScene(c);
  sa.pb = pathBuffers(c);
  sa.lights = c->lights.p;
  sa.cam = frameCamera(c);
  sa.counters = batchCounters(c);
  sa.numSlots = numSlots;
  sa.maxDepth = c->maxDepth;
  sa.sampBegin = sampBegin;
  sa.primaryInRegs = fused ? 1u : 0u;
  sa.aovDepths = c->aovDepths;
  sa.aovTri = c->aovTri.p;
  sa.aovShadow = c->aovShadow.p;
  const float tmin = 1e-4f, tmax = 1e20f; // CPURenderer.cpp:194, CPUScene.cpp:34 (double literals narrowed to float)
  bool sparse = false;
  uint32_t bound = numSlots; // upper bound of the paths alive at the current depth
  for (uint32_t d = 0;; ++d)
  {
    const bool last = d == c->maxDepth;
    const uint32_t* list = sparse ? ((d & 1u) ? c->listB.p : c->listA.p) : nullptr;
    uint32_t* listNext = (d & 1u) ? c->listA.p : c->listB.p;
    int rc;
    // shadow rays of the hits of depth d-1; at the last depth the same launch also folds the lights and finishes the paths
    // A depth's shadow launch (reads rayO, writes mask) and closest-hit launch (reads rayO/rayD, writes hit) are
    // independent, and each ends in a tail where a few long rays keep a few waves busy.  On two streams the second
    // launch's blocks move in as the first one's run dry, so one tail per depth is hidden; shade(d) joins both.
    const bool side = d > 0 && !last && haveDirLights && c->overlapShadow;
    if (side)
    {
      HIPCHK(hipEventRecord(c->evFork, c->stream));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->evFork, 0));
    }
    if (d > 0 && haveDirLights && (rc = launchTrace<SHADOW>(c, numSlots, d, sampBegin, tmin, tmax, list, bound, last, side)))
      return rc;
    if (side)
    {
      HIPCHK(hipEventRecord(c->evJoin, c->stream2));
    }
    if (last && haveDirLights)
      break;
    if (!last)
    {
      // coherent primary rays over a scene of a few nodes cost the same everywhere: static 64-slot groups, no atomics;
      // otherwise traversal length varies pixel to pixel and lanes are refilled from dynamically fetched chunks
      if (d == 0 && fused && c->info.numNodes <= 64)
        rc = launchTrace<PRIMARY_GROUP>(c, numSlots, 0, sampBegin, tmin, tmax);
      else if (d == 0 && fused)
        rc = launchTrace<PRIMARY>(c, numSlots, 0, sampBegin, tmin, tmax);
      else
        rc = launchTrace<CLOSEST>(c, numSlots, d, sampBegin, tmin, tmax, list, bound);
      if (rc)
        return rc;
    }
    if (side)
      HIPCHK(hipStreamWaitEvent(c->stream, c->evJoin, 0));
    sa.depth = d;
    sa.finalFold = last ? 1u : 0u;
    sa.list = list;
    sa.listNext = listNext;
    {
      KernelTimer kt(c, 2);
      if (sparse)
      {
        const uint32_t blocks = std::max(1u, std::min((bound + kBlock - 1) / kBlock, (uint32_t)c->numCUs * 8u));
        hipLaunchKernelGGL(k_shade<true>, dim3(blocks), dim3(kBlock), 0, c->stream, sa);
      }
      else
        hipLaunchKernelGGL(k_shade<false>, dim3(shadeBlocksDense), dim3(kBlock), 0, c->stream, sa);
      HIPCHK(hipGetLastError());
    }
    if (last)
      break;
    if (!sparse && d + 1 < c->maxDepth)
    {
      // how many paths go on?  stop when none do; switch to slot lists when few do (open scenes lose most paths
      // within a bounce or two, and scanning every slot's liveness byte for a handful of paths is wasted work)
      unsigned long long alive = 0;
      HIPCHK(hipMemcpyAsync(&alive, &batchCounters(c)->alive[d + 1], sizeof(alive), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (alive == 0)
        break;
      bound = (uint32_t)alive;
      if (alive <= c->sparseCap / 2)
      {
        KernelTimer kt(c, 5);
        hipLaunchKernelGGL(k_compact, dim3((numSlots + kBlock * 16 - 1) / (kBlock * 16)), dim3(kBlock), 0, c->stream, pathBuffers(c), numSlots,
                           d + 1, listNext, c->sparseCap, batchCounters(c));
        HIPCHK(hipGetLastError());
        sparse = true;
      }
    }
  }
  if (accumulateAfter) // the previous batch ran on the other slot: per-pixel accumulation stays in sample order
    HIPCHK(hipStreamWaitEvent(c->stream, accumulateAfter, 0));
  {
    KernelTimer kt(c, 4);
    hipLaunchKernelGGL(k_accumulate, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->geom, pathBuffers(c), c->acc.p, count, c->accFresh ? 1u : 0u);
    c->accFresh = false;
    HIPCHK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, c->stream, batchCounters(c), frameTotals(c));
  HIPCHK(hipGetLastError());
  c->lastBatchBegin = sampBegin;
  c->lastBatchCount = count;
  c->lastSamp = sampBegin + count - 1;
  c->renderedAny = true;
  return 0;
}

int renderPasses(rthip_ctx* c, uint32_t sampBegin, uint32_t count, bool sync)
{
  if (!c->haveScene || !c->haveFrame)
    return RTHIP_E_STATE;
  if (count == 0 || sampBegin + count > c->geom.numSamples)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  // Two batches in flight (see rthip_ctx::Slot) when a batch needs no host round trip: the passes are split in two pieces
  // (more pieces measured slower: 3 -> -13 %, 8 -> -33 %) that alternate between the slots.  Nothing joins the slots at
  // the end of the call: the next frame's first piece starts while this frame's second piece is still shading, and
  // only the accumulate kernels are chained (per-pixel sums stay in sample order).
  const bool pipelined = c->pipelineBatches && c->maxDepth == 1 && c->aovDepths == 0 && count >= 2;
  uint32_t piece = c->batchCap;
  if (pipelined)
  {
    piece = std::min(c->batchCap, (count + 1) / 2);
    const uint32_t keepSparseCap = c->sparseCap;
    swapSlot(c);
    int rc = ensurePathBuffers(c, (size_t)c->geom.numLocalPixels * piece);
    swapSlot(c);
    c->sparseCap = keepSparseCap;
    if (rc)
      return rc;
    HIPCHK(hipEventRecord(c->evFrame, c->stream));              // the frame prologue (and everything before it) ...
    HIPCHK(hipStreamWaitEvent(c->alt.stream, c->evFrame, 0));   // ... precedes the second slot's work
  }
  uint32_t done = 0;
  for (uint32_t k = 0; done < count; ++k)
  {
    const uint32_t n = std::min(piece, count - done);
    const bool useAlt = pipelined && (k & 1u);
    if (useAlt)
      swapSlot(c);
    hipEvent_t after = (c->lastAcc && c->lastAccStream != c->stream) ? c->lastAcc : nullptr;
    int rc = renderBatch(c, sampBegin + done, n, after);
    if (!rc)
      rc = (int)hipEventRecord(c->evDone, c->stream);
    c->lastAcc = c->evDone;
    c->lastAccStream = c->stream;
    if (useAlt)
    {
      c->altPending = true;
      swapSlot(c);
    }
    if (rc)
      return rc;
    done += n; // (a later batch on the same slot reuses its path buffers: same stream, so it runs after this one)
  }
  if (sync)
  {
    if (int rc = joinSlots(c))
      return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}
} // namespace

extern "C" {

const char* rthip_error_string(int code)
{
  switch (code)
  {
  case 0: return "success";
  case RTHIP_E_ARG: return "rthip: invalid argument";
  case RTHIP_E_STATE: return "rthip: call out of order (scene or frame missing)";
  case RTHIP_E_BOUNDS: return "rthip: camera outside the origin bound the scene was flattened for";
  case RTHIP_E_NODEVICE: return "rthip: no usable HIP device";
  case RTHIP_E_LIMIT: return "rthip: size beyond a compiled-in limit";
  default: return code > 0 ? hipGetErrorString((hipError_t)code) : "rthip: unknown error";
  }
}

int rthip_create(int device, rthip_ctx** out)
{
  if (!out)
    return RTHIP_E_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
    return RTHIP_E_NODEVICE;
  HIPCHK(hipSetDevice(device));
  rthip_ctx* c = new (std::nothrow) rthip_ctx();
  if (!c)
    return RTHIP_E_ARG;
  c->device = device;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* t = getenv("RTHIP_TUNE")) // "refill,leaf[,blocksPerCU]" -- tuning experiments only
  {
    unsigned r = 0, l = 0, b = 0, st = 1, ov = 1, pl = 1;
    const int n = sscanf(t, "%u,%u,%u,%u,%u,%u", &r, &l, &b, &st, &ov, &pl);
    if (n >= 2 && r >= 1 && r <= 64 && l >= 1 && l <= 64)
      c->tuneRefill = r, c->tuneLeaf = l;
    if (n >= 3 && b >= 1 && b <= 8)
      c->tuneBlocks = b;
    if (n >= 4)
      c->tuneSteal = st ? kFetchShards : 1u;
    if (n >= 5)
      c->overlapShadow = ov ? 1u : 0u;
    if (n >= 6)
      c->pipelineBatches = std::min(pl, 16u);
  }
  c->traceBlocks = (uint32_t)c->numCUs * c->tuneBlocks; // 6 blocks x 4 waves = 24 waves per CU: what 24 KB of LDS per block admits
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&c->evA));
  HIPCHK(hipEventCreate(&c->evB));
  HIPCHK(hipEventCreate(&c->evK0));
  HIPCHK(hipEventCreate(&c->evK1));
  HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->evDone, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->evFrame, hipEventDisableTiming));
  HIPCHK(hipStreamCreateWithFlags(&c->alt.stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&c->alt.stream2, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->alt.evFork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->alt.evJoin, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->alt.evDone, hipEventDisableTiming));
  *out = c;
  return 0;
}

int rthip_destroy(rthip_ctx* c)
{
  if (!c)
    return RTHIP_E_ARG;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->alt.stream);
  (void)hipStreamSynchronize(c->stream);
  c->modelNodes.release(), c->nodes.release(), c->tris.release(), c->lights.release(), c->tileIds.release(), c->camConsts.release();
  c->rayO.release(), c->rayD.release(), c->s0.release(), c->s1.release(), c->nz.release(), c->hit.release(), c->live.release(), c->mask.release();
  c->anyOut.release(), c->acc.release(), c->img.release(), c->rowScratch.release(), c->aovTri.release(), c->aovShadow.release();
  c->spill.release(), c->spill2.release(), c->fetch.release(), c->listA.release(), c->listB.release(), c->totals.release();
  (void)hipEventDestroy(c->evA), (void)hipEventDestroy(c->evB), (void)hipEventDestroy(c->evK0), (void)hipEventDestroy(c->evK1);
  for (hipEvent_t e : c->timerEvents)
    (void)hipEventDestroy(e);
  (void)hipEventDestroy(c->evFork), (void)hipEventDestroy(c->evJoin), (void)hipEventDestroy(c->evDone), (void)hipEventDestroy(c->evFrame);
  (void)hipEventDestroy(c->alt.evFork), (void)hipEventDestroy(c->alt.evJoin), (void)hipEventDestroy(c->alt.evDone);
  (void)hipStreamDestroy(c->alt.stream), (void)hipStreamDestroy(c->alt.stream2);
  c->alt.rayO.release(), c->alt.rayD.release(), c->alt.s0.release(), c->alt.s1.release(), c->alt.nz.release(), c->alt.hit.release();
  c->alt.live.release(), c->alt.mask.release(), c->alt.spill.release(), c->alt.spill2.release(), c->alt.fetch.release();
  c->alt.listA.release(), c->alt.listB.release();
  (void)hipStreamDestroy(c->stream2);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int rthip_upload_scene(rthip_ctx* c, const rtflat_info* info, const rtflat_model_node* modelNodes, const rtflat_node* nodes,
                       const rtflat_tri* tris, const rthip_dir_light* dirLights, uint32_t numDirLights,
                       const rthip_point_light* pointLights, uint32_t numPointLights)
{
  if (!c || !info || !modelNodes || !nodes || !tris || (numDirLights && !dirLights) || (numPointLights && !pointLights))
    return RTHIP_E_ARG;
  if (info->numModelNodes == 0 || info->numTris == 0 || info->numNodes == 0)
    return RTHIP_E_ARG;
  if (numDirLights > kMaxLights || numPointLights > kMaxLights || info->numTris >= (1u << 28) - 1u)
    return RTHIP_E_LIMIT;
  // the per-lane stack must hold up to three postponed children per level of the deepest chain
  if (3 * info->maxDepthBlas + 2 > kMaxStack)
    return RTHIP_E_LIMIT;
  // host-side shape checks before any kernel dereferences these indices
  for (uint32_t k = 0; k < info->numModelNodes; ++k)
  {
    const rtflat_model_node& mn = modelNodes[k];
    if (mn.subtreeSize == 0 || k + mn.subtreeSize > info->numModelNodes)
      return RTHIP_E_ARG;
    if (mn.triCount > 0)
    {
      const uint32_t ref = mn.blas;
      if (rtflat_is_leaf(ref) ? (rtflat_leaf_first(ref) + rtflat_leaf_count(ref) > info->numTris) : (ref >= info->numNodes))
        return RTHIP_E_ARG;
    }
  }
  for (uint32_t k = 0; k < info->numNodes; ++k)
    for (uint32_t ref : nodes[k].child)
    {
      if (ref == RTFLAT_INVALID)
        continue; // unused slot (its box is inverted)
      if (rtflat_is_leaf(ref) ? (rtflat_leaf_first(ref) + rtflat_leaf_count(ref) > info->numTris) : (ref >= info->numNodes || ref <= k))
        return RTHIP_E_ARG; // children follow their parent (pre-order): no cycles
    }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->alt.stream)); // frames still in flight on either slot read the old scene
  c->altPending = false;
  HIPCHK(hipStreamSynchronize(c->stream));
  int rc;
  if ((rc = c->modelNodes.ensure(info->numModelNodes)) || (rc = c->nodes.ensure(info->numNodes)) || (rc = c->tris.ensure(info->numTris)) ||
      (rc = c->lights.ensure(1)))
    return rc;
  HIPCHK(hipMemcpy(c->modelNodes.p, modelNodes, sizeof(rtflat_model_node) * info->numModelNodes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->nodes.p, nodes, sizeof(rtflat_node) * info->numNodes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->tris.p, tris, sizeof(rtflat_tri) * info->numTris, hipMemcpyHostToDevice));
  Lights& L = c->hostLights;
  std::memset(&L, 0, sizeof(L));
  L.numDir = numDirLights;
  L.numPoint = numPointLights;
  for (uint32_t i = 0; i < numDirLights; ++i)
    L.dir[i] = dirLights[i];
  for (uint32_t i = 0; i < numPointLights; ++i)
    L.point[i] = pointLights[i];
  HIPCHK(hipMemcpy(c->lights.p, &L, sizeof(L), hipMemcpyHostToDevice));
  c->info = *info;
  c->spillCap = 3 * info->maxDepthBlas + 2 > kStackLds ? 3 * info->maxDepthBlas + 2 - kStackLds : 1u;
  c->spill.release(); // sized per scene in ensurePathBuffers
  c->haveScene = true;
  c->haveFrame = false;
  return 0;
}

int rthip_begin(rthip_ctx* c, const rthip_camera* camera, const rthip_frame_desc* f)
{
  if (!c || !camera || !f || f->width == 0 || f->height == 0 || f->numSamples == 0)
    return RTHIP_E_ARG;
  if (!c->haveScene)
    return RTHIP_E_STATE;
  if (f->maxDepth == 0 || f->maxDepth > kMaxDepthSlots || f->maxDepth > 254)
    return RTHIP_E_LIMIT;
  const uint32_t ts = f->tileSize ? f->tileSize : 64u;
  if (ts < 8u || (ts & (ts - 1u)) || f->width > 65535u || f->height > 65535u)
    return RTHIP_E_ARG; // tiles are powers of two >= 8; tile origins are packed into 16 bits each
  // the padding of the BVH boxes was sized for ray origins within originBound of the scene centre
  {
    float d2 = 0;
    for (int k = 0; k < 3; ++k)
    {
      const float ctr = 0.5f * (c->info.sceneLo[k] + c->info.sceneHi[k]);
      d2 += (camera->position[k] - ctr) * (camera->position[k] - ctr);
    }
    if (!(d2 <= c->info.originBound * c->info.originBound))
      return RTHIP_E_BOUNDS;
  }
  HIPCHK(hipSetDevice(c->device));
  // No host synchronisation here: a frame is enqueued behind the previous one on the context's stream, so a caller that
  // uses rthip_enqueue_passes keeps the GPU busy across frames (buffers only ever grow, and growing one synchronises).
  const uint32_t tilesX = (f->width + ts - 1) / ts, tilesY = (f->height + ts - 1) / ts;
  FrameGeom g{};
  g.width = f->width, g.height = f->height, g.numSamples = f->numSamples;
  g.tileSize = ts, g.tilesX = tilesX;
  c->hostTileIds.clear();
  if (f->numTiles)
  {
    if (!f->tileIds)
      return RTHIP_E_ARG;
    for (uint32_t k = 0; k < f->numTiles; ++k)
    {
      if (f->tileIds[k] >= tilesX * tilesY)
        return RTHIP_E_ARG;
      c->hostTileIds.push_back(f->tileIds[k]);
    }
  }
  else
    for (uint32_t k = 0; k < tilesX * tilesY; ++k)
      c->hostTileIds.push_back(k);
  g.numLocalTiles = (uint32_t)c->hostTileIds.size();
  const uint64_t P64 = (uint64_t)g.numLocalTiles * ts * ts;
  if (P64 == 0 || P64 > (1ull << 30))
    return RTHIP_E_LIMIT;
  g.numLocalPixels = (uint32_t)P64;
  int rc;
  {
    const uint32_t* before = c->tileIds.p;
    if ((rc = c->tileIds.ensure(g.numLocalTiles)))
      return rc;
    if (c->tileIds.p != before)
      c->hostTileXY.clear();
  }
  {
    std::vector<uint32_t> xy(g.numLocalTiles);
    for (uint32_t k = 0; k < g.numLocalTiles; ++k)
      xy[k] = ((c->hostTileIds[k] % tilesX) * ts) | (((c->hostTileIds[k] / tilesX) * ts) << 16);
    if (xy != c->hostTileXY) // frames of one animation keep their tiling: upload the table only when it changes
    {
      HIPCHK(hipStreamSynchronize(c->alt.stream)); // kernels of the previous frame (either slot) may still read the old table
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipMemcpy(c->tileIds.p, xy.data(), sizeof(uint32_t) * g.numLocalTiles, hipMemcpyHostToDevice));
      c->hostTileXY.swap(xy);
    }
  }
  g.tileXY = c->tileIds.p;
  g.tileShift = (uint32_t)__builtin_ctz(ts);
  g.sampleMagic = g.numLocalTiles > 1 ? (uint32_t)(((1ull << 32) + g.numLocalTiles - 1) / g.numLocalTiles) : 0u;
  c->geom = g;
  c->camera = *camera;
  c->maxDepth = f->maxDepth;
  c->aovDepths = std::min(f->aovDepths, f->maxDepth);
  // sample passes per launch batch: as many as fit in 2^25 paths (about 90 B of state each)
  const uint64_t pathCap = 1ull << 25;
  c->batchCap = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(f->numSamples, pathCap / P64));
  if (c->aovDepths)
    c->batchCap = std::min(c->batchCap, 4u);
  // sampleOf's multiply-high division is exact while (batch * tiles) * tiles < 2^32
  while (c->batchCap > 1 && (uint64_t)c->batchCap * g.numLocalTiles * g.numLocalTiles >= (1ull << 32))
    c->batchCap--;
  if ((uint64_t)c->batchCap * g.numLocalTiles * g.numLocalTiles >= (1ull << 32))
    return RTHIP_E_LIMIT;
  const size_t numSlots = (size_t)P64 * c->batchCap;
  if ((rc = ensurePathBuffers(c, numSlots)) || (rc = c->acc.ensure(P64 * 3)) || (rc = c->img.ensure(P64 * 3)) || (rc = c->camConsts.ensure(2)) ||
      (rc = c->totals.ensure(2)))
    return rc;
  if (c->aovDepths)
  {
    if ((rc = c->aovTri.ensure(numSlots * c->aovDepths)) || (rc = c->aovShadow.ensure(numSlots * c->aovDepths)))
      return rc;
  }
  c->accFresh = true; // CPURenderer.cpp:142-143: the clear is folded into the first k_accumulate (or done on a read before it)
  c->frameParity ^= 1u; // the previous frame's second slot may still be reading its camera and folding its totals
  hipLaunchKernelGGL(k_camera, dim3(1), dim3(64), 0, c->stream, *camera, frameCamera(c), frameTotals(c));
  HIPCHK(hipGetLastError());
  c->renderedAny = false;
  c->haveFrame = true;
  return 0;
}

int rthip_render_passes(rthip_ctx* c, uint32_t sampBegin, uint32_t count, float* msOut)
{
  if (!c)
    return RTHIP_E_ARG;
  if (!c->haveScene || !c->haveFrame)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventRecord(c->evA, c->stream));
  int rc = renderPasses(c, sampBegin, count, false);
  if (!rc)
    rc = joinSlots(c);
  if (rc)
    return rc;
  HIPCHK(hipEventRecord(c->evB, c->stream));
  HIPCHK(hipEventSynchronize(c->evB));
  if (msOut)
    HIPCHK(hipEventElapsedTime(msOut, c->evA, c->evB));
  return 0;
}

int rthip_enqueue_passes(rthip_ctx* c, uint32_t sampBegin, uint32_t count)
{
  if (!c)
    return RTHIP_E_ARG;
  return renderPasses(c, sampBegin, count, false);
}

int rthip_synchronize(rthip_ctx* c)
{
  if (!c)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (int rc = joinSlots(c))
    return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// the accumulator is cleared lazily: by the first k_accumulate of a frame, or here when it is read before that
static int settleAccumulator(rthip_ctx* c)
{
  if (c->accFresh)
  {
    HIPCHK(hipMemsetAsync(c->acc.p, 0, c->acc.n * sizeof(float), c->stream));
    c->accFresh = false;
  }
  return 0;
}

static int readTiled(rthip_ctx* c, const float* tileMajorDev, float* rgbOut)
{
  const size_t full = (size_t)c->geom.width * c->geom.height * 3;
  int rc = c->rowScratch.ensure(full);
  if (rc)
    return rc;
  // pixels of tiles this context does not own keep the caller's values
  HIPCHK(hipMemcpyAsync(c->rowScratch.p, rgbOut, full * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_untile, dim3((c->geom.numLocalPixels + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->geom, tileMajorDev,
                     c->rowScratch.p, c->geom.numLocalPixels);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(rgbOut, c->rowScratch.p, full * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int rthip_read_accumulator(rthip_ctx* c, float* rgbOut)
{
  if (!c || !rgbOut)
    return RTHIP_E_ARG;
  if (!c->haveFrame)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  if (int rcJoin = joinSlots(c))
    return rcJoin;
  if (int rc = settleAccumulator(c))
    return rc;
  return readTiled(c, c->acc.p, rgbOut);
}

int rthip_read_image(rthip_ctx* c, float* rgbOut)
{
  if (!c || !rgbOut)
    return RTHIP_E_ARG;
  if (!c->haveFrame)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  if (int rcJoin = joinSlots(c))
    return rcJoin;
  const uint32_t nf = c->geom.numLocalPixels * 3;
  if (int rc = settleAccumulator(c))
    return rc;
  const float resolver = c->renderedAny ? float(c->geom.numSamples) / float(c->lastSamp + 1) : 1.0f; // CPURenderer.cpp:196
  hipLaunchKernelGGL(k_tonemap, dim3((nf + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->acc.p, c->img.p, nf, resolver);
  HIPCHK(hipGetLastError());
  return readTiled(c, c->img.p, rgbOut);
}

int rthip_read_aov(rthip_ctx* c, uint32_t samp, uint32_t depth, uint32_t* triIdOut, uint8_t* shadowOut)
{
  if (!c || !triIdOut || !shadowOut)
    return RTHIP_E_ARG;
  if (!c->haveFrame || !c->renderedAny || !c->aovDepths)
    return RTHIP_E_STATE;
  if (depth >= c->aovDepths || samp < c->lastBatchBegin || samp >= c->lastBatchBegin + c->lastBatchCount)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (int rcJoin = joinSlots(c))
    return rcJoin;
  const uint32_t P = c->geom.numLocalPixels;
  const size_t off = ((size_t)(samp - c->lastBatchBegin) * c->aovDepths + depth) * P;
  std::vector<uint32_t> tri(P);
  std::vector<uint8_t> sh(P);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(tri.data(), c->aovTri.p + off, P * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(sh.data(), c->aovShadow.p + off, P, hipMemcpyDeviceToHost));
  const FrameGeom& g = c->geom;
  const uint32_t ts = g.tileSize, per = ts * ts;
  for (uint32_t i = 0; i < P; ++i)
  {
    const uint32_t lt = i / per, k = i - lt * per;
    const uint32_t tile = c->hostTileIds[lt];
    const uint32_t ty = tile / g.tilesX, tx = tile - ty * g.tilesX;
    const uint32_t blk = k >> 6, l = k & 63u, bpr = ts >> 3;
    const uint32_t by = blk / bpr, bx = blk - by * bpr;
    const uint32_t w = tx * ts + bx * 8 + (l & 7u), h = ty * ts + by * 8 + (l >> 3);
    if (w < g.width && h < g.height)
    {
      triIdOut[(size_t)h * g.width + w] = tri[i];
      shadowOut[(size_t)h * g.width + w] = sh[i];
    }
  }
  return 0;
}

int rthip_accumulator_device(rthip_ctx* c, float** devPtr, uint64_t* numFloats)
{
  if (!c || !devPtr || !numFloats)
    return RTHIP_E_ARG;
  if (!c->haveFrame)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  if (int rcJoin = joinSlots(c))
    return rcJoin;
  if (int rc = settleAccumulator(c))
    return rc;
  *devPtr = c->acc.p;
  *numFloats = (uint64_t)c->geom.numLocalPixels * 3;
  return 0;
}

int rthip_copy_accumulator_to(rthip_ctx* c, float* dstDev, uint64_t numFloats)
{
  if (!c || !dstDev)
    return RTHIP_E_ARG;
  if (!c->haveFrame)
    return RTHIP_E_STATE;
  if (numFloats != (uint64_t)c->geom.numLocalPixels * 3)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (int rcJoin = joinSlots(c))
    return rcJoin;
  if (int rc = settleAccumulator(c))
    return rc;
  HIPCHK(hipMemcpyAsync(dstDev, c->acc.p, numFloats * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int rthip_untile(rthip_ctx* c, const float* tileMajorDev, const uint32_t* tileIds, uint32_t numTiles, float* rowMajorDev)
{
  if (!c || !tileMajorDev || !tileIds || !numTiles || !rowMajorDev)
    return RTHIP_E_ARG;
  if (!c->haveFrame)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  const uint32_t ts = c->geom.tileSize;
  const uint32_t tilesY = (c->geom.height + ts - 1) / ts;
  for (uint32_t k = 0; k < numTiles; ++k)
    if (tileIds[k] >= c->geom.tilesX * tilesY)
      return RTHIP_E_ARG;
  DevBuf<uint32_t> ids;
  int rc = ids.ensure(numTiles);
  if (rc)
    return rc;
  {
    std::vector<uint32_t> xy(numTiles);
    for (uint32_t k = 0; k < numTiles; ++k)
      xy[k] = ((tileIds[k] % c->geom.tilesX) * ts) | (((tileIds[k] / c->geom.tilesX) * ts) << 16);
    HIPCHK(hipMemcpy(ids.p, xy.data(), sizeof(uint32_t) * numTiles, hipMemcpyHostToDevice));
  }
  FrameGeom g = c->geom;
  g.tileXY = ids.p;
  g.numLocalTiles = numTiles;
  g.numLocalPixels = numTiles * ts * ts;
  hipLaunchKernelGGL(k_untile, dim3((g.numLocalPixels + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, g, tileMajorDev, rowMajorDev,
                     g.numLocalPixels);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  ids.release();
  return 0;
}

int rthip_get_counters(rthip_ctx* c, rthip_counters* out)
{
  if (!c || !out)
    return RTHIP_E_ARG;
  if (!c->haveFrame)
  {
    std::memset(out, 0, sizeof(*out));
    return 0;
  }
  HIPCHK(hipSetDevice(c->device));
  if (int rcJoin = joinSlots(c))
    return rcJoin;
  HIPCHK(hipMemcpyAsync(out, frameTotals(c), sizeof(*out), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int rthip_set_work_counting(rthip_ctx* c, int enabled)
{
  if (!c)
    return RTHIP_E_ARG;
  c->countWork = enabled != 0;
  return 0;
}

int rthip_set_fused_primary(rthip_ctx* c, int enabled)
{
  if (!c)
    return RTHIP_E_ARG;
  c->fusedPrimary = enabled != 0;
  return 0;
}

int rthip_set_pipelining(rthip_ctx* c, int enabled)
{
  if (!c)
    return RTHIP_E_ARG;
  c->pipelineBatches = enabled ? 2u : 0u;
  return 0;
}

int rthip_set_kernel_timing(rthip_ctx* c, int enabled)
{
  if (!c)
    return RTHIP_E_ARG;
  c->kernelTiming = enabled != 0;
  if (enabled) // a fresh measurement: times and launch counts accumulate from here until they are switched on again
  {
    HIPCHK(hipSetDevice(c->device));
    if (int rc = joinSlots(c))
      return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->timerLog.clear();
    std::memset(c->kernelMs, 0, sizeof(c->kernelMs));
    std::memset(c->kernelLaunches, 0, sizeof(c->kernelLaunches));
  }
  return 0;
}

int rthip_get_kernel_times(rthip_ctx* c, float msOut[6], uint32_t launchesOut[6])
{
  if (!c || !msOut || !launchesOut)
    return RTHIP_E_ARG;
  if (!c->timerLog.empty())
  {
    HIPCHK(hipSetDevice(c->device));
    if (int rc = joinSlots(c))
      return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->stream2));
    HIPCHK(hipStreamSynchronize(c->alt.stream));
    HIPCHK(hipStreamSynchronize(c->alt.stream2));
    for (size_t k = 0; k < c->timerLog.size(); ++k)
    {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->timerEvents[2 * k], c->timerEvents[2 * k + 1]) == hipSuccess)
        c->kernelMs[c->timerLog[k]] += ms;
    }
    c->timerLog.clear();
  }
  for (int i = 0; i < 6; ++i)
  {
    msOut[i] = c->kernelMs[i];
    launchesOut[i] = c->kernelLaunches[i];
  }
  return 0;
}

// ---- probes -------------------------------------------------------------------------------------
int rthip_probe_rand01(rthip_ctx* c, const uint32_t* xyz, uint32_t n, float* out3)
{
  if (!c || !xyz || !out3 || !n)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevBuf<uint32_t> in;
  DevBuf<float> out;
  int rc;
  if ((rc = in.ensure(3 * (size_t)n)) || (rc = out.ensure(3 * (size_t)n)))
    return rc;
  HIPCHK(hipMemcpy(in.p, xyz, 3 * (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_rand01, dim3((n + 255) / 256), dim3(256), 0, c->stream, in.p, n, out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out3, out.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  in.release(), out.release();
  return 0;
}

int rthip_probe_sincos(rthip_ctx* c, const float* x, uint32_t n, float* sinOut, float* cosOut)
{
  if (!c || !x || !sinOut || !cosOut || !n)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevBuf<float> in, s, co;
  int rc;
  if ((rc = in.ensure(n)) || (rc = s.ensure(n)) || (rc = co.ensure(n)))
    return rc;
  HIPCHK(hipMemcpy(in.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_sincos, dim3((n + 255) / 256), dim3(256), 0, c->stream, in.p, n, s.p, co.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(sinOut, s.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cosOut, co.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  in.release(), s.release(), co.release();
  return 0;
}

int rthip_probe_primary_rays(rthip_ctx* c, const rthip_camera* cam, uint32_t width, uint32_t height, const uint32_t* whs, uint32_t n,
                             float* origin3, float* dir3)
{
  if (!c || !cam || !whs || !origin3 || !dir3 || !n || !width || !height)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevBuf<uint32_t> in;
  DevBuf<float> o, d;
  DevBuf<CameraConsts> kc;
  int rc;
  if ((rc = in.ensure(3 * (size_t)n)) || (rc = o.ensure(3 * (size_t)n)) || (rc = d.ensure(3 * (size_t)n)) || (rc = kc.ensure(1)))
    return rc;
  HIPCHK(hipMemcpy(in.p, whs, 3 * (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_camera, dim3(1), dim3(64), 0, c->stream, *cam, kc.p, (rthip_counters*)nullptr);
  hipLaunchKernelGGL(k_probe_rays, dim3((n + 255) / 256), dim3(256), 0, c->stream, kc.p, width, height, in.p, n, o.p, d.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(origin3, o.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(dir3, d.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  in.release(), o.release(), d.release(), kc.release();
  return 0;
}

int rthip_probe_bounce(rthip_ctx* c, const float* normalLight3, const float* rnd3, uint32_t n, float* dir3)
{
  if (!c || !normalLight3 || !rnd3 || !dir3 || !n)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevBuf<float> a, b, o;
  int rc;
  if ((rc = a.ensure(3 * (size_t)n)) || (rc = b.ensure(3 * (size_t)n)) || (rc = o.ensure(3 * (size_t)n)))
    return rc;
  HIPCHK(hipMemcpy(a.p, normalLight3, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b.p, rnd3, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_bounce, dim3((n + 255) / 256), dim3(256), 0, c->stream, a.p, b.p, n, o.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(dir3, o.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  a.release(), b.release(), o.release();
  return 0;
}

int rthip_probe_trace(rthip_ctx* c, const float* origin3, const float* dir3, uint32_t n, float tmin, float tmax, uint32_t* triIdOut,
                      float* tOut, uint8_t* anyOut)
{
  if (!c || !origin3 || !dir3 || !n || !triIdOut || !tOut || !anyOut)
    return RTHIP_E_ARG;
  if (!c->haveScene)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  int rc;
  if ((rc = ensurePathBuffers(c, n)) || (rc = c->anyOut.ensure(n)))
    return rc;
  c->haveFrame = false; // path buffers may have been reallocated
  DevBuf<float> o, d, t;
  DevBuf<uint32_t> id;
  if ((rc = o.ensure(3 * (size_t)n)) || (rc = d.ensure(3 * (size_t)n)) || (rc = t.ensure(n)) || (rc = id.ensure(n)))
    return rc;
  HIPCHK(hipMemcpy(o.p, origin3, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.p, dir3, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemsetAsync(c->fetch.p, 0, c->fetch.n * sizeof(uint32_t), c->stream)); // batch counters + work-fetch counters
  c->fetchCursor = 0;
  hipLaunchKernelGGL(k_probe_pack, dim3((n + 255) / 256), dim3(256), 0, c->stream, o.p, d.p, n, c->rayO.p, c->rayD.p, c->live.p);
  HIPCHK(hipGetLastError());
  if ((rc = launchTrace<CLOSEST>(c, n, 0, 0, tmin, tmax)) || (rc = launchTrace<ANYPROBE>(c, n, 0, 0, tmin, tmax)))
    return rc;
  hipLaunchKernelGGL(k_probe_unpack, dim3((n + 255) / 256), dim3(256), 0, c->stream, devScene(c), c->hit.p, n, id.p, t.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(triIdOut, id.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(tOut, t.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(anyOut, c->anyOut.p, (size_t)n, hipMemcpyDeviceToHost));
  o.release(), d.release(), t.release(), id.release();
  return 0;
}
}
