// rt_kernel_common.h: constants, device-side records of the scene / frame / paths, lane and pixel helpers, the light fold -- part of the gfx950 kernel library (see rthip.hip for the pipeline overview and DESIGN.md section 5).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "rt_device.h"

using rt::vec3;
using rtd::CameraConsts;

namespace rtk
{
constexpr uint32_t kBlock = 256;       // 4 waves
constexpr uint32_t kWavesPerBlock = kBlock / 64;
constexpr uint32_t kChunk = 512;       // work items a wave scans and compacts at a time
constexpr uint32_t kStackLds = 16;     // stack entries per lane kept in LDS
constexpr uint32_t kMaxStack = 96;     // deepest triangle BVH accepted (LDS part + spill)
constexpr uint32_t kLeafTrigger = 4;    // lanes that must be waiting on a leaf before a wave takes a leaf step (once no more arrive)
constexpr uint32_t kLeafTriggerHi = 16; // ... or takes it regardless (see the leaf-step rule in k_trace; 24 until the sweep of tools/ab_env.sh: 16 is 1.5-5 % faster on hf10m, soup1m and hf10m at 4 spp alike)
constexpr uint32_t kDone = 0xffffffffu; // `cur`: the ray has no node, leaf or model left (== RTFLAT_INVALID)
constexpr uint32_t kNone = 0xfffffffeu; // `cur` / `postponed`: empty
constexpr uint32_t kRefillThreshold = 16; // refill once this many lanes are idle (or no lane is active): incoherent rays
constexpr uint32_t kRefillCoherent = 56;  // the same for coherent rays, whose lanes finish together (see the refill rule in k_trace)
constexpr uint32_t kMaxRun = 1024;     // most work items one wave is handed (static mode)
constexpr uint32_t kShadeRun = 1024;   // items a wave shades between two appends to the next depth's slot list (one atomic per run)
constexpr uint32_t kFetchShards = 8;   // one work counter per XCD label (blockIdx % 8): 8x the atomic rate of one word, L2-local ranges
constexpr uint32_t kFetchStride = 64;  // counters 256 B apart
constexpr uint32_t kMaxTraceLaunches = 2 * (64 + 2); // two traversal launches per depth, depth <= kMaxDepthSlots
constexpr uint8_t kDead = 0xff;        // path finished, radiance in s0
constexpr uint8_t kDeadZero = 0xfe;    // path finished with exactly zero radiance (primary miss) or pixel outside the image
constexpr uint32_t kMaxShadowedLights = 32; // lights that cast shadow rays: one occlusion bit each in a 32-bit mask per path
constexpr uint32_t kMaxPointLights = 1u << 16;
constexpr size_t kMaxTimedLaunches = 16384; // event pairs kept by the kernel timer between two reads
constexpr uint32_t kMaxDepthSlots = 64;

enum TraceMode
{
  CLOSEST = 0,  // closest hit of the stored ray (depth >= 1, or depth 0 when primary rays are stored)
  SHADOW = 1,   // any-hit toward every directional light from the stored origin
  ANYPROBE = 2, // any-hit of the stored ray (probe entry point)
  PRIMARY = 3,  // ray generation fused in (the primary ray is computed in registers, never stored); dynamic work fetch
  PRIMARY_GROUP = 4, // the same for tiny scenes: static runs, one 8x8 pixel block at a time, no refilling
  SHADOW_PT = 5  // SHADOW, then any-hit toward every point light as well (Compute.metal:238-254): per-lane origin and t range
};
constexpr bool isPrimary(int mode) { return mode == PRIMARY || mode == PRIMARY_GROUP; }
constexpr bool isShadow(int mode) { return mode == SHADOW || mode == SHADOW_PT; }
constexpr bool isClosest(int mode) { return mode == CLOSEST || isPrimary(mode); }

// The packet kernel's copy of a node (device only, 128 B, built from the 64-byte record by k_fatten after every upload,
// device build or refit): the child planes as plain floats, dequantised and rounded outward, grouped so that the planes a ray
// ENTERS through and the planes it LEAVES through are each three 16-byte loads at offsets chosen by the ray direction's signs.
// A wave whose rays share those signs (shadow rays toward a directional light always do; a pixel's camera samples nearly
// always) then tests a child with six FMAs on scalar operands: no byte-to-float conversions (24 per visit, at half the FMA
// rate), no near/far selects, no exponent decoding.
struct FatNode
{
  float plane[6][4]; // [2 * axis + (0: low, 1: high)][child]; an unused slot is inverted (low = FLT_MAX, high = -FLT_MAX)
  uint32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(FatNode) == 128, "two 64-byte lines, one 128-byte L2 line");

struct DeviceScene
{
  const rtflat_model_node* modelNodes;
  const rtflat_node* nodes;
  const FatNode* fat;      // nullptr: not built (the packet kernel then decodes the 64-byte records)
  const rtflat_tri* tris;
  // the packet kernel's per-light triangle operands (k_shadow_operands; nullptr: not built): for directional light l and record
  // r, entry [l * numTris + r] = (cross(d_l, e1), 1 / dot(cross(d_l, e1), e0)) -- the part of the reference's triangle test
  // (CPUScene.cpp:163,166) that is the same for every shadow ray toward that light
  const float4* shadowOperands;
  uint32_t numTris;
  uint32_t numModelNodes;
  float cullRel, cullAbs;
  // shading inputs (rthip_upload_materials); materialsOn == 0: the reference's hard-coded albedo
  const rtflat_shade* shade;
  const rtflat_material* materials;
  const float* texels;
  uint32_t materialsOn;
  uint32_t pointShadows; // point lights are shadowed: their occlusion bits follow the directional lights' in mask[]
};

struct Lights
{
  rthip_dir_light dir[kMaxShadowedLights];
  const rthip_point_light* point; // device array of numPoint entries (any number while they cast no shadow rays)
  uint32_t numDir, numPoint;
};

// device-side tallies; one 64-bit word each, updated once per wave/block at kernel end
struct DeviceCounters
{
  unsigned long long extendPrimary, extendSecondary, shadow, nodesVisited, trisTested, nonFinite, nodesExtend, trisExtend;
  unsigned long long waveNodeSteps, waveLeafSteps; // work counting: node / leaf steps executed by waves (x64 = lane slots offered)
  unsigned long long fetchExtend, fetchShadow;     // work counting: 64-byte scene records fetched (per lane, or per wave by packet launches)
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
  // the current launch batch (set per batch by the host): S sample passes over the local pixels [batchPixelBase,
  // batchPixelBase + batchPixels), a whole number of tiles
  uint32_t batchSamples;    // S
  uint32_t batchLg;         // log2 of the pixels per micro-block (see sampleOf)
  uint32_t batchPixelBase, batchPixels;
  uint64_t batchMagic;      // ceil(2^64 / S) (0 when S == 1): slot -> (micro-block, sample) without a division
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
  uint32_t* mask; // occlusion bits of the previous hit's shadow rays (bit l = light l occluded)
  float4* alb;   // albedo of the previous hit (materials only; without them it is the constant of CPUScene.cpp:245)
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

// Slot layout of a launch batch (S sample passes over a range of the local pixels): the S samples of a micro-block of
// 2^lg consecutive pixels are adjacent,
//     slot = ((j >> lg) * S + s) << lg | (j & (2^lg - 1))      j = pixel index within the batch, s = sample index,
// and the host picks 2^lg = 64 / S (S >= 64: one pixel), so a wave's 64 consecutive slots are as few pixels with as many
// samples each as the batch allows: rays that walk the same BVH nodes and hit the same triangles.  Node fetches are bound
// by the rate at which a CU pulls DISTINCT 64-byte lines out of L2 (tools/ubench/gather64: 2.5 cycles per line, 11 when
// it misses L2; lanes that want a line another lane of the wave just fetched cost 0.4), so the sample-major order of
// round 1 (slot = s * P + i, one sample per wave) paid for every line S times.
// slot -> (sample index within the batch, local pixel index of the frame)
__device__ __forceinline__ void sampleOf(const FrameGeom& g, uint32_t slot, uint32_t& s, uint32_t& i)
{
  const uint32_t q = slot >> g.batchLg;
  const uint32_t mb = g.batchMagic ? (uint32_t)__umul64hi((uint64_t)q, g.batchMagic) : q; // q / S, exact for every 32-bit q
  s = q - mb * g.batchSamples;
  i = g.batchPixelBase + ((mb << g.batchLg) | (slot & ((1u << g.batchLg) - 1u)));
}
// (sample index within the batch, pixel index within the batch) -> slot
__device__ __forceinline__ uint32_t slotOf(const FrameGeom& g, uint32_t s, uint32_t j)
{
  return (((j >> g.batchLg) * g.batchSamples + s) << g.batchLg) | (j & ((1u << g.batchLg) - 1u));
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

__device__ __forceinline__ vec3 referenceAlbedo() { return vec3{0.0f, 1.0f, 0.0f}; } // CPUScene.cpp:245

// Adds the direct light of the hit a path has just left (CPUScene.cpp:31-49, lights in that order): directional lights
// unless their shadow ray was occluded, then the point lights -- unshadowed as on the reference's CPU path, or, with
// pointShadows, skipped when their bit (after the directional lights') is set.  `o` is the hit position, nPrev its face
// normal, albedo the hit's.  One function for k_shade and the shadow kernel's final fold, so both produce the same bits.
__device__ __forceinline__ vec3 foldLights(const Lights* lights, uint32_t occluded, vec3 o, vec3 nPrev, vec3 radiance, vec3 albedo, uint32_t pointShadows)
{
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
    if (pointShadows && ((occluded >> (lights->numDir + l)) & 1u))
      continue;
    const vec3 lightDir = vec3{L.position[0], L.position[1], L.position[2]} - o;
    const float dist = rt::length(lightDir);
    const float illuminance = rt::maxRef(1 - dist / L.attenuation, 0.0f);
    radiance = radiance + illuminance * rtd::brdfEvaluate(albedo, nPrev, lightDir, vec3{L.color[0], L.color[1], L.color[2]});
  }
  return radiance;
}
} // namespace rtk
