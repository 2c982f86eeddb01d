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
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: the library is bound at run time (see RcclApi)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <rocprim/rocprim.hpp> // device radix sort of the Morton codes (rthip_upload_scene_build)

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

#include "rt_kernel_common.h"
#include "rt_lbvh.h"
#include "rt_probes.h"
#include "rt_shade.h"
#include "rt_trace.h"
#include "rt_packet.h"

using namespace rtk;

namespace
{
template <typename T>
struct DevBuf
{
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept
  {
    if (this != &o)
    {
      release();
      p = o.p, n = o.n;
      o.p = nullptr, o.n = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); } // every early return of a probe / untile call frees what it had allocated
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

// Lends memory the context does not own (the static scratch array of rthip_trace_rays_static) to one of its buffers for
// the duration of a call; the buffer's own pointer comes back when the guard goes out of scope, so nothing borrowed is
// ever freed.
template <typename T>
struct Borrow
{
  DevBuf<T>& b;
  T* p;
  size_t n;
  Borrow(DevBuf<T>& buf, void* mem, size_t count) : b(buf), p(buf.p), n(buf.n) { b.p = static_cast<T*>(mem), b.n = count; }
  ~Borrow() { b.p = p, b.n = n; }
  Borrow(const Borrow&) = delete;
  Borrow& operator=(const Borrow&) = delete;
};
} // namespace

// ---- static scratch of the allocation-free entry point (rthip_trace_rays_static) -------------------------------------
// One array per GPU the library runs on, held until the process ends.  Layout (all offsets multiples of 256 B):
//   [rayO | rayD | hit | live | batch counters + work-fetch counters | stack spill]
// Host code never uses the array's name as an address: rthip_create resolves it once per device with
// hipGetSymbolAddress (that device current) and every launch / memset takes the resolved pointer.
namespace
{
constexpr uint32_t kStaticMaxRays = RTHIP_STATIC_MAX_RAYS;
constexpr uint32_t kStaticSpillCap = rtk::kMaxStack - rtk::kStackLds; // deepest spill any accepted BVH can need
constexpr size_t kStaticCounterWords = ((sizeof(rtk::DeviceCounters) + 255) / 256) * 64;
constexpr size_t kStaticFetchWords = kStaticCounterWords + 2 * (size_t)rtk::kFetchShards * rtk::kFetchStride; // two traversal launches
constexpr size_t kStaticOffRayO = 0;
constexpr size_t kStaticOffRayD = kStaticOffRayO + (size_t)kStaticMaxRays * sizeof(float4);
constexpr size_t kStaticOffHit = kStaticOffRayD + (size_t)kStaticMaxRays * sizeof(float4);
constexpr size_t kStaticOffLive = kStaticOffHit + (size_t)kStaticMaxRays * sizeof(uint2);
constexpr size_t kStaticOffFetch = kStaticOffLive + (((size_t)kStaticMaxRays + 64 + 255) / 256) * 256;
constexpr size_t kStaticOffSpill = kStaticOffFetch + ((kStaticFetchWords * 4 + 255) / 256) * 256;
// a launch over <= kStaticMaxRays items uses 64-item chunks, i.e. at most kStaticMaxRays/64 waves = kStaticMaxRays lanes
constexpr size_t kStaticSpillWords = (size_t)kStaticSpillCap * kStaticMaxRays;
constexpr size_t kStaticBytes = kStaticOffSpill + kStaticSpillWords * 4;
static_assert(kStaticBytes <= (8u << 20), "the static scratch is meant to stay modest (8 MiB)");
constexpr int kMaxDevices = 64;
} // namespace
static __device__ __align__(256) unsigned char scratch[kStaticBytes];
namespace
{
unsigned char* g_scratchPtr[kMaxDevices] = {}; // device address of `scratch` on each GPU (each GPU holds its own copy)
}

struct rthip_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr; // second stream for the shadow launch of a depth (see renderBatch)
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  hipEvent_t evA = nullptr, evB = nullptr, evK0 = nullptr, evK1 = nullptr;
  int numCUs = 256;
  uint32_t traceBlocks = 2048; // persistent grid of the list launches (depth >= 1); dense launches use traceBlocksMax
  uint32_t fetchCursor = 0;    // next unused counter set of the current batch
  // scene
  bool haveScene = false;
  rtflat_info info{};
  DevBuf<rtflat_model_node> modelNodes;
  DevBuf<rtflat_node> nodes;
  DevBuf<rtflat_tri> tris;
  DevBuf<FatNode> fatNodes;      // the packet kernel's copy of `nodes` (k_fatten), rebuilt whenever the nodes change
  DevBuf<float4> shadowOperands; // the packet kernel's per-light triangle operands (k_shadow_operands), rebuilt with the fat nodes
  bool useOperands = true;       // (RTHIP_SHADOW_OPERANDS=0: the packet kernel computes them per ray, the A/B baseline)
  bool useFat = true;            // (RTHIP_FAT=0: the packet kernel decodes the 64-byte records, the A/B baseline)
  DevBuf<Lights> lights;
  DevBuf<rthip_point_light> pointLights;
  Lights hostLights{};
  // shading inputs (rthip_upload_materials): per-triangle texture coordinates + material index, materials, texel pool
  DevBuf<rtflat_shade> shade;
  DevBuf<rtflat_material> materials;
  DevBuf<float> texels;
  bool materialsOn = false;
  bool pointShadows = false; // rthip_set_point_shadows
  // frame
  bool haveFrame = false;
  rthip_camera camera{};
  FrameGeom geom{};
  uint32_t maxDepth = 12, aovDepths = 0;
  uint64_t pathCap = 1ull << 28; // most paths (slots) a launch batch may hold: about 80 B of state each, per slot set (2 x 22 GB at most)
  int groupLg = -1;              // log2 pixels per micro-block of the slot layout (-1: 64 / samples of the batch, see sampleOf)
  std::vector<uint32_t> hostTileIds, hostTileXY;
  DevBuf<uint32_t> tileIds;
  DevBuf<CameraConsts> camConsts;
  DevBuf<float4> rayO, rayD, s0, alb;
  DevBuf<float2> s1;
  DevBuf<float> nz;
  DevBuf<uint2> hit;
  DevBuf<uint8_t> live, anyOut;
  DevBuf<uint32_t> mask;
  DevBuf<float> acc, img, rowScratch;
  DevBuf<uint32_t> aovTri;
  DevBuf<uint8_t> aovShadow;
  DevBuf<uint32_t> spill;
  DevBuf<uint32_t> fetch;        // [batch counters (kCounterWords) | work-fetch counters]: everything a batch zeroes, one memset
  DevBuf<uint32_t> spill2;       // stack spill area of launches on stream2
  DevBuf<uint32_t> listA, listB; // slot lists of the paths alive at even / odd depths (one entry per slot of the batch at most)
  uint32_t spillCap = 1;         // per-lane spill entries the uploaded BVH can need
  uint32_t tuneBlocks = 6;       // resident blocks per CU of k_trace's persistent grid (what 24 KB of LDS per block admits) ...
  uint32_t tuneBlocksLow = 3;    // ... and of its launches over slot lists (depth >= 1), see launchTrace (RTHIP_LIST_BLOCKS)
  uint32_t tuneBlocksTail = 3;   // ... and of the list launches at depth >= 2 (RTHIP_TAIL_BLOCKS): short lists, latency-bound launches
  uint32_t traceBlocksTail = 768;
  uint32_t shadeTailBlocks = 128; // k_shade grid at depth >= 2, blocks per CU (RTHIP_SHADE_TAIL)
  int buildQuality = 0;          // device BVH build: 0 = Morton LBVH (default), 1 = PLOC over the Morton order (rthip_set_build_quality, RTHIP_BUILD_QUALITY)
  bool buildNormalAware = true;   // device build: Morton bits dealt by cell extents that take the surfaces' orientation into account (RTHIP_BUILD_NORMALS=0: scene extents only)
  uint32_t buildRotate = 2;      // device build: rounds of bottom-up tree rotations, done in the refit passes (RTHIP_BUILD_ROTATE=0: the radix tree as it is; hf10m -0.8 %, soup1m -2.9 %, bunny -1.2 % with two rounds, four add nothing)
  bool buildDfsOrder = true;      // device build: nodes re-ordered depth-first after the collapse (RTHIP_BUILD_DFS=0: left breadth-first)
  float buildTraversalCost = 1.0f; // SAH cost of a node visit relative to a triangle test in the device build's leaf rule (RTHIP_BUILD_TCOST)
  uint32_t plocRadius = 16;      // neighbours searched on either side (RTHIP_PLOC_RADIUS)
  bool splitCalls = true;        // a call that fits one batch is still cut in two, one half per slot (RTHIP_NO_SPLIT=1: not)
  uint32_t traceBlocksMax = 2048; // numCUs * tuneBlocks: what the spill areas are sized for
  uint32_t overlapShadow = 1;    // run a depth's shadow and closest-hit launches on two streams (their tails overlap)
  uint32_t tuneSteal = kFetchShards; // spans a wave may pull chunks from: its own first, then the others' (1 = its own only).  Off in round 1 (+4 % on the soup
                                     // then); with lists and the measured grid it is -5 % on hf10m at 4 spp and neutral on hf10m / soup1m (tools/ab_env.sh)
  uint32_t tuneRefill = kRefillThreshold, tuneLeaf = kLeafTrigger, tuneLeafHi = kLeafTriggerHi;
  bool tuneLeafSet = false;      // RTHIP_TUNE named the leaf trigger: no per-mode choice on top of it
  uint32_t tuneRefillShadow = kRefillThreshold, tuneRefillPrimary = 32; // the low thresholds of shadow / dynamic primary launches (RTHIP_TUNE fields 8, 9)
  uint32_t tuneRefillHi = kRefillCoherent;                              // the high threshold (field 10)
  // Which refill thresholds the traversal launches use: low (incoherent rays) or high (coherent rays), see k_trace.  -1 in
  // refillWanted = measure it (calibrateRefill) whenever the scene or the frame's shape changes; 0 / 1 = fixed by the caller.
  int refillWanted = -1, refillMode = 0;
  uint64_t refillKey = 0;        // what the current refillMode was measured for
  uint32_t sceneEpoch = 0;       // counts scene uploads
  float refillMs[4] = {0, 0, 0, 0}; // the calibration timings (low, high, packets, packets for camera rays only), for inspection
  uint32_t packetBlocks = 4;     // resident blocks per CU of the packet kernel (RTHIP_PACKET_BLOCKS): 4 x 4 waves leave room for the other batch's
                                 // list launches beside it (hf10m 157.6 -> 156.4 ms in two interleaved A/B runs; 6, 8, 12: no better)
  bool packetShadow = true;      // (RTHIP_PACKET_SHADOW=0: packets for camera rays only)
  bool splitLights = true;       // (RTHIP_SPLIT_LIGHTS=0: deep shadow launches test a path's lights in one lane, one after the other)
  bool closestLow = true;        // (RTHIP_CLOSEST_LOW=0: the first bounce rays use the high threshold)
  bool deepLow = true;           // bounce rays and the shadow rays of their hits are scattered whatever the camera rays were: low thresholds (RTHIP_DEEP_LOW=0: off; hf10m -2 %)
  uint32_t calibLog2 = 24;       // paths in the calibration sample (RTHIP_CALIB_LOG2): fewer, and launches are tail-bound, which hides the difference
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
    DevBuf<float4> rayO, rayD, s0, alb;
    DevBuf<float2> s1;
    DevBuf<float> nz;
    DevBuf<uint2> hit;
    DevBuf<uint8_t> live;
    DevBuf<uint32_t> mask, spill, spill2, fetch, listA, listB;
  } alt;
  hipEvent_t evDone = nullptr;   // this slot's "batch finished" event (recorded after every batch)
  hipEvent_t evFrame = nullptr;  // fork point: the second slot's stream waits for the frame prologue on the main stream
  hipEvent_t lastAcc = nullptr;  // evDone of the most recent batch, whichever slot ran it: the next accumulate waits for it
  hipStream_t lastAccStream = nullptr;
  bool altPending = false;       // work may be in flight on the second slot's stream that the main stream has not joined
  uint32_t frameParity = 0;      // camera constants and counter totals are double-buffered per frame (frames overlap)
  uint32_t pipelineBatches = 2;  // 2: batches alternate between the two slots (two in flight); 0 = one slot only
  uint32_t batchToggle = 0;      // which slot the next batch goes to (alternates across calls and frames)
  bool framePipelined = false;   // decided by rthip_begin (both slots sized for the frame's batches)
  // dynamic tile ownership (rthip_begin_dynamic / rthip_enqueue_tiles): the frame's tiles arrive while it renders
  bool dynamic = false;
  uint32_t dynCapacity = 0, dynCount = 0; // most tiles this frame may take / taken so far
  std::vector<uint32_t> dynIds;           // their ids, in arrival order (== the order of their blocks in the accumulator)
  DevBuf<uint32_t> dynTable;              // 2 x capacity tile origins: one table per frame parity (two frames may be in flight)
  uint32_t* dynStage = nullptr;           // pinned host staging of the same shape (the uploads are truly asynchronous)
  size_t dynStageWords = 0;
  static constexpr uint32_t kDynEvents = 8;
  hipEvent_t dynEvents[kDynEvents] = {};  // one per rthip_enqueue_tiles call, round robin: rthip_wait_tiles' back-pressure
  uint64_t dynIssued = 0, dynRetired = 0;
  DevBuf<uint32_t> gatherHdr;             // rthip_gather_tiles: [count | tile ids] per rank
  // multi-GPU assembly (rthip_comm_init / rthip_gather)
  ncclComm_t comm = nullptr;
  int commRank = 0, commSize = 0;
  DevBuf<float> gatherStage;     // this rank's accumulator padded to the common per-rank size
};

// RCCL is bound at run time with dlopen/dlsym instead of being linked: a host process that already carries a copy of RCCL
// (PyTorch ships one) then shares that one copy, and a process that never gathers never loads it.
namespace
{
struct RcclApi
{
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGetErrorString) errorString = nullptr;
  bool ok = false;
};
char g_commError[256] = "";
RcclApi& rccl()
{
  static RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
        break;
    if (!a.handle)
    {
      snprintf(g_commError, sizeof(g_commError), "rthip: cannot load RCCL (%s)", dlerror());
      return a;
    }
    a.getUniqueId = reinterpret_cast<decltype(a.getUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
    a.commInitRank = reinterpret_cast<decltype(a.commInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
    a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
    a.groupStart = reinterpret_cast<decltype(a.groupStart)>(dlsym(a.handle, "ncclGroupStart"));
    a.groupEnd = reinterpret_cast<decltype(a.groupEnd)>(dlsym(a.handle, "ncclGroupEnd"));
    a.send = reinterpret_cast<decltype(a.send)>(dlsym(a.handle, "ncclSend"));
    a.recv = reinterpret_cast<decltype(a.recv)>(dlsym(a.handle, "ncclRecv"));
    a.errorString = reinterpret_cast<decltype(a.errorString)>(dlsym(a.handle, "ncclGetErrorString"));
    a.ok = a.getUniqueId && a.commInitRank && a.commDestroy && a.groupStart && a.groupEnd && a.send && a.recv && a.errorString;
    if (!a.ok)
      snprintf(g_commError, sizeof(g_commError), "rthip: the RCCL library lacks an entry point");
    return a;
  }();
  return api;
}
int commFail(ncclResult_t r, const char* what)
{
  snprintf(g_commError, sizeof(g_commError), "rthip: %s: %s", what, rccl().errorString ? rccl().errorString(r) : "RCCL error");
  return RTHIP_E_COMM;
}
#define NCCLCHK(x, what)             \
  do                                 \
  {                                  \
    ncclResult_t r_ = (x);           \
    if (r_ != ncclSuccess)           \
      return commFail(r_, what);     \
  } while (0)
} // namespace

namespace
{
DeviceScene devScene(const rthip_ctx* c)
{
  DeviceScene s;
  s.modelNodes = c->modelNodes.p;
  s.nodes = c->nodes.p;
  s.tris = c->tris.p;
  s.fat = (c->useFat && c->fatNodes.n >= c->info.numNodes) ? c->fatNodes.p : nullptr;
  s.numTris = c->info.numTris;
  s.shadowOperands = (c->useOperands && c->hostLights.numDir && c->shadowOperands.n >= (size_t)c->hostLights.numDir * c->info.numTris) ? c->shadowOperands.p : nullptr;
  s.numModelNodes = c->info.numModelNodes;
  s.cullRel = c->info.cullRel;
  s.cullAbs = c->info.cullAbs;
  s.shade = c->shade.p;
  s.materials = c->materials.p;
  s.texels = c->texels.p;
  s.materialsOn = c->materialsOn ? 1u : 0u;
  s.pointShadows = (c->pointShadows && c->hostLights.numPoint) ? 1u : 0u;
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
  std::swap(c->hit, a.hit), std::swap(c->live, a.live), std::swap(c->mask, a.mask), std::swap(c->alb, a.alb);
  std::swap(c->spill, a.spill), std::swap(c->spill2, a.spill2), std::swap(c->fetch, a.fetch);
  std::swap(c->listA, a.listA), std::swap(c->listB, a.listB);
}

PathBuffers pathBuffers(const rthip_ctx* c)
{
  PathBuffers pb;
  pb.rayO = c->rayO.p, pb.rayD = c->rayD.p, pb.hit = c->hit.p, pb.s0 = c->s0.p, pb.s1 = c->s1.p, pb.nz = c->nz.p, pb.live = c->live.p, pb.mask = c->mask.p;
  pb.alb = c->alb.p;
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
  const uint32_t waves = c->traceBlocksMax * kWavesPerBlock;
  uint32_t cl = (total + waves * 8u - 1) / (waves * 8u);
  cl = ((cl + 63u) / 64u) * 64u;
  chunkLen = std::max(64u, std::min(cl, 4u * kChunk)); // a fetched span is scanned kChunk items at a time
  const uint32_t wavesNeeded = (total + chunkLen - 1) / chunkLen;
  blocks = std::max(kFetchShards, std::min(c->traceBlocksMax, (wavesNeeded + kWavesPerBlock - 1) / kWavesPerBlock));
}

int ensurePathBuffers(rthip_ctx* c, size_t numSlots)
{
  int rc;
  if ((rc = c->rayO.ensure(numSlots)) || (rc = c->rayD.ensure(numSlots)) || (rc = c->s0.ensure(numSlots)) || (rc = c->s1.ensure(numSlots)) || (rc = c->nz.ensure(numSlots)) ||
      (rc = c->hit.ensure(numSlots)) || (rc = c->live.ensure(numSlots + 64)) || (rc = c->mask.ensure(numSlots)))
    return rc;
  if (c->materialsOn && (rc = c->alb.ensure(numSlots)))
    return rc;
  uint32_t blocks, runLen;
  staticGeometry((uint32_t)numSlots, blocks, runLen);
  blocks = std::max(blocks, c->traceBlocksMax);
  if ((rc = c->spill.ensure((size_t)c->spillCap * blocks * kBlock)) || (rc = c->spill2.ensure((size_t)c->spillCap * blocks * kBlock)))
    return rc;
  if ((rc = c->fetch.ensure(kCounterWords + (size_t)kMaxTraceLaunches * kFetchShards * kFetchStride)))
    return rc;
  if ((rc = c->listA.ensure(numSlots)) || (rc = c->listB.ensure(numSlots)))
    return rc;
  return 0;
}

// Deep shadow launches (depth >= 2: scattered rays, few of them) make one item of every (path, light) pair when the scene has two
// or four directional lights: log2 of that number, 0 = a lane tests its path's lights one after the other (k_trace, splitLg)
uint32_t shadowSplitLg(const rthip_ctx* c, uint32_t depth)
{
  if (!c->splitLights || depth < 2)
    return 0;
  return c->hostLights.numDir == 2 ? 1u : (c->hostLights.numDir == 4 ? 2u : 0u);
}

// list == nullptr: dense launch over all numSlots slots; otherwise a launch over the device-side slot list of `depth`,
// whose length (counters->listCount[depth]) only the device knows: the whole persistent grid is launched and sizes its
// chunks from that length
template <int MODE>
int launchTrace(rthip_ctx* c, uint32_t numSlots, uint32_t depth, uint32_t sampBegin, float tmin, float tmax, const uint32_t* list = nullptr,
                bool foldFinal = false, bool sideStream = false, const FrameGeom* batchGeom = nullptr)
{
  TraceArgs a;
  a.scene = devScene(c);
  a.pb = pathBuffers(c);
  a.lights = c->lights.p;
  a.geom = batchGeom ? *batchGeom : c->geom;
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
  a.refillAt = (c->refillMode >= 1 && !(c->deepLow && (depth >= 2 || (MODE == CLOSEST && c->closestLow)))) ? c->tuneRefillHi : (isShadow(MODE) ? c->tuneRefillShadow : (MODE == PRIMARY ? c->tuneRefillPrimary : c->tuneRefill));
  // scenes whose camera rays and first shadow rays go as packets (mode 2: coherent rays) let a few more lanes gather before a
  // leaf step: hf10m 149.1 / 149.3 -> 147.7 / 147.5 ms (tools/ab_env.sh, interleaved); the soup (mode 0 / 3) loses 3.6 % with it
  a.leafTrigger = (c->refillMode == 2 && !c->tuneLeafSet) ? std::max(c->tuneLeaf, 8u) : c->tuneLeaf;
  a.leafTriggerHi = c->tuneLeafHi;
  a.stealSpans = c->tuneSteal;
  a.splitLg = (MODE == SHADOW && list && !foldFinal) ? shadowSplitLg(c, depth) : 0u;
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
    if (list)
    {
      blocks = depth >= 2 ? c->traceBlocksTail : c->traceBlocks;
      a.chunkLen = 0;
    }
    else
      dynamicGeometry(c, numSlots, blocks, a.chunkLen);
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

// packet traversal needs its wave stack (64 entries, up to 3 pushes per level) to hold the deepest chain of the uploaded BVH
bool packetsPossible(const rthip_ctx* c) { return 3u * c->info.maxDepthBlas + 2u <= kPacketStack; }

// (re)builds the packet kernel's copy of the nodes; called wherever the node records change (upload, device build, refit)
int fattenNodes(rthip_ctx* c)
{
  if (!c->useFat || !packetsPossible(c) || c->info.numNodes == 0)
  {
    c->fatNodes.release();
    c->shadowOperands.release();
    return 0;
  }
  if (int rc = c->fatNodes.ensure(c->info.numNodes))
    return rc;
  hipLaunchKernelGGL(k_fatten, dim3((c->info.numNodes + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->nodes.p, c->info.numNodes, c->fatNodes.p);
  HIPCHK(hipGetLastError());
  // ... and its per-light triangle operands, for up to four directional lights (16 bytes per light and record)
  if (c->useOperands && c->hostLights.numDir >= 1 && c->hostLights.numDir <= 4 && c->info.numTris)
  {
    if (int rc = c->shadowOperands.ensure((size_t)c->hostLights.numDir * c->info.numTris))
      return rc;
    hipLaunchKernelGGL(k_shadow_operands, dim3((c->info.numTris + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->tris.p, c->info.numTris, c->lights.p,
                       c->hostLights.numDir, c->shadowOperands.p);
    HIPCHK(hipGetLastError());
  }
  else
    c->shadowOperands.release();
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// the packet kernel (rt_packet.h) in place of k_trace<PRIMARY> / k_trace<SHADOW>: same arguments, same results
template <int MODE>
int launchTracePacket(rthip_ctx* c, uint32_t numSlots, uint32_t depth, uint32_t sampBegin, float tmin, float tmax, const uint32_t* list, bool foldFinal, bool sideStream,
                      const FrameGeom* batchGeom)
{
  TraceArgs a{};
  a.scene = devScene(c);
  a.pb = pathBuffers(c);
  a.lights = c->lights.p;
  a.geom = batchGeom ? *batchGeom : c->geom;
  a.cam = frameCamera(c);
  a.sampBegin = sampBegin;
  a.numSlots = numSlots;
  a.list = list;
  a.listCount = list ? &batchCounters(c)->listCount[depth] : nullptr;
  a.depth = depth;
  a.counters = batchCounters(c);
  a.tmin = tmin;
  a.tmax = tmax;
  a.foldFinal = foldFinal ? 1u : 0u;
  a.aovDepths = c->aovDepths;
  a.aovShadow = c->aovShadow.p;
  if (c->fetchCursor >= kMaxTraceLaunches)
    return RTHIP_E_LIMIT;
  a.fetch = fetchBase(c) + (size_t)(c->fetchCursor++) * kFetchShards * kFetchStride;
  const uint32_t blocks = (uint32_t)c->numCUs * c->packetBlocks; // 8 blocks x 4 waves: the kernel keeps within 64 VGPRs and uses no LDS
  hipStream_t stream = sideStream ? c->stream2 : c->stream;
  KernelTimer kt(c, isClosest(MODE) ? 1 : 3, stream);
  if (c->countWork)
    hipLaunchKernelGGL((k_trace_packet<MODE, true>), dim3(blocks), dim3(kBlock), 0, stream, a);
  else
    hipLaunchKernelGGL((k_trace_packet<MODE, false>), dim3(blocks), dim3(kBlock), 0, stream, a);
  return (int)hipGetLastError();
}

// One batch of `count` sample passes starting at sampBegin.  Nothing here waits for the device: from depth 1 on every
// launch works on the slot list the previous depth's k_shade wrote, whose length stays in device memory, so all depths up
// to the cap are queued unconditionally (a launch whose list is empty returns at once) and batches and frames can overlap.
int renderBatch(rthip_ctx* c, uint32_t sampBegin, uint32_t count, uint32_t tileBegin, uint32_t tileCount, bool fresh, hipEvent_t accumulateAfter = nullptr,
                bool calibrating = false)
{
  const uint32_t P = c->geom.numLocalPixels; // of the frame (AOV arrays are indexed by it)
  const uint32_t perTile = c->geom.tileSize * c->geom.tileSize;
  FrameGeom g = c->geom; // + this batch's pixel range and slot layout (see sampleOf)
  g.batchSamples = count;
  g.batchPixelBase = tileBegin * perTile;
  g.batchPixels = tileCount * perTile;
  g.batchMagic = count > 1 ? (~0ull / count) + 1ull : 0ull; // ceil(2^64 / count) for count >= 2
  {
    uint32_t lg = 6;
    for (uint32_t sN = count; sN > 1 && lg > 0; sN >>= 1)
      lg--;
    if (c->groupLg >= 0)
      lg = (uint32_t)std::min(c->groupLg, 6);
    g.batchLg = lg;
  }
  const uint32_t numSlots = g.batchPixels * count;
  // every kernel below indexes the path arrays by slot without bounds checks: the batch must fit what rthip_begin sized
  if ((uint64_t)g.batchPixels * count > std::min<uint64_t>(std::min<uint64_t>(c->rayO.n, c->hit.n), std::min<uint64_t>(c->listA.n, c->listB.n)) ||
      (c->materialsOn && (uint64_t)g.batchPixels * count > c->alb.n) || tileBegin + tileCount > c->geom.numLocalTiles)
    return RTHIP_E_LIMIT;
  // k_shade's grid: one full-length run per wave as far as 128 blocks per CU go -- sixteen times what is resident, so that blocks whose
  // runs were cheap (misses) make room for the next ones instead of the whole launch waiting on the dearest (8 per CU, one
  // resident set: hf10m 164.1 ms; 16 / 32 / 128: 162.7 / 161.3 / 160.2 ms; cube -1.3 %, bunny +2 %, soup1m unchanged)
  const uint32_t shadeBlocks = std::max(1u, std::min((numSlots + kShadeRun * kWavesPerBlock - 1) / (kShadeRun * kWavesPerBlock), (uint32_t)c->numCUs * 128u));
  HIPCHK(hipStreamWaitEvent(c->stream, c->evFrame, 0)); // the frame prologue (k_camera) may have run on the other slot's stream
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
    hipLaunchKernelGGL(k_generate, dim3((numSlots + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, g, frameCamera(c), pathBuffers(c), sampBegin,
                       numSlots);
    HIPCHK(hipGetLastError());
  }
  // shadow rays: toward every directional light (CPUScene.cpp:31-38) and, when asked for, every point light
  const bool shadowPoints = c->pointShadows && c->hostLights.numPoint > 0;
  const bool haveDirLights = c->hostLights.numDir > 0 || shadowPoints;
  ShadeArgs sa;
  sa.g = g;
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
  for (uint32_t d = 0;; ++d)
  {
    const bool last = d == c->maxDepth;
    // list of the paths alive at depth d (written by k_shade(d-1)); depth 0 is every slot
    const uint32_t* list = d == 0 ? nullptr : ((d & 1u) ? c->listB.p : c->listA.p);
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
    const bool packets = c->refillMode >= 2 && packetsPossible(c); // 2: camera rays and their hits' shadow rays, 3: camera rays only
    if (d > 0 && haveDirLights &&
        (rc = (packets && d == 1 && c->refillMode == 2 && c->packetShadow) // the camera rays' hits: still one pixel per wave
                  ? (shadowPoints ? launchTracePacket<SHADOW_PT>(c, numSlots, d, sampBegin, tmin, tmax, list, last, side, &g)
                                  : launchTracePacket<SHADOW>(c, numSlots, d, sampBegin, tmin, tmax, list, last, side, &g))
                  : (shadowPoints ? launchTrace<SHADOW_PT>(c, numSlots, d, sampBegin, tmin, tmax, list, last, side, &g)
                                  : launchTrace<SHADOW>(c, numSlots, d, sampBegin, tmin, tmax, list, last, side, &g))))
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
        rc = launchTrace<PRIMARY_GROUP>(c, numSlots, 0, sampBegin, tmin, tmax, nullptr, false, false, &g);
      else if (d == 0 && fused && packets)
        rc = launchTracePacket<PRIMARY>(c, numSlots, 0, sampBegin, tmin, tmax, nullptr, false, false, &g);
      else if (d == 0 && fused)
        rc = launchTrace<PRIMARY>(c, numSlots, 0, sampBegin, tmin, tmax, nullptr, false, false, &g);
      else
        rc = launchTrace<CLOSEST>(c, numSlots, d, sampBegin, tmin, tmax, list, false, false, &g);
      if (rc)
        return rc;
    }
    if (side)
      HIPCHK(hipStreamWaitEvent(c->stream, c->evJoin, 0));
    sa.depth = d;
    sa.finalFold = last ? 1u : 0u;
    // (the shadow launch of this depth, above: per-lane over the list, not the last depth's fold, no point-light shadow rays)
    sa.maskSplit = (d >= 2 && !last && haveDirLights && !shadowPoints && shadowSplitLg(c, d)) ? c->hostLights.numDir : 0u;
    sa.list = list;
    sa.listNext = last ? nullptr : listNext;
    {
      KernelTimer kt(c, 2);
      // from depth 2 on the lists are short (hf10m: 4 % of the paths): a grid sized for the batch would be thousands of blocks
      // that start only to find nothing to do
      const uint32_t shadeGrid = d >= 2 ? std::min(shadeBlocks, (uint32_t)c->numCUs * c->shadeTailBlocks) : shadeBlocks;
      if (c->materialsOn)
      {
        if (d > 0)
          hipLaunchKernelGGL((k_shade<true, true>), dim3(shadeGrid), dim3(kBlock), 0, c->stream, sa);
        else
          hipLaunchKernelGGL((k_shade<false, true>), dim3(shadeBlocks), dim3(kBlock), 0, c->stream, sa);
      }
      else if (d > 0)
        hipLaunchKernelGGL((k_shade<true, false>), dim3(shadeGrid), dim3(kBlock), 0, c->stream, sa);
      else
        hipLaunchKernelGGL((k_shade<false, false>), dim3(shadeBlocks), dim3(kBlock), 0, c->stream, sa);
      HIPCHK(hipGetLastError());
    }
    if (last)
      break;
  }
  if (calibrating) // a timing run (calibrateRefill): the paths are traced and shaded, nothing of them reaches the frame
    return 0;
  if (accumulateAfter) // the previous batch ran on the other slot: per-pixel accumulation stays in sample order
    HIPCHK(hipStreamWaitEvent(c->stream, accumulateAfter, 0));
  {
    KernelTimer kt(c, 4);
    hipLaunchKernelGGL(k_accumulate, dim3((g.batchPixels + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, g, pathBuffers(c), c->acc.p, count, fresh ? 1u : 0u);
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

// How a render call's passes are cut into launch batches.  A batch is (a range of the local tiles) x (a range of the
// call's sample passes): all samples of a pixel that the call renders stay together as long as one tile's worth of them
// fits the path cap -- that is what makes a wave's rays coherent (see sampleOf) -- and the image is cut by tiles.
struct BatchPlan
{
  uint32_t samples; // sample passes per batch
  uint32_t tiles;   // tiles per batch
};
BatchPlan planBatches(const rthip_ctx* c, uint32_t count, const FrameGeom& g, uint32_t aovDepths, bool pipelined)
{
  BatchPlan bp;
  if (aovDepths)
  {
    bp.samples = std::min(count, 4u); // AOV arrays are [sample][depth][pixel of the frame]
    bp.tiles = g.numLocalTiles;
    return bp;
  }
  const uint64_t perTile = (uint64_t)g.tileSize * g.tileSize;
  bp.samples = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(count, c->pathCap / perTile));
  bp.tiles = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(g.numLocalTiles, c->pathCap / (perTile * bp.samples)));
  // a call that fits one batch but is big enough to fill the GPU twice over is cut in two, one half per slot
  if (pipelined && c->splitCalls && bp.tiles == g.numLocalTiles && bp.samples == count && g.numLocalTiles >= 2 && perTile * g.numLocalTiles * count >= (1ull << 23))
    bp.tiles = (g.numLocalTiles + 1) / 2;
  return bp;
}

// Decides how the traversal launches treat a wave's 64 rays -- low refill thresholds (incoherent rays), high thresholds
// (coherent rays), packet traversal of camera rays and their hits' shadow rays, or packets for the camera rays only -- by
// measuring, in the regime frames run in: TWO sample batches at a time, one on each slot (two runs of tiles from the middle of
// the local tile list, the sample passes one call would bring, about 16M paths each, as far as the path buffers allow), so that
// every candidate is timed with another batch beside it as in a pipelined frame (round 2 timed one batch alone, which is blind
// to what overlap changes: the grid a setting leaves to the other batch, mode 3's win at few samples per pixel).  A frame that
// is not pipelined (one slot) is measured on one slot.  The first candidate runs once untimed (caches warm; its duration sets how many sample pairs a measurement queues), nothing reaches
// the accumulator, the AOVs or the counters.  Called by rthip_begin when the scene, its shading switches or the frame's shape
// have changed since the last measurement: a few batches and one host synchronisation, then frames queue without waiting.
uint64_t calibrationKey(const rthip_ctx* c)
{
  // FNV-1a over the fields themselves (round 2 XORed shifted fields, which overlap for large frames)
  const uint32_t fields[] = {c->sceneEpoch, c->geom.width, c->geom.height, c->geom.numSamples, c->maxDepth, c->materialsOn ? 1u : 0u,
                             (c->pointShadows && c->hostLights.numPoint) ? 1u : 0u, c->framePipelined ? 1u : 0u, c->geom.tileSize};
  uint64_t h = 1469598103934665603ull;
  for (uint32_t f : fields)
    for (int b = 0; b < 4; ++b)
    {
      h ^= (f >> (8 * b)) & 0xffu;
      h *= 1099511628211ull;
    }
  return h | 1ull; // never 0 (0 = nothing measured yet); the tile set is left out on purpose: a rank handed other tiles of the same frame keeps its choice
}

int calibrateRefill(rthip_ctx* c)
{
  const uint64_t key = calibrationKey(c);
  if (c->refillWanted >= 0)
  {
    c->refillMode = c->refillWanted;
    return 0;
  }
  if (c->aovDepths)
  {
    c->refillMode = 0; // AOV frames are parity runs: no timing runs between them
    return 0;
  }
  if (key == c->refillKey)
    return 0;
  const uint32_t perTile = c->geom.tileSize * c->geom.tileSize;
  const uint32_t samples = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(c->geom.numSamples, 64u), c->pathCap / perTile));
  const uint64_t wantTiles = std::max<uint64_t>(1, (1ull << c->calibLog2) / ((uint64_t)perTile * samples));
  const bool both = c->framePipelined && c->geom.numLocalTiles >= 2; // one sample batch per slot, side by side
  // never more slots than the path buffers were sized for (rthip_begin: the widest batch a render call of this frame makes)
  uint64_t slotCap = std::min<uint64_t>(std::min<uint64_t>(c->rayO.n, c->listA.n), std::min<uint64_t>(c->hit.n, c->pathCap));
  if (both)
    slotCap = std::min<uint64_t>(slotCap, std::min<uint64_t>(std::min<uint64_t>(c->alt.rayO.n, c->alt.listA.n), c->alt.hit.n));
  // a sample batch is no wider than the batches the frame itself will be cut into; a frame that is ONE batch (the 1M soup: two
  // whole frames in flight on the two slots) is sampled whole on both slots -- half-frame batches have other tails than it has
  const BatchPlan plan = planBatches(c, c->geom.numSamples, c->geom, 0, c->framePipelined);
  const uint32_t tiles = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(wantTiles, plan.tiles), slotCap / ((uint64_t)perTile * samples));
  if (tiles == 0)
  {
    c->refillMode = 0;
    return 0;
  }
  const bool sideBySide = both && 2ull * tiles <= c->geom.numLocalTiles; // else both slots trace the same tiles
  // where in the tile list: one sample from the middle of its first half, the other from the middle of its second half -- a frame's
  // batches are runs of the list, so these are what the two slots will see (the default list is the even tile rows, then the odd
  // ones: both samples lie around the image centre; the middle of THAT list is where the bottom rows meet the top rows)
  const uint32_t nTiles = c->geom.numLocalTiles;
  const uint32_t first = sideBySide ? nTiles / 4 - std::min(nTiles / 4, tiles / 2) : (nTiles - tiles) / 2;
  const uint32_t second = sideBySide ? std::min(nTiles - tiles, std::max(first + tiles, nTiles / 2 + nTiles / 4 - std::min(nTiles / 4, tiles / 2))) : first;
  const uint32_t keepBegin = c->lastBatchBegin, keepCount = c->lastBatchCount, keepSamp = c->lastSamp;
  const bool keepRendered = c->renderedAny, keepTiming = c->kernelTiming;
  c->kernelTiming = false;
  uint32_t keepLaunches[6];
  for (int k = 0; k < 6; ++k)
    keepLaunches[k] = c->kernelLaunches[k]; // (the measurement's launches are not the frame's: rthip_get_kernel_times must not count them)
  float ms[4] = {0, 0, 0, 0};
  int rc = 0;
  const bool packets = packetsPossible(c) && c->fusedPrimary && c->info.numNodes > 64;
  const int numModes = packets ? 4 : 2;
  // `reps` sample pairs queued back to back between the two events: a small frame (the 1M soup: 2M paths) is over in a few
  // milliseconds, most of them ramp and tail, which frames queued one behind the other overlap with their neighbours -- one pair
  // timed alone favoured whatever shortens a lone frame's critical path (it picked mode 3 for the soup, 4.6 % slower in a run of frames)
  auto timedRun = [&](float* msOut, uint32_t reps) {
    rc = (int)hipEventRecord(c->evA, c->stream);
    if (!rc && both)
      rc = (int)hipStreamWaitEvent(c->alt.stream, c->evA, 0);
    for (uint32_t rep = 0; rep < reps && !rc; ++rep)
    {
      rc = renderBatch(c, 0, samples, first, tiles, false, nullptr, true);
      if (!rc && both)
      {
        swapSlot(c);
        rc = renderBatch(c, 0, samples, second, tiles, false, nullptr, true);
        swapSlot(c);
      }
    }
    if (!rc && both)
    {
      rc = (int)hipEventRecord(c->alt.evDone, c->alt.stream);
      if (!rc)
        rc = (int)hipStreamWaitEvent(c->stream, c->alt.evDone, 0);
    }
    if (!rc)
      rc = (int)hipEventRecord(c->evB, c->stream);
    if (!rc)
      rc = (int)hipEventSynchronize(c->evB);
    if (!rc && msOut)
    {
      rc = (int)hipEventElapsedTime(msOut, c->evA, c->evB);
      *msOut /= (float)reps;
    }
  };
  uint32_t reps = 1;
  for (int mode = 0; mode < numModes && !rc; ++mode)
  {
    c->refillMode = mode;
    if (mode == 0)
    {
      float warm = 0.0f;
      timedRun(&warm, 1); // warms the caches, and says how many pairs make a measurement of about 16 ms
      if (!rc && warm > 0.0f)
        reps = (uint32_t)std::max(1.0f, std::min(8.0f, 16.0f / warm));
    }
    if (!rc)
      timedRun(&ms[mode], reps);
  }
  c->kernelTiming = keepTiming;
  for (int k = 0; k < 6; ++k)
    c->kernelLaunches[k] = keepLaunches[k];
  c->lastBatchBegin = keepBegin, c->lastBatchCount = keepCount, c->lastSamp = keepSamp, c->renderedAny = keepRendered;
  c->refillMode = 0;
  if (rc)
    return rc;
  // anything but the low thresholds must win by more than the noise of the measurement
  float bestMs = 0.98f * ms[0];
  for (int m = 1; m < numModes; ++m)
    if (ms[m] > 0.0f && ms[m] < bestMs)
      bestMs = ms[m], c->refillMode = m;
  c->refillMs[0] = ms[0], c->refillMs[1] = ms[1], c->refillMs[2] = ms[2], c->refillMs[3] = ms[3];
  c->refillKey = key;
  return 0;
}

// tileFirst / tileCount: the local tiles to render (dynamic frames render the tiles of one rthip_enqueue_tiles call); the whole
// local tile list by default
int renderPasses(rthip_ctx* c, uint32_t sampBegin, uint32_t count, bool sync, uint32_t tileFirst = 0, uint32_t tileCount = ~0u)
{
  if (!c->haveScene || !c->haveFrame)
    return RTHIP_E_STATE;
  if (count == 0 || sampBegin + count > c->geom.numSamples)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  // Two batches in flight (see rthip_ctx::Slot): no batch needs a host round trip, so batches alternate between two sets of
  // path buffers and streams -- across calls and frames too -- and the ramp and tail of one batch's launches are filled by
  // the other's.  Nothing joins the slots at the end of the call; only the accumulate kernels are chained (per-pixel sums
  // stay in sample order).
  const bool pipelined = c->framePipelined;
  if (tileCount == ~0u)
    tileCount = c->geom.numLocalTiles - std::min(tileFirst, c->geom.numLocalTiles);
  if (tileCount == 0 || (uint64_t)tileFirst + tileCount > c->geom.numLocalTiles)
    return RTHIP_E_ARG;
  FrameGeom range = c->geom; // (planning only: how the tiles of this call are cut into batches)
  range.numLocalTiles = tileCount;
  const BatchPlan bp = planBatches(c, count, range, c->aovDepths, pipelined);
  // first call of the frame: its batches cover every local pixel once per sample range; in a dynamic frame every call brings
  // tiles that have not been rendered yet
  const bool fresh = c->accFresh || c->dynamic;
  const uint32_t tileEnd = tileFirst + tileCount;
  for (uint32_t done = 0; done < count; done += bp.samples)
  {
    const uint32_t n = std::min(bp.samples, count - done);
    for (uint32_t t0 = tileFirst; t0 < tileEnd; t0 += bp.tiles)
    {
      const uint32_t nt = std::min(bp.tiles, tileEnd - t0);
      const bool useAlt = pipelined && (c->batchToggle++ & 1u);
      if (useAlt)
        swapSlot(c);
      hipEvent_t after = (c->lastAcc && c->lastAccStream != c->stream) ? c->lastAcc : nullptr;
      int rc = 0;
      if (c->dynamic && done == 0)
      {
        // the origins of these tiles go up on the batch's own stream, right ahead of the kernels that read them (pinned
        // staging: nothing here waits for the device)
        const size_t off = (size_t)c->frameParity * c->dynCapacity + t0;
        rc = (int)hipMemcpyAsync(c->dynTable.p + off, c->dynStage + off, sizeof(uint32_t) * nt, hipMemcpyHostToDevice, c->stream);
      }
      if (!rc)
        rc = renderBatch(c, sampBegin + done, n, t0, nt, fresh && done == 0, after);
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
      // (a later batch on the same slot reuses its path buffers: same stream, so it runs after this one)
    }
  }
  c->accFresh = false;
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
  case RTHIP_E_COMM: return g_commError[0] ? g_commError : "rthip: RCCL error";
  default: return code > 0 ? hipGetErrorString((hipError_t)code) : "rthip: unknown error";
  }
}

// the light block of a scene: directional lights inline, point lights in an array of their own (the block points at it)
static int uploadLights(rthip_ctx* c, const rthip_dir_light* dirLights, uint32_t numDirLights, const rthip_point_light* pointLights, uint32_t numPointLights)
{
  int rc;
  if ((rc = c->lights.ensure(1)) || (rc = c->pointLights.ensure(std::max(1u, numPointLights))))
    return rc;
  Lights& L = c->hostLights;
  std::memset(&L, 0, sizeof(L));
  L.numDir = numDirLights;
  L.numPoint = numPointLights;
  L.point = c->pointLights.p;
  for (uint32_t i = 0; i < numDirLights; ++i)
    L.dir[i] = dirLights[i];
  if (numPointLights)
    HIPCHK(hipMemcpy(c->pointLights.p, pointLights, sizeof(rthip_point_light) * numPointLights, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->lights.p, &L, sizeof(L), hipMemcpyHostToDevice));
  return 0;
}

static int createContext(rthip_ctx* c, int device)
{
  c->device = device;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* t = getenv("RTHIP_TUNE")) // "refill,leaf[,blocksPerCU]" -- tuning experiments only
  {
    unsigned r = 0, l = 0, b = 0, st = 1, ov = 1, pl = 1, lh = 0, rs = 0, rp = 0, rh = 0, pa = 0;
    const int n = sscanf(t, "%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u", &r, &l, &b, &st, &ov, &pl, &lh, &rs, &rp, &rh, &pa);
    if (n >= 10 && rh >= 1 && rh <= 64)
      c->tuneRefillHi = rh;
    (void)pa;
    if (n >= 8 && rs >= 1 && rs <= 64)
      c->tuneRefillShadow = rs;
    if (n >= 9 && rp >= 1 && rp <= 64)
      c->tuneRefillPrimary = rp;
    if (n >= 7 && lh >= 1 && lh <= 64)
      c->tuneLeafHi = lh;
    if (n >= 2 && r >= 1 && r <= 64 && l >= 1 && l <= 64)
      c->tuneRefill = r, c->tuneLeaf = l, c->tuneLeafSet = true;
    if (n >= 3 && b >= 1 && b <= 8)
      c->tuneBlocks = b;
    if (n >= 4)
      c->tuneSteal = st ? kFetchShards : 1u;
    if (n >= 5)
      c->overlapShadow = ov ? 1u : 0u;
    if (n >= 6)
      c->pipelineBatches = std::min(pl, 16u);
  }
  if (const char* t = getenv("RTHIP_LIST_BLOCKS")) // blocks per CU of the list launches (depth >= 1), fixed instead of measured
  {
    const int v = atoi(t);
    if (v >= 1 && v <= 8)
      c->tuneBlocksLow = (uint32_t)v;
  }
  if (const char* t = getenv("RTHIP_TAIL_BLOCKS"))
  {
    const int v = atoi(t);
    if (v >= 1 && v <= 8)
      c->tuneBlocksTail = (uint32_t)v;
  }
  if (const char* t = getenv("RTHIP_SHADE_TAIL"))
    c->shadeTailBlocks = (uint32_t)std::max(1, std::min(128, atoi(t)));
  if (const char* t = getenv("RTHIP_BUILD_QUALITY"))
    c->buildQuality = std::max(0, std::min(1, atoi(t)));
  if (const char* t = getenv("RTHIP_BUILD_NORMALS"))
    c->buildNormalAware = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_BUILD_ROTATE"))
    c->buildRotate = (uint32_t)std::max(0, std::min(4, atoi(t)));
  if (const char* t = getenv("RTHIP_BUILD_DFS"))
    c->buildDfsOrder = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_BUILD_TCOST"))
    c->buildTraversalCost = (float)std::max(0.0, atof(t));
  if (const char* t = getenv("RTHIP_PLOC_RADIUS"))
    c->plocRadius = (uint32_t)std::max(1, std::min((int)kPlocMaxRadius, atoi(t)));
  if (const char* t = getenv("RTHIP_SHADOW_OPERANDS"))
    c->useOperands = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_FAT"))
    c->useFat = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_NO_SPLIT"))
    c->splitCalls = atoi(t) == 0;
  if (const char* t = getenv("RTHIP_PATHCAP_LOG2")) // tuning experiments only
  {
    const int v = atoi(t);
    if (v >= 16 && v <= 30)
      c->pathCap = 1ull << v;
  }
  if (const char* t = getenv("RTHIP_PACKET_BLOCKS"))
    c->packetBlocks = (uint32_t)std::max(1, std::min(16, atoi(t)));
  if (const char* t = getenv("RTHIP_PACKET_SHADOW"))
    c->packetShadow = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_CLOSEST_LOW"))
    c->closestLow = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_DEEP_LOW"))
    c->deepLow = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_CALIB_LOG2"))
    c->calibLog2 = (uint32_t)std::max(16, std::min(27, atoi(t)));
  if (const char* t = getenv("RTHIP_SPLIT_LIGHTS"))
    c->splitLights = atoi(t) != 0;
  if (const char* t = getenv("RTHIP_REFILL_MODE")) // 0 / 1: fixed low / high refill thresholds instead of the measured choice
    c->refillWanted = std::max(0, std::min(3, atoi(t)));
  if (const char* t = getenv("RTHIP_GROUP_LG"))
    c->groupLg = atoi(t);
  c->traceBlocksMax = (uint32_t)c->numCUs * c->tuneBlocks; // 6 blocks x 4 waves = 24 waves per CU: what 24 KB of LDS per block admits
  // Launches over slot lists (depth >= 1) get a smaller grid: scattered rays finish one by one, and with fewer resident waves
  // each wave has more of the launch's rays to refill its idle lanes with (1 M soup: node-step lane use 0.41 -> 0.53), while
  // the other batch in flight fills the rest of the GPU.  Camera-ray launches keep the full grid.  Frames queued back to back,
  // list launches at 3 / 4 / 6 blocks per CU on one box: soup1m 3.28 / 3.38 / 3.47 ms, hf10m 165 / 167 / 169 ms, hf10m at 4 spp
  // 18.8 / 18.9 / 19.6 ms, cube 1.05 / 1.06 / 1.18 ms.  (A single small frame alone on the GPU prefers 6 by up to 8 %.)
  c->traceBlocks = std::min(c->traceBlocksMax, (uint32_t)c->numCUs * c->tuneBlocksLow);
  c->traceBlocksTail = std::min(c->traceBlocksMax, (uint32_t)c->numCUs * c->tuneBlocksTail);
  // the static scratch of rthip_trace_rays_static: resolved once per device, with that device current
  if (device < kMaxDevices && !g_scratchPtr[device])
  {
    void* sp = nullptr;
    HIPCHK(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
    g_scratchPtr[device] = static_cast<unsigned char*>(sp);
  }
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
  return 0;
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
  if (int rc = createContext(c, device))
  {
    rthip_destroy(c); // streams / events created so far (null handles are skipped)
    return rc;
  }
  *out = c;
  return 0;
}

int rthip_destroy(rthip_ctx* c)
{
  if (!c)
    return RTHIP_E_ARG;
  (void)hipSetDevice(c->device);
  if (c->comm && rccl().ok)
    (void)rccl().commDestroy(c->comm);
  c->comm = nullptr;
  auto dropStream = [](hipStream_t& s) {
    if (s)
    {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
    s = nullptr;
  };
  auto dropEvent = [](hipEvent_t& e) {
    if (e)
      (void)hipEventDestroy(e);
    e = nullptr;
  };
  if (c->alt.stream)
    (void)hipStreamSynchronize(c->alt.stream);
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  dropStream(c->alt.stream2), dropStream(c->alt.stream), dropStream(c->stream2), dropStream(c->stream);
  dropEvent(c->evA), dropEvent(c->evB), dropEvent(c->evK0), dropEvent(c->evK1);
  for (hipEvent_t e : c->timerEvents)
    (void)hipEventDestroy(e);
  dropEvent(c->evFork), dropEvent(c->evJoin), dropEvent(c->evDone), dropEvent(c->evFrame);
  dropEvent(c->alt.evFork), dropEvent(c->alt.evJoin), dropEvent(c->alt.evDone);
  for (hipEvent_t& e : c->dynEvents)
    dropEvent(e);
  if (c->dynStage)
    (void)hipHostFree(c->dynStage);
  c->dynStage = nullptr;
  delete c; // the DevBuf members free their device memory
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
  if (numDirLights > kMaxShadowedLights || numPointLights > kMaxPointLights || info->numTris >= (1u << 28) - 1u)
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
  // children follow their parent (pre-order), so one ascending sweep both rules out cycles and measures the deepest chain
  // of 4-wide nodes: the spill stack is sized from what the arrays really hold, not from what the caller says
  uint32_t deepest = 1;
  {
    std::vector<uint8_t> level(info->numNodes, 1);
    for (uint32_t k = 0; k < info->numNodes; ++k)
      for (uint32_t ref : nodes[k].child)
      {
        if (ref == RTFLAT_INVALID)
          continue; // unused slot (its box is inverted)
        if (rtflat_is_leaf(ref) ? (rtflat_leaf_first(ref) + rtflat_leaf_count(ref) > info->numTris) : (ref >= info->numNodes || ref <= k))
          return RTHIP_E_ARG;
        if (!rtflat_is_leaf(ref))
        {
          if (level[k] >= 250)
            return RTHIP_E_LIMIT;
          level[ref] = std::max<uint8_t>(level[ref], (uint8_t)(level[k] + 1));
          deepest = std::max<uint32_t>(deepest, level[ref]);
        }
      }
  }
  deepest = std::max(deepest, info->maxDepthBlas);
  if (3 * deepest + 2 > kMaxStack)
    return RTHIP_E_LIMIT;
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
  if (int rcLights = uploadLights(c, dirLights, numDirLights, pointLights, numPointLights))
    return rcLights;
  c->info = *info;
  c->info.maxDepthBlas = deepest;
  c->spillCap = 3 * deepest + 2 > kStackLds ? 3 * deepest + 2 - kStackLds : 1u;
  c->spill.release(); // sized per scene in ensurePathBuffers
  c->alt.spill.release();
  c->materialsOn = false; // they described the previous scene's triangles
  if (int rcFat = fattenNodes(c))
    return rcFat;
  c->sceneEpoch++;
  c->haveScene = true;
  c->haveFrame = false;
  return 0;
}

// Uploads the scene's triangles and BUILDS the triangle-level BVH on the device (rt_lbvh.h): Morton codes, radix sort, binary
// radix tree, level-by-level refit, collapse into the same quantised 4-wide rtflat_node records the host builder writes.
int rthip_upload_scene_build(rthip_ctx* c, rtflat_info* info, rtflat_model_node* modelNodes, const rtflat_tri* trisModelOrder, uint32_t maxLeafTris,
                             const rthip_dir_light* dirLights, uint32_t numDirLights, const rthip_point_light* pointLights, uint32_t numPointLights,
                             float* buildMsOut)
{
  if (!c || !info || !modelNodes || !trisModelOrder || (numDirLights && !dirLights) || (numPointLights && !pointLights))
    return RTHIP_E_ARG;
  if (info->numModelNodes == 0 || info->numTris == 0)
    return RTHIP_E_ARG;
  if (numDirLights > kMaxShadowedLights || numPointLights > kMaxPointLights || info->numTris >= (1u << 28) - 1u)
    return RTHIP_E_LIMIT;
  const uint32_t maxLeaf = std::max(1u, std::min(maxLeafTris, (uint32_t)RTFLAT_MAX_LEAF_TRIS));
  const uint32_t numTris = info->numTris;
  uint32_t largest = 0;
  for (uint32_t k = 0; k < info->numModelNodes; ++k)
  {
    const rtflat_model_node& mn = modelNodes[k];
    if (mn.subtreeSize == 0 || k + mn.subtreeSize > info->numModelNodes)
      return RTHIP_E_ARG;
    if (mn.triCount > 0 && ((uint64_t)mn.indicesOffset + mn.triCount > numTris))
      return RTHIP_E_ARG; // a model's records are [indicesOffset, indicesOffset + triCount) of the model-order array
    largest = std::max(largest, mn.triCount);
  }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->alt.stream)); // frames still in flight on either slot read the old scene
  c->altPending = false;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->haveScene = false;
  c->haveFrame = false;
  c->materialsOn = false;
  int rc;
  const uint32_t nodeCap = std::max(1u, numTris);
  const bool ploc = c->buildQuality >= 1;
  const uint32_t radius = std::max(1u, std::min(c->plocRadius, kPlocMaxRadius));
  DevBuf<rtflat_tri> trisIn;
  DevBuf<LbvhBox> primBoxes, innerBoxes, clusterBoxA, clusterBoxB;
  DevBuf<uint64_t> codes, codesSorted;
  DevBuf<uint32_t> ids, idsSorted, order, small; // small: [nodeCounter | overflow | level counts (kLevels + 2) | binary node counter | cluster count]
  DevBuf<uint32_t> clusterRefA, clusterRefB, nearest, flags, offsets;
  DevBuf<LbvhNode> binary;
  DevBuf<uint8_t> done, sortTemp, scanTemp;
  DevBuf<uint4> frontier;
  DevBuf<uint32_t> dfsSizes, dfsIndex;
  DevBuf<rtflat_node> dfsNodes;
  constexpr uint32_t kLevels = (kMaxStack - 2) / 3; // deepest chain of 4-wide nodes the traversal stack can take
  constexpr uint32_t kSmallBinary = 2 + kLevels + 2, kSmallClusters = kSmallBinary + 1, kSmallWords = kSmallClusters + 1;
  if ((rc = c->modelNodes.ensure(info->numModelNodes)) || (rc = c->nodes.ensure(nodeCap)) || (rc = c->tris.ensure(numTris)) || (rc = c->lights.ensure(1)) ||
      (rc = trisIn.ensure(numTris)) || (rc = primBoxes.ensure(largest)) || (rc = innerBoxes.ensure(largest)) || (rc = codes.ensure(largest)) ||
      (rc = codesSorted.ensure(largest)) || (rc = ids.ensure(largest)) || (rc = idsSorted.ensure(largest)) || (rc = binary.ensure(largest)) ||
      (rc = order.ensure(largest)) || (rc = frontier.ensure(2 * (size_t)largest)) || (rc = small.ensure(kSmallWords)))
    return rc;
  if (ploc)
  {
    if ((rc = clusterBoxA.ensure(largest)) || (rc = clusterBoxB.ensure(largest)) || (rc = clusterRefA.ensure(largest)) || (rc = clusterRefB.ensure(largest)) ||
        (rc = nearest.ensure(largest)) || (rc = flags.ensure(largest)) || (rc = offsets.ensure(largest)))
      return rc;
  }
  else if ((rc = done.ensure(largest)))
    return rc;
  // temporary storage of the radix sort (and of PLOC's prefix sum): rocPRIM picks its algorithm by size, so the requirement is
  // asked for every model size that will be processed and the largest answer is allocated (not assumed monotone in n)
  size_t sortBytes = 0, scanBytes = 0;
  for (uint32_t k = 0; k < info->numModelNodes; ++k)
  {
    const uint32_t n = modelNodes[k].triCount;
    if (n <= maxLeaf)
      continue;
    size_t need = 0;
    if (rocprim::radix_sort_pairs(nullptr, need, codes.p, codesSorted.p, ids.p, idsSorted.p, n, 0, 64, c->stream) != hipSuccess)
      return RTHIP_E_ARG;
    sortBytes = std::max(sortBytes, need);
    if (ploc)
    {
      // (the scan runs on shrinking cluster arrays; powers of two down from n cover the size classes rocPRIM distinguishes)
      for (uint32_t m = n; m >= 1; m = (m > 1 ? m / 2 : 0))
      {
        need = 0;
        if (rocprim::exclusive_scan(nullptr, need, flags.p, offsets.p, 0u, m, rocprim::plus<uint32_t>(), c->stream) != hipSuccess)
          return RTHIP_E_ARG;
        scanBytes = std::max(scanBytes, need);
        if (m == 1)
          break;
      }
    }
  }
  if ((rc = sortTemp.ensure(sortBytes)) || (rc = scanTemp.ensure(2 * scanBytes + 256)))
    return rc;
  HIPCHK(hipMemcpy(trisIn.p, trisModelOrder, sizeof(rtflat_tri) * numTris, hipMemcpyHostToDevice));
  HIPCHK(hipEventRecord(c->evA, c->stream));
  uint32_t nodesUsed = 0, deepest = 1;
  const float3 sceneLo = make_float3(info->sceneLo[0], info->sceneLo[1], info->sceneLo[2]);
  auto inv = [](float e) { return e > 0.0f ? 1.0f / e : 0.0f; };
  const float3 sceneInvExt = make_float3(inv(info->sceneHi[0] - info->sceneLo[0]), inv(info->sceneHi[1] - info->sceneLo[1]), inv(info->sceneHi[2] - info->sceneLo[2]));
  // Morton bit layout: the code's 63 bits, most significant first, go to whichever axis a cell is longest along at that
  // point (its extent is then halved), each axis taking at most 21.  A cube-shaped cloud gets the usual x,y,z interleave.
  // How long a cell is along an axis is not only the scene's extent: a surface that locally runs ACROSS an axis (a terrain's
  // height) spans, inside a small cell, only its slope times the cell's width however tall the whole scene is -- spending a
  // bit there splits a patch into "higher" and "lower" triangles whose boxes overlap in plan.  The triangles' normals say how
  // the surfaces run: w_a = sqrt(1 - mean(n_a^2)) is large along axes the surfaces extend in; a cell's extent along a is taken
  // as min(scene extent so far, w_a / w_max x its longest extent).  Terrain: x and z take their 21 bits each before the height
  // gets one; a triangle soup (normals every way) keeps the round robin.
  MortonLayout layout;
  {
    double n2[3] = {0, 0, 0};
    const uint32_t stride = std::max(1u, numTris / (1u << 20));
    uint64_t samples = 0;
    for (uint32_t t = 0; t < numTris; t += stride, ++samples)
      for (int a = 0; a < 3; ++a)
        n2[a] += (double)trisModelOrder[t].n[a] * trisModelOrder[t].n[a];
    double w[3], wmax = 0.0;
    for (int a = 0; a < 3; ++a)
    {
      const double m = samples ? n2[a] / (double)samples : 1.0 / 3.0;
      w[a] = std::sqrt(std::max(0.0, 1.0 - std::min(1.0, m)));
      if (!(w[a] == w[a]))
        w[a] = 0.8; // (NaN normals of degenerate records)
      wmax = std::max(wmax, w[a]);
    }
    if (!c->buildNormalAware || wmax <= 0.0)
      w[0] = w[1] = w[2] = wmax = 1.0;
    double ext[3] = {std::max(0.0, (double)info->sceneHi[0] - info->sceneLo[0]), std::max(0.0, (double)info->sceneHi[1] - info->sceneLo[1]),
                     std::max(0.0, (double)info->sceneHi[2] - info->sceneLo[2])};
    uint32_t taken[3] = {0, 0, 0};
    for (int bit = 62; bit >= 0; --bit)
    {
      const double longest = std::max(ext[0], std::max(ext[1], ext[2]));
      int pick = -1;
      double best = -1.0;
      for (int a = 0; a < 3; ++a)
      {
        if (taken[a] >= 21)
          continue;
        const double eff = std::min(ext[a], w[a] / wmax * longest);
        if (eff > best)
          best = eff, pick = a;
      }
      layout.pos[pick][taken[pick]++] = (uint8_t)bit;
      ext[pick] *= 0.5;
    }
  }
  for (uint32_t k = 0; k < info->numModelNodes; ++k)
  {
    rtflat_model_node& mn = modelNodes[k];
    if (mn.triCount == 0)
      continue;
    const uint32_t n = mn.triCount, base = mn.indicesOffset;
    if (n <= maxLeaf)
    {
      // a model this small is one leaf: its records stay in model order
      HIPCHK(hipMemcpyAsync(c->tris.p + base, trisIn.p + base, sizeof(rtflat_tri) * n, hipMemcpyDeviceToDevice, c->stream));
      mn.blas = rtflat_make_leaf(base, n);
      continue;
    }
    const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    hipLaunchKernelGGL(k_lbvh_prims, grid, block, 0, c->stream, trisIn.p + base, n, info->padAbs, sceneLo, sceneInvExt, layout, primBoxes.p, codes.p, ids.p);
    HIPCHK(hipGetLastError());
    size_t sortBytesNow = sortBytes; // (rocPRIM takes the size by reference)
    if (rocprim::radix_sort_pairs(sortTemp.p, sortBytesNow, codes.p, codesSorted.p, ids.p, idsSorted.p, n, 0, 64, c->stream) != hipSuccess)
      return RTHIP_E_ARG;
    uint32_t binaryRoot = 0;
    if (!ploc)
    {
      hipLaunchKernelGGL(k_lbvh_topology, grid, block, 0, c->stream, codesSorted.p, n, binary.p);
      HIPCHK(hipGetLastError());
      // refit: one launch per level of the binary tree, until the root has its box (checked every 8 passes)
      bool rooted = false;
      for (uint32_t round = 0; round < std::max(1u, c->buildRotate); ++round)
      {
      HIPCHK(hipMemsetAsync(done.p, 0, n - 1, c->stream));
      rooted = false;
      for (uint32_t pass = 1; pass <= 200 && !rooted; ++pass)
      {
        hipLaunchKernelGGL(k_lbvh_refit, dim3((n - 1 + kBlock - 1) / kBlock), block, 0, c->stream, binary.p, n - 1, primBoxes.p, idsSorted.p, innerBoxes.p, done.p, pass, c->buildRotate ? 1u : 0u);
        HIPCHK(hipGetLastError());
        if ((pass & 7u) == 0)
        {
          uint8_t flag = 0;
          HIPCHK(hipMemcpyAsync(&flag, done.p, 1, hipMemcpyDeviceToHost, c->stream));
          HIPCHK(hipStreamSynchronize(c->stream));
          rooted = flag != 0;
        }
      }
      }
      if (!rooted)
        return RTHIP_E_LIMIT; // deeper than any tree the traversal stack could take
    }
    else
    {
      // PLOC: the Morton-ordered clusters merge with their nearest neighbours round after round (see rt_lbvh.h)
      HIPCHK(hipMemsetAsync(small.p + kSmallBinary, 0, sizeof(uint32_t), c->stream));
      hipLaunchKernelGGL(k_ploc_init, grid, block, 0, c->stream, primBoxes.p, idsSorted.p, n, clusterRefA.p, clusterBoxA.p);
      HIPCHK(hipGetLastError());
      uint32_t count = n;
      uint32_t *refIn = clusterRefA.p, *refOut = clusterRefB.p;
      LbvhBox *boxIn = clusterBoxA.p, *boxOut = clusterBoxB.p;
      for (uint32_t round = 0; count > 1; ++round)
      {
        if (round > 4096)
          return RTHIP_E_LIMIT; // (every round merges at least the closest pair: n rounds at the very worst)
        const dim3 g((count + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_ploc_nearest, g, block, 0, c->stream, boxIn, count, radius, nearest.p);
        hipLaunchKernelGGL(k_ploc_merge, g, block, 0, c->stream, refIn, boxIn, count, nearest.p, small.p + kSmallBinary, binary.p, innerBoxes.p, flags.p);
        HIPCHK(hipGetLastError());
        size_t scanNow = scanTemp.n;
        if (rocprim::exclusive_scan(scanTemp.p, scanNow, flags.p, offsets.p, 0u, count, rocprim::plus<uint32_t>(), c->stream) != hipSuccess)
          return RTHIP_E_ARG;
        hipLaunchKernelGGL(k_ploc_compact, g, block, 0, c->stream, refIn, boxIn, flags.p, offsets.p, count, refOut, boxOut, small.p + kSmallClusters);
        HIPCHK(hipGetLastError());
        uint32_t next = 0;
        HIPCHK(hipMemcpyAsync(&next, small.p + kSmallClusters, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (next == 0 || next >= count)
          return RTHIP_E_LIMIT; // cannot happen (the globally closest pair is mutual); guards the loop
        count = next;
        std::swap(refIn, refOut), std::swap(boxIn, boxOut);
      }
      HIPCHK(hipMemcpy(&binaryRoot, refIn, sizeof(uint32_t), hipMemcpyDeviceToHost));
      if (binaryRoot & 0x80000000u)
        return RTHIP_E_ARG; // n > maxLeaf >= 1 triangles always end in an inner node
    }
    // collapse to 4-wide nodes, one launch per level of the wide tree
    const uint32_t root = nodesUsed;
    if (root >= nodeCap)
      return RTHIP_E_LIMIT;
    {
      std::vector<uint32_t> init(2 + kLevels + 2, 0u);
      init[0] = root + 1; // next free node
      init[2] = 1;        // level 0 holds the root
      HIPCHK(hipMemcpyAsync(small.p, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      const uint4 first = make_uint4(binaryRoot, root, 0u, 0u);
      HIPCHK(hipMemcpyAsync(frontier.p, &first, sizeof(first), hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream)); // (the two host sources above are stack / local memory)
    }
    LbvhEmitArgs ea;
    ea.nodes = binary.p, ea.innerBoxes = innerBoxes.p, ea.primBoxes = primBoxes.p, ea.ids = idsSorted.p;
    ea.out = c->nodes.p, ea.outCapacity = nodeCap, ea.nodeCounter = small.p, ea.overflow = small.p + 1;
    ea.recordBase = base, ea.maxLeaf = maxLeaf, ea.order = order.p, ea.traversalCost = c->buildTraversalCost;
    for (uint32_t level = 0; level <= kLevels; ++level)
    {
      ea.frontierIn = frontier.p + (size_t)(level & 1u) * largest;
      ea.frontierOut = frontier.p + (size_t)((level + 1) & 1u) * largest;
      ea.countIn = small.p + 2 + level;
      ea.countOut = small.p + 2 + level + 1;
      hipLaunchKernelGGL(k_lbvh_emit, dim3(std::min<uint32_t>((n + kBlock - 1) / kBlock, (uint32_t)c->numCUs * 8u)), block, 0, c->stream, ea);
      HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_lbvh_gather, grid, block, 0, c->stream, trisIn.p + base, order.p, n, c->tris.p + base);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> fin(2 + kLevels + 2, 0u);
    HIPCHK(hipMemcpyAsync(fin.data(), small.p, fin.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (fin[1] || fin[2 + kLevels + 1]) // node array overflow, or nodes left beyond the deepest level the stack can take
      return RTHIP_E_LIMIT;
    uint32_t modelLevels = 0;
    for (uint32_t level = 0; level <= kLevels; ++level)
      if (fin[2 + level])
        modelLevels = level + 1;
    deepest = std::max(deepest, modelLevels);
    if (c->buildDfsOrder && fin[0] - root > 1)
    {
      // breadth-first (as emitted) -> depth-first pre-order, the host builder's layout: a subtree becomes one contiguous run
      const uint32_t count = fin[0] - root;
      if ((rc = dfsSizes.ensure(nodeCap)) || (rc = dfsIndex.ensure(nodeCap)) || (rc = dfsNodes.ensure(largest)))
        return rc;
      std::vector<uint32_t> levelBegin(modelLevels + 1, root);
      for (uint32_t level = 0; level < modelLevels; ++level)
        levelBegin[level + 1] = levelBegin[level] + fin[2 + level];
      if (levelBegin[modelLevels] != fin[0] || count > dfsNodes.n)
        return RTHIP_E_LIMIT;
      for (uint32_t level = modelLevels; level-- > 0;)
        hipLaunchKernelGGL(k_dfs_sizes, dim3((fin[2 + level] + kBlock - 1) / kBlock), block, 0, c->stream, c->nodes.p, levelBegin[level], fin[2 + level], dfsSizes.p);
      HIPCHK(hipMemcpyAsync(dfsIndex.p + root, &root, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream)); // (&root is a local)
      for (uint32_t level = 0; level < modelLevels; ++level)
        hipLaunchKernelGGL(k_dfs_offsets, dim3((fin[2 + level] + kBlock - 1) / kBlock), block, 0, c->stream, c->nodes.p, levelBegin[level], fin[2 + level], dfsSizes.p,
                           dfsIndex.p);
      // (out[newIndex] with newIndex in [root, root + count): staged through a buffer of the model's size, then copied back)
      hipLaunchKernelGGL(k_dfs_permute, dim3((count + kBlock - 1) / kBlock), block, 0, c->stream, c->nodes.p, root, count, dfsIndex.p, dfsNodes.p - root);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(c->nodes.p + root, dfsNodes.p, sizeof(rtflat_node) * count, hipMemcpyDeviceToDevice, c->stream));
    }
    nodesUsed = fin[0];
    mn.blas = root;
  }
  if (nodesUsed == 0)
  {
    // every model was a single leaf: keep the node array non-empty (one node without children, as the host builder does)
    rtflat_node none{};
    for (int a = 0; a < 3; ++a)
      for (int k = 0; k < 4; ++k)
        none.qlo[a][k] = 255, none.qhi[a][k] = 0;
    for (uint32_t& ch : none.child)
      ch = RTFLAT_INVALID;
    none.exp[0] = none.exp[1] = none.exp[2] = 27;
    HIPCHK(hipMemcpy(c->nodes.p, &none, sizeof(none), hipMemcpyHostToDevice));
    nodesUsed = 1;
  }
  HIPCHK(hipEventRecord(c->evB, c->stream));
  HIPCHK(hipEventSynchronize(c->evB));
  if (buildMsOut)
    HIPCHK(hipEventElapsedTime(buildMsOut, c->evA, c->evB));
  if (3 * deepest + 2 > kMaxStack)
    return RTHIP_E_LIMIT;
  HIPCHK(hipMemcpy(c->modelNodes.p, modelNodes, sizeof(rtflat_model_node) * info->numModelNodes, hipMemcpyHostToDevice));
  if (int rcLights = uploadLights(c, dirLights, numDirLights, pointLights, numPointLights))
    return rcLights;
  if ((size_t)nodesUsed * 2 < c->nodes.n)
  {
    // the node array was sized for the worst case (one node per triangle); keep what the tree uses
    DevBuf<rtflat_node> tight;
    if ((rc = tight.ensure(nodesUsed)))
      return rc;
    HIPCHK(hipMemcpy(tight.p, c->nodes.p, sizeof(rtflat_node) * nodesUsed, hipMemcpyDeviceToDevice));
    c->nodes = std::move(tight);
  }
  info->numNodes = nodesUsed;
  info->maxDepthBlas = deepest;
  c->info = *info;
  c->spillCap = 3 * deepest + 2 > kStackLds ? 3 * deepest + 2 - kStackLds : 1u;
  c->spill.release(); // sized per scene in ensurePathBuffers
  c->alt.spill.release();
  if (int rcFat = fattenNodes(c))
    return rcFat;
  c->sceneEpoch++;
  c->haveScene = true;
  return 0;
}

// Moved vertices, same topology: the records take the new operands and every box of the triangle-level tree is recomputed
// bottom-up (one launch per level, as many as the tree is deep).  No counterpart in the reference, whose CPU path has no
// triangle-level tree to keep up to date.
int rthip_refit(rthip_ctx* c, const rtflat_model_node* modelNodes, const rtflat_tri* trisModelOrder, float* msOut)
{
  if (!c || !modelNodes || !trisModelOrder)
    return RTHIP_E_ARG;
  if (!c->haveScene)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  // frames still in flight on either slot read the scene
  HIPCHK(hipStreamSynchronize(c->alt.stream));
  HIPCHK(hipStreamSynchronize(c->alt.stream2));
  HIPCHK(hipStreamSynchronize(c->stream2));
  c->altPending = false;
  HIPCHK(hipStreamSynchronize(c->stream));
  const uint32_t numTris = c->info.numTris, numNodes = c->info.numNodes, numModelNodes = c->info.numModelNodes;
  DevBuf<rtflat_tri> incoming;
  DevBuf<rtflat_model_node> incomingModels;
  DevBuf<LbvhBox> nodeBoxes;
  DevBuf<uint8_t> done;
  DevBuf<uint32_t> flag;
  int rc;
  if ((rc = incoming.ensure(numTris)) || (rc = incomingModels.ensure(numModelNodes)) || (rc = nodeBoxes.ensure(numNodes)) || (rc = done.ensure(numNodes)) ||
      (rc = flag.ensure(1)))
    return rc;
  HIPCHK(hipMemcpy(incoming.p, trisModelOrder, sizeof(rtflat_tri) * numTris, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(incomingModels.p, modelNodes, sizeof(rtflat_model_node) * numModelNodes, hipMemcpyHostToDevice));
  HIPCHK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), c->stream));
  HIPCHK(hipMemsetAsync(done.p, 0, numNodes, c->stream));
  HIPCHK(hipEventRecord(c->evA, c->stream));
  const float3 lo = make_float3(c->info.sceneLo[0], c->info.sceneLo[1], c->info.sceneLo[2]), hi = make_float3(c->info.sceneHi[0], c->info.sceneHi[1], c->info.sceneHi[2]);
  // How far may the triangles move?  The padding of every box (padAbs) was sized for coordinates up to `reach` = |centre| +
  // originBound + diagonal (BvhFlattener.cpp): vertices anywhere within one scene diagonal of the flattened box stay inside it,
  // and so do the "effective" triangles the reference's edge indexing makes of another triangle's edges.
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  const float slack = std::sqrt(dx * dx + dy * dy + dz * dz);
  hipLaunchKernelGGL(k_refit_tris, dim3((numTris + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->tris.p, incoming.p, numTris, c->info.padAbs,
                     make_float3(lo.x - slack, lo.y - slack, lo.z - slack), make_float3(hi.x + slack, hi.y + slack, hi.z + slack), flag.p);
  HIPCHK(hipGetLastError());
  for (uint32_t pass = 1; pass <= c->info.maxDepthBlas + 1; ++pass)
  {
    hipLaunchKernelGGL(k_refit_nodes, dim3((numNodes + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->nodes.p, numNodes, c->tris.p, c->info.padAbs, nodeBoxes.p,
                       done.p, pass);
    HIPCHK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_refit_model_boxes, dim3((numModelNodes + 63) / 64), dim3(64), 0, c->stream, c->modelNodes.p, incomingModels.p, numModelNodes);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->evB, c->stream));
  uint32_t outside = 0;
  HIPCHK(hipMemcpyAsync(&outside, flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (msOut)
    HIPCHK(hipEventElapsedTime(msOut, c->evA, c->evB));
  c->sceneEpoch++; // (the traversal-mode measurement is taken again: the rays may behave differently now)
  if (!outside)
    if (int rcFat = fattenNodes(c))
      return rcFat;
  if (outside)
  {
    c->haveScene = false; // the records have been overwritten; the caller flattens and uploads again
    return RTHIP_E_BOUNDS;
  }
  return 0;
}

int rthip_set_build_quality(rthip_ctx* c, int quality)
{
  if (!c || quality < 0 || quality > 1)
    return RTHIP_E_ARG;
  c->buildQuality = quality;
  return 0;
}

// the scene arrays as the device holds them (after rthip_upload_scene_build: what the device built)
int rthip_download_scene(rthip_ctx* c, rtflat_node* nodesOut, uint32_t nodeCapacity, rtflat_tri* trisOut, uint32_t triCapacity)
{
  if (!c || (!nodesOut && !trisOut))
    return RTHIP_E_ARG;
  if (!c->haveScene)
    return RTHIP_E_STATE;
  if ((nodesOut && nodeCapacity < c->info.numNodes) || (trisOut && triCapacity < c->info.numTris))
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (nodesOut)
    HIPCHK(hipMemcpy(nodesOut, c->nodes.p, sizeof(rtflat_node) * c->info.numNodes, hipMemcpyDeviceToHost));
  if (trisOut)
    HIPCHK(hipMemcpy(trisOut, c->tris.p, sizeof(rtflat_tri) * c->info.numTris, hipMemcpyDeviceToHost));
  return 0;
}

int rthip_upload_materials(rthip_ctx* c, const rtflat_shade* shade, uint32_t numTris, const rtflat_material* materials, uint32_t numMaterials,
                           const float* texels, uint64_t numTexels)
{
  if (!c || !shade || !materials || numMaterials == 0 || (numTexels && !texels))
    return RTHIP_E_ARG;
  if (!c->haveScene)
    return RTHIP_E_STATE;
  if (numTris != c->info.numTris)
    return RTHIP_E_ARG; // one record per triangle of the uploaded scene, indexed by triId
  // everything the kernels index with is checked here, once: they do no bounds checks of their own
  for (uint32_t m = 0; m < numMaterials; ++m)
  {
    const rtflat_material& mt = materials[m];
    if (mt.kind != RTFLAT_MAT_STUB && mt.kind != RTFLAT_MAT_FACTORS)
      return RTHIP_E_ARG;
    auto inPool = [&](uint32_t first, uint32_t w, uint32_t h) {
      if (w == 0)
        return true;
      return h != 0 && w <= (1u << 24) && h <= (1u << 24) && (uint64_t)first + (uint64_t)w * h <= numTexels;
    };
    if (!inPool(mt.albedoTexel, mt.albedoW, mt.albedoH) || !inPool(mt.emissiveTexel, mt.emissiveW, mt.emissiveH))
      return RTHIP_E_ARG;
  }
  for (uint32_t k = 0; k < numTris; ++k)
    if (shade[k].material >= numMaterials)
      return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->alt.stream)); // frames in flight shade with the old tables
  HIPCHK(hipStreamSynchronize(c->alt.stream2));
  HIPCHK(hipStreamSynchronize(c->stream2));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->altPending = false;
  c->materialsOn = false;
  int rc;
  if ((rc = c->shade.ensure(numTris)) || (rc = c->materials.ensure(numMaterials)) || (rc = c->texels.ensure(std::max<uint64_t>(1, 3 * numTexels))))
    return rc;
  HIPCHK(hipMemcpy(c->shade.p, shade, sizeof(rtflat_shade) * numTris, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->materials.p, materials, sizeof(rtflat_material) * numMaterials, hipMemcpyHostToDevice));
  if (numTexels)
    HIPCHK(hipMemcpy(c->texels.p, texels, sizeof(float) * 3 * numTexels, hipMemcpyHostToDevice));
  c->materialsOn = true;
  c->haveFrame = false; // the next rthip_begin sizes the albedo array of the path state
  return 0;
}

int rthip_set_point_shadows(rthip_ctx* c, int on)
{
  if (!c)
    return RTHIP_E_ARG;
  if (on && c->haveScene && c->hostLights.numDir + c->hostLights.numPoint > kMaxShadowedLights)
    return RTHIP_E_LIMIT;
  c->pointShadows = on != 0; // (read when launches are queued: takes effect with the next render call)
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
  if (c->pointShadows && c->hostLights.numDir + c->hostLights.numPoint > kMaxShadowedLights)
    return RTHIP_E_LIMIT; // one occlusion bit per shadowed light in the 32-bit mask
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
  // No host synchronisation here in the steady state: a frame is enqueued behind the previous one on the context's streams, so
  // a caller that uses rthip_enqueue_passes keeps the GPU busy across frames.  Exceptions, all one-off: a buffer that has to grow,
  // a changed tile table, and the traversal-mode measurement at the end (first frame of a scene / frame shape: calibrateRefill).
  // Everything is validated and sized into locals first; the context's frame state changes only once nothing can fail.
  const uint32_t tilesX = (f->width + ts - 1) / ts, tilesY = (f->height + ts - 1) / ts;
  FrameGeom g{};
  g.width = f->width, g.height = f->height, g.numSamples = f->numSamples;
  g.tileSize = ts, g.tilesX = tilesX;
  std::vector<uint32_t> tileIds;
  if (f->numTiles)
  {
    if (!f->tileIds)
      return RTHIP_E_ARG;
    tileIds.reserve(f->numTiles);
    for (uint32_t k = 0; k < f->numTiles; ++k)
    {
      if (f->tileIds[k] >= tilesX * tilesY)
        return RTHIP_E_ARG;
      tileIds.push_back(f->tileIds[k]);
    }
  }
  else
  {
    // Every tile, the even tile rows top to bottom and then the odd ones.  A frame that is cut into two batches (planBatches) takes
    // the first and the second half of this list; in row-major order those are the upper and the lower half of the image, which
    // on a scene under a horizon cost very different amounts, and one slot ends up carrying every frame's dear half (hf10m:
    // 145.4 ms per frame in row-major order, 139.5 this way; tools/tile_order_probe.py).  Interleaved rows halve any image evenly.
    tileIds.reserve((size_t)tilesX * tilesY);
    for (uint32_t parity = 0; parity < 2; ++parity)
      for (uint32_t y = parity; y < tilesY; y += 2)
        for (uint32_t x = 0; x < tilesX; ++x)
          tileIds.push_back(y * tilesX + x);
  }
  g.numLocalTiles = (uint32_t)tileIds.size();
  const uint64_t P64 = (uint64_t)g.numLocalTiles * ts * ts;
  if (P64 == 0 || P64 > (1ull << 30))
    return RTHIP_E_LIMIT;
  g.numLocalPixels = (uint32_t)P64;
  const uint32_t aovDepths = std::min(f->aovDepths, f->maxDepth);
  // the largest batch any render call of this frame can make (planBatches): both slots get buffers for it
  const bool pipelinedFrame = c->pipelineBatches && aovDepths == 0;
  const BatchPlan widest = planBatches(c, f->numSamples, g, aovDepths, pipelinedFrame);
  const uint64_t numSlots64 = (uint64_t)widest.tiles * ts * ts * widest.samples;
  if (numSlots64 > (1ull << 30))
    return RTHIP_E_LIMIT;
  const size_t numSlots = (size_t)numSlots64;
  int rc;
  // the previous frame may still be running on either slot: its kernels read the tile table and, when a buffer has to grow,
  // the memory about to be freed
  auto quiesce = [&]() -> int {
    HIPCHK(hipStreamSynchronize(c->alt.stream));
    HIPCHK(hipStreamSynchronize(c->alt.stream2));
    HIPCHK(hipStreamSynchronize(c->stream2));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->altPending = false;
    return 0;
  };
  const bool grows = numSlots > c->rayO.n || (pipelinedFrame && numSlots > c->alt.rayO.n) ||
                     (c->materialsOn && (numSlots > c->alb.n || (pipelinedFrame && numSlots > c->alt.alb.n))) || P64 * 3 > c->acc.n || g.numLocalTiles > c->tileIds.n || !c->camConsts.p || !c->totals.p ||
                     (aovDepths && (size_t)P64 * widest.samples * aovDepths > c->aovTri.n);
  if (grows && (rc = quiesce()))
    return rc;
  if ((rc = c->tileIds.ensure(g.numLocalTiles)) || (rc = ensurePathBuffers(c, numSlots)) || (rc = c->acc.ensure(P64 * 3)) ||
      (rc = c->img.ensure(P64 * 3)) || (rc = c->camConsts.ensure(2)) || (rc = c->totals.ensure(2)))
    return rc;
  if (aovDepths && ((rc = c->aovTri.ensure((size_t)P64 * widest.samples * aovDepths)) || (rc = c->aovShadow.ensure((size_t)P64 * widest.samples * aovDepths))))
    return rc;
  if (pipelinedFrame)
  {
    swapSlot(c);
    rc = ensurePathBuffers(c, numSlots);
    swapSlot(c);
    if (rc)
      return rc;
  }
  {
    std::vector<uint32_t> xy(g.numLocalTiles);
    for (uint32_t k = 0; k < g.numLocalTiles; ++k)
      xy[k] = ((tileIds[k] % tilesX) * ts) | (((tileIds[k] / tilesX) * ts) << 16);
    if (grows || xy != c->hostTileXY) // frames of one animation keep their tiling: upload the table only when it changes
    {
      if (!grows && (rc = quiesce()))
        return rc;
      c->hostTileXY.clear(); // (stays empty if the copy fails: the next frame uploads again)
      HIPCHK(hipMemcpy(c->tileIds.p, xy.data(), sizeof(uint32_t) * g.numLocalTiles, hipMemcpyHostToDevice));
      c->hostTileXY.swap(xy);
    }
  }
  // ---- commit
  g.tileXY = c->tileIds.p;
  g.tileShift = (uint32_t)__builtin_ctz(ts);
  g.batchSamples = 1, g.batchMagic = 0; // set per batch (renderBatch)
  c->hostTileIds.swap(tileIds);
  c->dynamic = false; // (rthip_begin_dynamic sets it again after this call)
  c->geom = g;
  c->camera = *camera;
  c->maxDepth = f->maxDepth;
  c->aovDepths = aovDepths;
  c->framePipelined = pipelinedFrame;
  c->accFresh = true; // CPURenderer.cpp:142-143: the clear is folded into the first k_accumulate (or done on a read before it)
  // Camera constants and counter totals are double-buffered per frame: at most two frames are in flight (one per slot; a
  // third queues behind the first on that slot's stream).  The prologue kernel goes to the stream the frame's first batch
  // will use, so a one-batch frame on one slot does not wait for the previous frame on the other; every batch of the frame
  // waits for evFrame.  A frame that never rendered keeps its parity (its buffers are nobody else's).
  if (c->renderedAny || !c->haveFrame)
    c->frameParity ^= 1u;
  const bool firstOnAlt = pipelinedFrame && (c->batchToggle & 1u);
  hipStream_t prologue = firstOnAlt ? c->alt.stream : c->stream;
  hipLaunchKernelGGL(k_camera, dim3(1), dim3(64), 0, prologue, *camera, frameCamera(c), frameTotals(c));
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->evFrame, prologue));
  if (firstOnAlt)
    c->altPending = true;
  c->renderedAny = false;
  c->haveFrame = true;
  // a no-op unless the scene, its shading switches or the frame's shape changed since the last measurement; when it does
  // measure, this call renders a few sample batches and waits for them (the one host synchronisation rthip_begin can make)
  if (int rcCal = calibrateRefill(c))
  {
    c->haveFrame = false; // the frame is not usable: the caller begins again
    return rcCal;
  }
  return 0;
}

// ---- dynamic tile ownership (BASELINE configs[4]): the tiles a rank renders are not known when the frame
// begins; they arrive in batches from a cursor shared by all ranks (any host-side atomic: raytracer_amd/dist.py TileQueue).
int rthip_begin_dynamic(rthip_ctx* c, const rthip_camera* camera, const rthip_frame_desc* f)
{
  if (!c || !camera || !f || f->width == 0 || f->height == 0)
    return RTHIP_E_ARG;
  if (f->aovDepths)
    return RTHIP_E_ARG; // AOV planes are indexed by a fixed local pixel order
  const uint32_t ts = f->tileSize ? f->tileSize : 64u;
  if (ts < 8u || (ts & (ts - 1u)))
    return RTHIP_E_ARG;
  const uint64_t allTiles = (uint64_t)((f->width + ts - 1) / ts) * ((f->height + ts - 1) / ts);
  const uint32_t cap = f->numTiles ? f->numTiles : (uint32_t)allTiles;
  if (cap == 0 || cap > allTiles)
    return RTHIP_E_ARG;
  // sized (and, the first time, calibrated) like a static frame over the first `cap` tiles
  std::vector<uint32_t> identity(cap);
  for (uint32_t k = 0; k < cap; ++k)
    identity[k] = k;
  rthip_frame_desc g = *f;
  g.numTiles = cap;
  g.tileIds = identity.data();
  if (int rc = rthip_begin(c, camera, &g))
    return rc;
  if (c->dynTable.n < 2 * (size_t)cap || c->dynStageWords < 2 * (size_t)cap)
  {
    // a larger table: nothing may still be reading the old one
    HIPCHK(hipStreamSynchronize(c->alt.stream));
    HIPCHK(hipStreamSynchronize(c->alt.stream2));
    HIPCHK(hipStreamSynchronize(c->stream2));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->altPending = false;
    if (int rc = c->dynTable.ensure(2 * (size_t)cap))
    {
      c->haveFrame = false;
      return rc;
    }
    if (c->dynStage)
      (void)hipHostFree(c->dynStage);
    c->dynStage = nullptr, c->dynStageWords = 0;
    if (hipHostMalloc((void**)&c->dynStage, sizeof(uint32_t) * 2 * (size_t)cap, hipHostMallocDefault) != hipSuccess)
    {
      c->haveFrame = false;
      return RTHIP_E_LIMIT;
    }
    c->dynStageWords = 2 * (size_t)cap;
  }
  for (hipEvent_t& e : c->dynEvents)
    if (!e)
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  c->dynamic = true;
  c->dynCapacity = cap;
  c->dynCount = 0;
  c->dynIds.clear();
  c->geom.tileXY = c->dynTable.p + (size_t)c->frameParity * cap;
  c->accFresh = false; // every tile batch starts its own pixels from zero (k_accumulate's `fresh`)
  return 0;
}

int rthip_enqueue_tiles(rthip_ctx* c, const uint32_t* tileIds, uint32_t numTiles)
{
  if (!c || !tileIds || !numTiles)
    return RTHIP_E_ARG;
  if (!c->haveScene || !c->haveFrame || !c->dynamic)
    return RTHIP_E_STATE;
  if ((uint64_t)c->dynCount + numTiles > c->dynCapacity)
    return RTHIP_E_LIMIT;
  const uint32_t ts = c->geom.tileSize, tilesX = c->geom.tilesX, tilesY = (c->geom.height + ts - 1) / ts;
  for (uint32_t k = 0; k < numTiles; ++k)
    if (tileIds[k] >= tilesX * tilesY)
      return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  // the ring of completion events is finite: retire the oldest call first when it is full
  if (c->dynIssued - c->dynRetired >= rthip_ctx::kDynEvents)
  {
    HIPCHK(hipEventSynchronize(c->dynEvents[c->dynRetired % rthip_ctx::kDynEvents]));
    c->dynRetired++;
  }
  uint32_t* stage = c->dynStage + (size_t)c->frameParity * c->dynCapacity + c->dynCount;
  for (uint32_t k = 0; k < numTiles; ++k)
    stage[k] = ((tileIds[k] % tilesX) * ts) | (((tileIds[k] / tilesX) * ts) << 16);
  const uint32_t first = c->dynCount;
  c->dynIds.insert(c->dynIds.end(), tileIds, tileIds + numTiles);
  c->dynCount += numTiles;
  if (int rc = renderPasses(c, 0, c->geom.numSamples, false, first, numTiles))
    return rc;
  // the call is complete when its last batch is (accumulates are chained, so that one finishes last)
  HIPCHK(hipEventRecord(c->dynEvents[c->dynIssued % rthip_ctx::kDynEvents], c->lastAccStream));
  c->dynIssued++;
  return 0;
}

int rthip_wait_tiles(rthip_ctx* c, uint32_t maxInFlight)
{
  if (!c)
    return RTHIP_E_ARG;
  if (!c->dynamic)
    return RTHIP_E_STATE;
  HIPCHK(hipSetDevice(c->device));
  while (c->dynIssued - c->dynRetired > maxInFlight)
  {
    HIPCHK(hipEventSynchronize(c->dynEvents[c->dynRetired % rthip_ctx::kDynEvents]));
    c->dynRetired++;
  }
  return 0;
}

int rthip_get_local_tiles(rthip_ctx* c, uint32_t* tileIdsOut, uint32_t capacity, uint32_t* countOut)
{
  if (!c || !countOut)
    return RTHIP_E_ARG;
  if (!c->haveFrame)
    return RTHIP_E_STATE;
  const std::vector<uint32_t>& ids = c->dynamic ? c->dynIds : c->hostTileIds;
  *countOut = (uint32_t)ids.size();
  if (tileIdsOut)
  {
    if (capacity < ids.size())
      return RTHIP_E_ARG;
    std::copy(ids.begin(), ids.end(), tileIdsOut);
  }
  return 0;
}

int rthip_set_refill_mode(rthip_ctx* c, int mode)
{
  if (!c || mode < -1 || mode > 3)
    return RTHIP_E_ARG;
  c->refillWanted = mode;
  c->refillKey = 0; // (back to -1: measure again at the next rthip_begin)
  return 0;
}

int rthip_get_grid_blocks(rthip_ctx* c, uint32_t* blocksPerCuOut)
{
  if (!c || !blocksPerCuOut)
    return RTHIP_E_ARG;
  *blocksPerCuOut = c->traceBlocks / (uint32_t)std::max(1, c->numCUs);
  return 0;
}

int rthip_get_refill_mode(rthip_ctx* c, int* modeOut, float msOut[4])
{
  if (!c)
    return RTHIP_E_ARG;
  if (modeOut)
    *modeOut = c->refillMode;
  if (msOut)
    msOut[0] = c->refillMs[0], msOut[1] = c->refillMs[1], msOut[2] = c->refillMs[2], msOut[3] = c->refillMs[3];
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

// the tiles the context holds pixels for: all local tiles of a static frame, the ones taken so far of a dynamic one
static uint32_t localTileCount(const rthip_ctx* c) { return c->dynamic ? c->dynCount : c->geom.numLocalTiles; }

static int readTiled(rthip_ctx* c, const float* tileMajorDev, float* rgbOut)
{
  const size_t full = (size_t)c->geom.width * c->geom.height * 3;
  const uint32_t numPixels = localTileCount(c) * c->geom.tileSize * c->geom.tileSize;
  if (numPixels == 0)
    return 0;
  int rc = c->rowScratch.ensure(full);
  if (rc)
    return rc;
  // pixels of tiles this context does not own keep the caller's values
  HIPCHK(hipMemcpyAsync(c->rowScratch.p, rgbOut, full * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_untile, dim3((numPixels + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->geom, tileMajorDev, c->rowScratch.p, numPixels);
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

// ---- multi-GPU assembly over RCCL (one gather of the fp32 accumulator at the end of a frame) ----------------
int rthip_comm_unique_id(unsigned char idOut[RTHIP_COMM_ID_BYTES])
{
  static_assert(RTHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id crosses the ABI as plain bytes");
  if (!idOut)
    return RTHIP_E_ARG;
  if (!rccl().ok)
    return RTHIP_E_COMM;
  ncclUniqueId id;
  NCCLCHK(rccl().getUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(idOut, id.internal, RTHIP_COMM_ID_BYTES);
  return 0;
}

int rthip_comm_init(rthip_ctx* c, const unsigned char id[RTHIP_COMM_ID_BYTES], int rank, int worldSize)
{
  if (!c || !id || worldSize < 1 || rank < 0 || rank >= worldSize)
    return RTHIP_E_ARG;
  if (c->comm)
    return RTHIP_E_STATE;
  if (!rccl().ok)
    return RTHIP_E_COMM;
  HIPCHK(hipSetDevice(c->device));
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, RTHIP_COMM_ID_BYTES);
  NCCLCHK(rccl().commInitRank(&c->comm, worldSize, uid, rank), "ncclCommInitRank");
  c->commRank = rank;
  c->commSize = worldSize;
  return 0;
}

int rthip_comm_destroy(rthip_ctx* c)
{
  if (!c)
    return RTHIP_E_ARG;
  if (!c->comm)
    return 0;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  NCCLCHK(rccl().commDestroy(c->comm), "ncclCommDestroy");
  c->comm = nullptr;
  c->commSize = 0;
  return 0;
}

int rthip_gather(rthip_ctx* c, float* recvDev, uint64_t floatsPerRank, int root)
{
  if (!c || !floatsPerRank)
    return RTHIP_E_ARG;
  if (!c->comm || !c->haveFrame)
    return RTHIP_E_STATE;
  if (root < 0 || root >= c->commSize || (c->commRank == root && !recvDev))
    return RTHIP_E_ARG;
  const uint64_t mine = (uint64_t)c->geom.numLocalPixels * 3;
  if (mine > floatsPerRank)
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (int rc = joinSlots(c)) // the frame's last batches may be on the second slot's stream
    return rc;
  if (int rc = settleAccumulator(c))
    return rc;
  // every rank sends the same number of floats: its tile-major accumulator, zero-padded when it owns fewer tiles
  const float* sendBuf = c->acc.p;
  if (mine < floatsPerRank)
  {
    if (int rc = c->gatherStage.ensure(floatsPerRank))
      return rc;
    HIPCHK(hipMemcpyAsync(c->gatherStage.p, c->acc.p, mine * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->gatherStage.p + mine, 0, (floatsPerRank - mine) * sizeof(float), c->stream));
    sendBuf = c->gatherStage.p;
  }
  // grouped point-to-point: over xGMI every peer has its own link to the root, so the transfers run side by side
  NCCLCHK(rccl().groupStart(), "ncclGroupStart");
  ncclResult_t r = rccl().send(sendBuf, floatsPerRank, ncclFloat, root, c->comm, c->stream);
  if (r == ncclSuccess && c->commRank == root)
    for (int k = 0; k < c->commSize && r == ncclSuccess; ++k)
      r = rccl().recv(recvDev + (size_t)k * floatsPerRank, floatsPerRank, ncclFloat, k, c->comm, c->stream);
  const ncclResult_t rEnd = rccl().groupEnd();
  if (r != ncclSuccess)
    return commFail(r, "ncclSend/ncclRecv");
  NCCLCHK(rEnd, "ncclGroupEnd");
  // The send reads the accumulator: a frame queued right behind this one may run its first accumulate on the other slot's
  // stream, so that launch is made to wait for the transfer the way it waits for the previous batch's accumulate.
  HIPCHK(hipEventRecord(c->evDone, c->stream));
  c->lastAcc = c->evDone;
  c->lastAccStream = c->stream;
  return 0; // queued on the context's stream: rthip_untile / rthip_synchronize order themselves behind it
}

// Variable tile ownership: every rank sends a header [tile count | tile ids] and then exactly the pixels of the tiles it holds;
// the root learns the counts from the headers (one read-back) and posts receives of matching sizes.  Works for static frames
// too (no zero padding travels).
int rthip_gather_tiles(rthip_ctx* c, float* recvDev, uint64_t recvCapacityFloats, uint32_t* tileIdsOut, uint32_t tileIdCapacity, uint32_t* tileCountsOut, int root)
{
  if (!c)
    return RTHIP_E_ARG;
  if (!c->comm || !c->haveFrame)
    return RTHIP_E_STATE;
  const bool isRoot = c->commRank == root;
  if (root < 0 || root >= c->commSize || (isRoot && (!recvDev || !tileIdsOut || !tileCountsOut)))
    return RTHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (int rc = joinSlots(c)) // the frame's last batches may be on the second slot's stream
    return rc;
  if (!c->dynamic)
    if (int rc = settleAccumulator(c))
      return rc;
  const uint32_t ts = c->geom.tileSize, tilesY = (c->geom.height + ts - 1) / ts;
  const uint32_t maxTiles = c->geom.tilesX * tilesY; // every rank computes the same header length from the frame
  const uint32_t hdrWords = 1 + maxTiles;
  const std::vector<uint32_t>& ids = c->dynamic ? c->dynIds : c->hostTileIds;
  const uint32_t mine = (uint32_t)ids.size();
  if (int rc = c->gatherHdr.ensure((size_t)hdrWords * (isRoot ? c->commSize + 1 : 1)))
    return rc;
  std::vector<uint32_t> hdr(hdrWords, 0u);
  hdr[0] = mine;
  std::copy(ids.begin(), ids.end(), hdr.begin() + 1);
  HIPCHK(hipMemcpyAsync(c->gatherHdr.p, hdr.data(), sizeof(uint32_t) * hdrWords, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream)); // (hdr is a local: the copy must have left it; the frame is finished by now anyway)
  // phase 1: headers
  NCCLCHK(rccl().groupStart(), "ncclGroupStart");
  ncclResult_t r = rccl().send(c->gatherHdr.p, hdrWords, ncclUint32, root, c->comm, c->stream);
  if (r == ncclSuccess && isRoot)
    for (int k = 0; k < c->commSize && r == ncclSuccess; ++k)
      r = rccl().recv(c->gatherHdr.p + (size_t)(k + 1) * hdrWords, hdrWords, ncclUint32, k, c->comm, c->stream);
  ncclResult_t rEnd = rccl().groupEnd();
  if (r != ncclSuccess)
    return commFail(r, "ncclSend/ncclRecv (tile headers)");
  NCCLCHK(rEnd, "ncclGroupEnd");
  const uint64_t perTile = (uint64_t)ts * ts * 3;
  std::vector<uint32_t> all;
  if (isRoot)
  {
    all.resize((size_t)hdrWords * c->commSize);
    HIPCHK(hipMemcpyAsync(all.data(), c->gatherHdr.p + hdrWords, sizeof(uint32_t) * all.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t total = 0;
    for (int k = 0; k < c->commSize; ++k)
    {
      const uint32_t n = all[(size_t)k * hdrWords];
      if (n > maxTiles)
        return RTHIP_E_COMM;
      total += n;
    }
    if (total > tileIdCapacity || total * perTile > recvCapacityFloats)
      return RTHIP_E_ARG; // (every rank still completes phase 2 below only if the root does: report before anything is posted)
  }
  // phase 2: the pixels, rank k's block after rank k-1's
  NCCLCHK(rccl().groupStart(), "ncclGroupStart");
  r = ncclSuccess;
  if (mine)
    r = rccl().send(c->acc.p, (size_t)mine * perTile, ncclFloat, root, c->comm, c->stream);
  if (r == ncclSuccess && isRoot)
  {
    uint64_t off = 0;
    for (int k = 0; k < c->commSize && r == ncclSuccess; ++k)
    {
      const uint32_t n = all[(size_t)k * hdrWords];
      tileCountsOut[k] = n;
      std::copy(all.begin() + (size_t)k * hdrWords + 1, all.begin() + (size_t)k * hdrWords + 1 + n, tileIdsOut + off);
      if (n)
        r = rccl().recv(recvDev + off * perTile, (size_t)n * perTile, ncclFloat, k, c->comm, c->stream);
      off += n;
    }
  }
  rEnd = rccl().groupEnd();
  if (r != ncclSuccess)
    return commFail(r, "ncclSend/ncclRecv (tile pixels)");
  NCCLCHK(rEnd, "ncclGroupEnd");
  // the send reads the accumulator: the next frame's first accumulate waits for it (as after rthip_gather)
  HIPCHK(hipEventRecord(c->evDone, c->stream));
  c->lastAcc = c->evDone;
  c->lastAccStream = c->stream;
  return 0; // queued on the context's stream: rthip_untile / rthip_synchronize order themselves behind it
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

int rthip_set_overlap_shadow(rthip_ctx* c, int enabled)
{
  if (!c)
    return RTHIP_E_ARG;
  c->overlapShadow = enabled ? 1u : 0u;
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

// The allocation-free variant of rthip_probe_trace (no counterpart in the reference; for callers that cannot allocate).
// Same kernels, same launches; every buffer beyond the caller's inputs and outputs lives in this GPU's static `scratch`.
int rthip_trace_rays_static(rthip_ctx* c, const float* originDev3, const float* dirDev3, uint32_t n, float tmin, float tmax,
                            uint32_t* triIdDev, float* tDev, uint8_t* anyDev)
{
  if (!c || !originDev3 || !dirDev3 || !n || !triIdDev || !tDev || !anyDev)
    return RTHIP_E_ARG;
  if (!c->haveScene)
    return RTHIP_E_STATE;
  if (n > kStaticMaxRays)
    return RTHIP_E_LIMIT; // "rthip: size beyond a compiled-in limit": at most RTHIP_STATIC_MAX_RAYS rays per call
  if (c->spillCap > kStaticSpillCap || c->device >= kMaxDevices || !g_scratchPtr[c->device])
    return RTHIP_E_LIMIT;
  HIPCHK(hipSetDevice(c->device));
  unsigned char* base = g_scratchPtr[c->device]; // the resolved device address, never the symbol's name
  // the context's own buffers step aside for the duration of the call (restored on every return path)
  Borrow<float4> bRayO(c->rayO, base + kStaticOffRayO, kStaticMaxRays), bRayD(c->rayD, base + kStaticOffRayD, kStaticMaxRays);
  Borrow<uint2> bHit(c->hit, base + kStaticOffHit, kStaticMaxRays);
  Borrow<uint8_t> bLive(c->live, base + kStaticOffLive, kStaticMaxRays + 64), bAny(c->anyOut, anyDev, n);
  Borrow<uint32_t> bFetch(c->fetch, base + kStaticOffFetch, kStaticFetchWords), bSpill(c->spill, base + kStaticOffSpill, kStaticSpillWords);
  const uint32_t keepCursor = c->fetchCursor;
  const bool keepTiming = c->kernelTiming;
  c->kernelTiming = false; // the kernel timer creates events on demand: not in a call that must not allocate
  struct Restore
  {
    rthip_ctx* c;
    uint32_t cursor;
    bool timing;
    uint32_t launches[6];
    // (the kernel timer counts launches per family even when it records no events: a call in the middle of a frame must
    // leave what rthip_get_kernel_times reports for that frame alone)
    ~Restore()
    {
      c->fetchCursor = cursor, c->kernelTiming = timing;
      for (int k = 0; k < 6; ++k)
        c->kernelLaunches[k] = launches[k];
    }
  } restore{c, keepCursor, keepTiming, {}};
  for (int k = 0; k < 6; ++k)
    restore.launches[k] = c->kernelLaunches[k];
  // the array keeps the previous call's contents: counters and work-fetch cursors must start at zero; ray records, hit
  // records and liveness bytes of the n slots in use are all rewritten below before anything reads them
  HIPCHK(hipMemsetAsync(c->fetch.p, 0, kStaticFetchWords * sizeof(uint32_t), c->stream));
  c->fetchCursor = 0;
  hipLaunchKernelGGL(k_probe_pack, dim3((n + 255) / 256), dim3(256), 0, c->stream, originDev3, dirDev3, n, c->rayO.p, c->rayD.p, c->live.p);
  HIPCHK(hipGetLastError());
  int rc;
  if ((rc = launchTrace<CLOSEST>(c, n, 0, 0, tmin, tmax)) || (rc = launchTrace<ANYPROBE>(c, n, 0, 0, tmin, tmax)))
    return rc;
  hipLaunchKernelGGL(k_probe_unpack, dim3((n + 255) / 256), dim3(256), 0, c->stream, devScene(c), c->hit.p, n, triIdDev, tDev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
}
