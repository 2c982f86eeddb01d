/*
 * rthip.h -- the thin C ABI between the host C++ backend (raytracer_amd/src/hip) and the hand-written
 * gfx950 kernels (raytracer_amd/csrc).  Built into raytracer_amd/librthip.so.
 *
 * The reference has no FFI: a backend is a C++ subclass of Renderer/Scene chosen at construction
 * (src/main.cpp:8).  These entry points are what the reference-side `src/hip/HIPRenderer.cpp` /
 * `HIPScene.cpp` bind; each cites the reference code whose work it takes over.
 * Citations are relative to /root/reference.
 *
 * Conventions: plain pointers and sizes only; every call returns 0 on success, a negative RTHIP_E_* code for
 * an argument/state error, or a positive hipError_t; nothing throws across the ABI; all host pointers are
 * caller-owned and copied from/to before the call returns; a context is bound to one device and may be used
 * from one host thread at a time (every call sets the device first, because the caller is the
 * Renderer::worker thread, src/scene/Renderer.cpp:20).
 */
#ifndef RTHIP_H
#define RTHIP_H
#include <stdint.h>
#include "rtflat.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTHIP_E_ARG (-1)       /* null pointer / zero size / inconsistent sizes */
#define RTHIP_E_STATE (-2)     /* call out of order (no scene, no begin) */
#define RTHIP_E_BOUNDS (-3)    /* camera outside rtflat_info.originBound: box padding would not be conservative */
#define RTHIP_E_NODEVICE (-4)  /* no usable HIP device */
#define RTHIP_E_LIMIT (-5)     /* size beyond a compiled-in maximum */
#define RTHIP_E_COMM (-6)      /* RCCL could not be loaded or returned an error (rthip_error_string has its text) */

typedef struct rthip_ctx rthip_ctx;

/* same fields as the reference's Camera (src/util/Camera.h:4-12) */
typedef struct rthip_camera
{
  float position[3];
  float target[3];
  float sensorSize[2];
  float S_O, f, A;
} rthip_camera;

/* 32-byte light records, same layout as src/scene/Scene.h:14-28 */
typedef struct rthip_point_light
{
  float position[3], pad, color[3], attenuation;
} rthip_point_light;
typedef struct rthip_dir_light
{
  float direction[3], pad, color[3], pad1;
} rthip_dir_light;

typedef struct rthip_frame_desc
{
  uint32_t width, height, numSamples; /* RenderParameter (src/scene/Renderer.h:7-12) */
  uint32_t maxDepth;                  /* path depth cap; the reference hard-codes 12 (src/cpu/CPUScene.cpp:15) */
  uint32_t aovDepths;                 /* record hit-id / shadow AOVs for this many depths of each pass (0 = off) */
  uint32_t tileSize;                  /* square tile edge in pixels (a power of two >= 8); 0 = default 64 */
  uint32_t numTiles;                  /* tiles this context renders; 0 = the whole image (every tile: the even tile rows top to bottom,
                                         then the odd ones).  The list's order is the order of the tiles' pixel blocks in the tile-major
                                         accumulator (rthip_get_local_tiles reports it), and a frame cut into batches takes contiguous
                                         runs of it: order an explicit list so that its halves cost about the same (hf10m: the upper
                                         and the lower image half as the two batches cost 4 % of the frame against interleaved rows) */
  uint32_t pad;
  const uint32_t* tileIds;            /* numTiles tile indices (row-major over ceil(W/ts) x ceil(H/ts)); NULL with numTiles 0 */
} rthip_frame_desc;

typedef struct rthip_counters
{
  uint64_t extendPrimary;   /* closest-hit rays cast at depth 0 */
  uint64_t extendSecondary; /* closest-hit rays cast at depth >= 1 */
  uint64_t shadow;          /* any-hit rays cast */
  uint64_t nodesVisited;    /* 64-B node records fetched (only counted when built with RTHIP_COUNT_WORK or profile mode on) */
  uint64_t trisTested;
  uint64_t nonFiniteAccepts;/* triangle tests that passed every rejection with a NaN: outside the parity domain */
  uint64_t hits[16];        /* closest hits per depth */
  uint64_t nodesVisitedExtend; /* the closest-hit share of nodesVisited / trisTested (the rest is any-hit) */
  uint64_t trisTestedExtend;
  uint64_t waveNodeSteps;   /* work counting: node steps / leaf steps executed by whole waves; x64 = lane slots offered, */
  uint64_t waveLeafSteps;   /* so nodesVisited / (64*waveNodeSteps) is the SIMD utilisation of the node step */
  /* work counting: 64-byte scene records (nodes + triangles) the traversal launches FETCHED.  A per-lane launch fetches every
   * record a ray visits, so these equal nodesVisited + trisTested of its rays; a packet launch fetches a record once for the
   * 64 rays of a wave, however many of them visit it.  These are the bytes the algorithm has to move. */
  uint64_t recordsFetchedExtend;
  uint64_t recordsFetchedShadow;
} rthip_counters;

/* creates a context on `device` (a HIP device ordinal); replaces nothing in the reference (it has no device) */
int rthip_create(int device, rthip_ctx** out);
int rthip_destroy(rthip_ctx* ctx);

/* Uploads the flattened scene (copies).  Takes over CPUScene::createRayTracingHierarchy's result
 * (src/cpu/CPUScene.cpp:67-105) and the pools intersectModel reads (src/scene/Scene.h:45-51).
 * Lights: up to 32 directional lights (each casts a shadow ray per hit, one occlusion bit per light and path) and up to
 * 65536 point lights; beyond that RTHIP_E_LIMIT (the reference's vectors are unbounded, src/scene/Scene.h:53-54). */
int rthip_upload_scene(rthip_ctx* ctx, const rtflat_info* info, const rtflat_model_node* modelNodes, const rtflat_node* nodes,
                       const rtflat_tri* tris, const rthip_dir_light* dirLights, uint32_t numDirLights,
                       const rthip_point_light* pointLights, uint32_t numPointLights);

/* Same, but the triangle-level BVH is BUILT ON THE DEVICE (Morton codes, radix sort, binary radix tree, bottom-up refit,
 * collapse into the same quantised 4-wide rtflat_node records).  The caller passes what precedes that step in
 * CPUScene::createRayTracingHierarchy's place (src/cpu/CPUScene.cpp:67-105): the model-level tree, whose leaves name their
 * triangles as records [indicesOffset, indicesOffset + triCount) of `trisModelOrder` (one rtflat_tri per triangle in the
 * order of the scene's pools), and info with numModelNodes, numTris, scene box, padAbs, originBound, cullRel/cullAbs set.
 * On return modelNodes[].blas, info->numNodes and info->maxDepthBlas describe the tree that was built (rthip_download_scene
 * fetches the records); buildMsOut (may be NULL) is the device time of the build.  Hit results do not depend on which
 * builder made the tree. */
int rthip_upload_scene_build(rthip_ctx* ctx, rtflat_info* info, rtflat_model_node* modelNodes, const rtflat_tri* trisModelOrder,
                             uint32_t maxLeafTris, const rthip_dir_light* dirLights, uint32_t numDirLights,
                             const rthip_point_light* pointLights, uint32_t numPointLights, float* buildMsOut);
/* how rthip_upload_scene_build forms the binary tree it then collapses: 0 (default) = Morton-order radix tree (LBVH; the code's
 * bits are dealt to the axes longest-extent-first), 1 = PLOC, parallel locally-ordered clustering over the Morton order
 * (nearest-neighbour merging by surface area; measured 4-7 % slower to trace than 0 on the benchmark scenes, kept as an
 * experiment).  Either way a subtree ends in a leaf where the surface-area heuristic says so.  Hit results do not depend on it. */
int rthip_set_build_quality(rthip_ctx* ctx, int quality);
/* Moved vertices, same triangles: every record takes the new operands of its triangle (trisModelOrder: one rtflat_tri per
 * triangle in pool order, as for rthip_upload_scene_build; triId / orderKey are kept) and the boxes of the triangle-level tree
 * are recomputed bottom-up and quantised afresh; the model-level nodes take the boxes of modelNodes (same tree, numModelNodes
 * entries: the reference's Model::transform boxes of the moved models, src/util/Model.cpp:3-26).  Topology and leaf contents
 * stay, so the tree is as good as the motion is small.  The new triangles must stay within one scene diagonal of the box the
 * scene was flattened for (rtflat_info.sceneLo/Hi: the padding of every box was sized for coordinates that far out); otherwise
 * RTHIP_E_BOUNDS and the scene has to be uploaded again.  Waits for frames in flight, returns when done; msOut (may be NULL): device milliseconds.  The reference has
 * no counterpart (its CPU path keeps no triangle-level tree). */
int rthip_refit(rthip_ctx* ctx, const rtflat_model_node* modelNodes, const rtflat_tri* trisModelOrder, float* msOut);
/* copies the scene's node / triangle records back from the device (either pointer may be NULL) */
int rthip_download_scene(rthip_ctx* ctx, rtflat_node* nodesOut, uint32_t nodeCapacity, rtflat_tri* trisOut, uint32_t triCapacity);

/* Materials on the hot path.  Until this is called every hit has the albedo (0,1,0) and emissive 0 that
 * the reference's intersector hard-codes (src/cpu/CPUScene.cpp:243-247); afterwards a hit's BRDF is its model's material
 * evaluated at the interpolated texture coordinates (the evident intent of src/util/Material.cpp:4-12 and
 * src/util/Texture.cpp:3-8, which the reference never wires in): albedo drives the throughput, the russian-roulette
 * probability (CPUScene.cpp:13) and the direct light (BRDF.cpp:4-14), emissive is added where the camera sees it
 * (CPUScene.cpp:27).  `shade` has one record per triangle of the uploaded scene, indexed by triId; `texels` holds
 * numTexels RGB float triples.  Copies everything; a later rthip_upload_scene* drops the materials again. */
int rthip_upload_materials(rthip_ctx* ctx, const rtflat_shade* shade, uint32_t numTris, const rtflat_material* materials,
                           uint32_t numMaterials, const float* texels, uint64_t numTexels);
/* on != 0: point lights cast shadow rays too, the way the reference's Metal kernel does (src/metal/Compute.metal:238-254:
 * origin = hit + normal * 1e-3, direction = light - hit, t in [0, 1]); the CPU path leaves them unshadowed
 * (src/cpu/CPUScene.cpp:42, commented out) and so does this backend by default.  Occlusion bits of the point lights
 * follow those of the directional lights in the path's 32-bit mask (the shadow AOV shows the low 8), so on requires
 * numDir + numPoint <= 32 (RTHIP_E_LIMIT). */
int rthip_set_point_shadows(rthip_ctx* ctx, int on);

/* How the traversal launches treat the 64 rays of a wave ("refill rule" and "packet traversal"):
 * 0 = every lane walks its own ray and idle lanes are refilled early (suits incoherent rays), 1 = the same but a wave runs
 * nearly empty before it takes new rays (suits coherent rays), 2 = camera rays and directional shadow rays are traced as
 * packets -- the wave visits one node at a time for all its rays -- and bounce rays as in 1 (suits very coherent rays: many
 * samples per pixel), 3 = packets for the camera rays only; -1 (default) = measured: the first rthip_begin after a scene upload or a change of the frame's shape
 * traces a sample of the frame with each setting (tens of milliseconds, one host synchronisation) and keeps the fastest.
 * Results do not depend on it.  rthip_get_refill_mode reports the setting in use and the timings in ms (0 when none was
 * taken): msOut[0..2] = settings 0, 1, 2; msOut[3] is unused (0).  rthip_get_grid_blocks reports the resident blocks per CU
 * of the launches of depth >= 1 (3; camera-ray launches use 6).  No counterpart in the reference (tuning only). */
int rthip_set_refill_mode(rthip_ctx* ctx, int mode);
int rthip_get_refill_mode(rthip_ctx* ctx, int* modeOut, float msOut[4]);
int rthip_get_grid_blocks(rthip_ctx* ctx, uint32_t* blocksPerCuOut);

/* Starts a frame: allocates and clears accumulator/image for the tiles of `frame`
 * (CPURenderer::render's prologue, src/cpu/CPURenderer.cpp:140-143). */
int rthip_begin(rthip_ctx* ctx, const rthip_camera* camera, const rthip_frame_desc* frame);

/* Renders sample passes [sampBegin, sampBegin+count) : ray generation, <= maxDepth rounds of extend / shade /
 * shadow, accumulation in `samp` order (the body of the `samp` loop, src/cpu/CPURenderer.cpp:144-206 with
 * src/cpu/CPUScene.cpp:6-65 inside).  msOut (may be NULL): device time of the whole call in ms, measured with
 * HIP events on the context's stream. Blocks until the passes are done. */
int rthip_render_passes(rthip_ctx* ctx, uint32_t sampBegin, uint32_t count, float* msOut);
/* same, but only enqueues on the context's streams: no host synchronisation at any depth cap (path counts stay on the
 * device), no timing */
int rthip_enqueue_passes(rthip_ctx* ctx, uint32_t sampBegin, uint32_t count);
int rthip_synchronize(rthip_ctx* ctx);

/* Row-major width*height*3 floats (src/cpu/CPURenderer.cpp:197-198 layout); pixels of tiles this context does
 * not own are left untouched in `out`. */
int rthip_read_accumulator(rthip_ctx* ctx, float* rgbOut);
int rthip_read_image(rthip_ctx* ctx, float* rgbOut);
/* AOVs of sample pass `samp` (must be one of the passes of the LAST render call), depth < aovDepths:
 * triIdOut[h*width+w] = indicesOffset+posIndex of the closest hit or 0xffffffff; shadowOut bit l = directional
 * light l occluded, 0xff = no hit.  The reference keeps no ids (HitInfo has none, src/util/Ray.h:19-28). */
int rthip_read_aov(rthip_ctx* ctx, uint32_t samp, uint32_t depth, uint32_t* triIdOut, uint8_t* shadowOut);

/* Device-side access for the multi-GPU gather: the context's accumulator in its own (tile-major) order.
 * *devPtr points at numPixels*3 floats; pixel k of local tile j is at ((j*tileSize*tileSize)+k)*3. */
int rthip_accumulator_device(rthip_ctx* ctx, float** devPtr, uint64_t* numFloats);
/* copies the context's tile-major accumulator (numFloats floats as reported above) into caller-owned DEVICE memory
 * on the same GPU (e.g. a torch tensor's data_ptr feeding an RCCL gather) and waits for the copy */
int rthip_copy_accumulator_to(rthip_ctx* ctx, float* dstDev, uint64_t numFloats);
/* scatters a gathered tile-major buffer (device memory, tiles listed in tileIds) into a row-major
 * width*height*3 device image */
int rthip_untile(rthip_ctx* ctx, const float* tileMajorDev, const uint32_t* tileIds, uint32_t numTiles, float* rowMajorDev);

/* Multi-GPU assembly over RCCL (one process per GPU, full scene replica per rank, image tiles as the shard).
 * The reference has no counterpart (single process, src/main.cpp); these take the place of the per-frame GL upload
 * (src/cpu/CPURenderer.cpp:127-136) on a multi-GPU host.  RCCL is loaded at run time (dlopen), not linked.
 *  rthip_comm_unique_id: one rank creates the communicator id (ncclGetUniqueId) and hands the bytes to the others by any side
 *                        channel (a file, MPI, a torch.distributed store);
 *  rthip_comm_init:      every rank joins (ncclCommInitRank) with its context, i.e. its GPU;
 *  rthip_gather:         every rank sends its tile-major accumulator, zero-padded to floatsPerRank floats, to `root` with grouped
 *                        ncclSend / ncclRecv on the context's stream; on root recvDev (device memory on its GPU, worldSize *
 *                        floatsPerRank floats) receives rank k's block at k * floatsPerRank.  Enqueue-only: rthip_untile on
 *                        the root context (same stream) or rthip_synchronize orders itself behind it. */
#define RTHIP_COMM_ID_BYTES 128
int rthip_comm_unique_id(unsigned char idOut[RTHIP_COMM_ID_BYTES]);
int rthip_comm_init(rthip_ctx* ctx, const unsigned char id[RTHIP_COMM_ID_BYTES], int rank, int worldSize);
int rthip_comm_destroy(rthip_ctx* ctx);
int rthip_gather(rthip_ctx* ctx, float* recvDev, uint64_t floatsPerRank, int root);

/* Dynamic tile ownership (BASELINE configs[4]: "dynamic tile rebalancing"): the ranks draw tiles from a shared
 * cursor while the frame renders, so a rank whose tiles were cheap takes more.  The cursor is the host's business (any atomic
 * counter every rank can reach; raytracer_amd/dist.py uses a torch.distributed store); the C ABI carries the rest.  A whole tile
 * -- all its sample passes -- is rendered by one rank, so per-pixel accumulation order, and with it every bit of the frame, is
 * what a single GPU produces (src/cpu/CPURenderer.cpp:197).  No counterpart in the reference (one process).
 *  rthip_begin_dynamic:  a frame whose tiles are not known yet.  frame->numTiles = the most tiles this rank may end up with
 *                        (0 = all tiles of the image; buffers are sized for it), frame->tileIds is ignored, aovDepths must be 0.
 *  rthip_enqueue_tiles:  queues ALL sample passes of these tiles on the context's streams and returns (no host synchronisation;
 *                        consecutive calls alternate between the two sets of path buffers, so two tile batches are in flight).
 *                        The tiles' pixels follow those of earlier calls in the context's tile-major accumulator.
 *  rthip_wait_tiles:     blocks until at most maxInFlight rthip_enqueue_tiles calls are unfinished: the back-pressure that makes
 *                        the drawing dynamic (a rank asks the cursor for more only when it is about to run dry; 2 keeps both slots busy).
 *  rthip_get_local_tiles: the tile ids this context holds pixels for, in accumulator order (static frames: the frame's tile list).
 *  rthip_gather_tiles:   the assembly for variable ownership: every rank sends a header [count | tile ids] and then exactly its
 *                        tiles' pixels (grouped ncclSend / ncclRecv on the context's stream); on root, rank k's pixels follow rank
 *                        k-1's in recvDev (device memory, recvCapacityFloats >= all tiles of the frame x tileSize^2 x 3),
 *                        tileIdsOut (host, tileIdCapacity >= all tiles) lists the tiles in the same order and tileCountsOut
 *                        (host, worldSize entries) says how many each rank brought; one rthip_untile(recvDev, tileIdsOut, total)
 *                        then writes the frame.  The root reads the headers back (one synchronisation per frame); other ranks
 *                        only enqueue.  Capacities too small for the frame are the caller's error on EVERY rank (checked on root
 *                        before any pixel is posted, but the other ranks' sends then never complete). */
int rthip_begin_dynamic(rthip_ctx* ctx, const rthip_camera* camera, const rthip_frame_desc* frame);
int rthip_enqueue_tiles(rthip_ctx* ctx, const uint32_t* tileIds, uint32_t numTiles);
int rthip_wait_tiles(rthip_ctx* ctx, uint32_t maxInFlight);
int rthip_get_local_tiles(rthip_ctx* ctx, uint32_t* tileIdsOut, uint32_t capacity, uint32_t* countOut);
int rthip_gather_tiles(rthip_ctx* ctx, float* recvDev, uint64_t recvCapacityFloats, uint32_t* tileIdsOut, uint32_t tileIdCapacity,
                       uint32_t* tileCountsOut, int root);

int rthip_get_counters(rthip_ctx* ctx, rthip_counters* out); /* totals since rthip_begin */
/* 1: count nodesVisited/trisTested in the traversal kernels (slower); 0: off (default) */
int rthip_set_work_counting(rthip_ctx* ctx, int enabled);
/* 1 (default): primary rays are generated inside the depth-0 extend kernel and never stored;
 * 0: a stand-alone generate kernel writes them first (the A/B baseline) */
int rthip_set_fused_primary(rthip_ctx* ctx, int enabled);
/* 1 (default): launch batches alternate between two sets of path buffers and streams (two batches in flight, at any depth
 * cap, across calls and frames); 0: one batch at a time (the A/B baseline; results are bit-identical).  Takes effect at
 * the next rthip_begin. */
int rthip_set_pipelining(rthip_ctx* ctx, int enabled);
/* 1 (default): a depth's shadow and closest-hit launches go to two streams (their tails overlap); 0: one after the other
 * (with rthip_set_pipelining(ctx, 0) every kernel then runs alone: the A/B baseline and the way to time a kernel by itself) */
int rthip_set_overlap_shadow(rthip_ctx* ctx, int enabled);
/* per-kernel device milliseconds and launch counts accumulated since timing was last switched on (HIP event pairs recorded
 * around every launch on the stream it goes to, resolved when the times are read, so batches and frames overlap exactly as
 * they do untimed): order generate, extend, shade, shadow, accumulate, compact.  rthip_set_kernel_timing(ctx,1) starts a
 * fresh measurement; rthip_get_kernel_times waits for the device. */
int rthip_set_kernel_timing(rthip_ctx* ctx, int enabled);
int rthip_get_kernel_times(rthip_ctx* ctx, float msOut[6], uint32_t launchesOut[6]);

/* bit-exactness probes of the device arithmetic against the host (tests only): n inputs each */
int rthip_probe_rand01(rthip_ctx* ctx, const uint32_t* xyz, uint32_t n, float* out3);
int rthip_probe_sincos(rthip_ctx* ctx, const float* x, uint32_t n, float* sinOut, float* cosOut);
int rthip_probe_primary_rays(rthip_ctx* ctx, const rthip_camera* cam, uint32_t width, uint32_t height, const uint32_t* whs,
                             uint32_t n, float* origin3, float* dir3);
int rthip_probe_bounce(rthip_ctx* ctx, const float* normalLight3, const float* rnd3, uint32_t n, float* dir3);
/* closest / any hit of n explicit rays (o,d as 3 floats each) through the uploaded scene */
int rthip_probe_trace(rthip_ctx* ctx, const float* origin3, const float* dir3, uint32_t n, float tmin, float tmax,
                      uint32_t* triIdOut, float* tOut, uint8_t* anyOut);

/* Allocation-free variant of rthip_probe_trace.  It has NO counterpart in the reference; it exists for callers that cannot
 * allocate: tight serving loops, harnesses that forbid hipMalloc.  (It is NOT capturable into a HIP graph: it selects the
 * context's device, works on the context's own stream and returns only after hipStreamSynchronize.)  Same kernels and results
 * as rthip_probe_trace, but every pointer is DEVICE memory on the context's GPU (origins / directions: 3 floats per ray;
 * outputs: n entries each) and all working memory (ray and hit records, liveness bytes, work counters, traversal-stack
 * spill) lives in a static device array of about 6.3 MiB that each GPU the library runs on holds until the process ends.
 * The call performs no hipMalloc / hipFree (nor any other allocating routine) and returns once the outputs are written.
 * Limits: n <= RTHIP_STATIC_MAX_RAYS, otherwise RTHIP_E_LIMIT ("size beyond a compiled-in limit").  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (two contexts, two threads) race; serialise them. */
#define RTHIP_STATIC_MAX_RAYS 16384u
int rthip_trace_rays_static(rthip_ctx* ctx, const float* originDev3, const float* dirDev3, uint32_t n, float tmin, float tmax,
                            uint32_t* triIdDev, float* tDev, uint8_t* anyDev);

const char* rthip_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif
