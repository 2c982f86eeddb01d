/*
 * raytracer_c.h -- C ABI over the host-side C++ backend interface (raytracer_amd/libraytracer_amd.so).
 *
 * The reference selects a backend by constructing a Renderer subclass in C++ (src/main.cpp:8) and drives it
 * through the virtual interface of src/scene/Renderer.h:14-42.  This header exposes exactly that interface as
 * plain C functions so that foreign hosts (ctypes in tests/bench.py, or any FFI) can drive the `hip` backend;
 * every function names the reference member it forwards to.  Return value: 0, a negative RTHIP_E_* code or a
 * positive hipError_t (see rthip.h / rthip_error_string).
 */
#ifndef RAYTRACER_C_H
#define RAYTRACER_C_H
#include <stdint.h>
#include "rtflat.h"
#include "rthip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_renderer rt_renderer;
typedef struct rt_mesh rt_mesh;

/* ---- meshes: what ModelLoader::loadModel returns (src/util/ModelLoader.cpp:8-43) ---- */
rt_mesh* rt_mesh_cube(void);                                                   /* stand-in for res/models/cube.fbx */
rt_mesh* rt_mesh_soup(uint32_t numTriangles, uint64_t seed, float edgeRadius); /* BASELINE.md config 3 */
rt_mesh* rt_mesh_heightfield(uint32_t quadsPerSide, uint64_t seed, float amplitude); /* BASELINE.md config 4/5 */
rt_mesh* rt_mesh_load_obj(const char* path);                                   /* first model of the file; NULL on failure */
/* every model of a file (.obj: one; .glb: one per mesh primitive, like the loop over aiScene::mMeshes): returns how many the
 * file holds and stores up to `capacity` of them in out[] (call with out == NULL to size the array) */
uint32_t rt_mesh_load_all(const char* path, rt_mesh** out, uint32_t capacity);
rt_mesh* rt_mesh_from_arrays(const float* positions, uint32_t numPositions, const float* texCoords, const uint32_t* indices,
                             uint32_t numTriangles);
uint32_t rt_mesh_num_positions(const rt_mesh*);
uint32_t rt_mesh_num_triangles(const rt_mesh*);
void rt_mesh_copy(const rt_mesh*, float* positions3, uint32_t* indices3);
void rt_mesh_copy_texcoords(const rt_mesh*, float* texCoords2);               /* Model::texCoords, one pair per position */
/* The mesh's Material (src/util/Material.h:7-14), read by the path only when the scene option "materials" is set.
 * kind 0 = Material::evaluate as the reference has it (src/util/Material.cpp:4-12: albedo = (u, v, 0), emissive 0; factors
 * and textures ignored); kind 1 = albedo = baseColor [* albedo texture], emissive = emissive [* emissive texture].
 * Textures are w*h RGB float triples, row-major, sampled like src/util/Texture.cpp:3-8 (NULL = none); copied.
 * A mesh that never had a material set has kind 1, baseColor (0,1,0), emissive 0: the constants of CPUScene.cpp:245-246.
 * Loaded .glb meshes arrive with their primitive's baseColorFactor / emissiveFactor (images are not decoded). */
int rt_mesh_set_material(rt_mesh*, uint32_t kind, const float baseColor[3], const float emissive[3], const float* albedoTexels,
                         uint32_t albedoW, uint32_t albedoH, const float* emissiveTexels, uint32_t emissiveW, uint32_t emissiveH);
int rt_mesh_get_material(const rt_mesh*, uint32_t* kind, float baseColor[3], float emissive[3]);
void rt_mesh_destroy(rt_mesh*);

/* ---- renderer: Renderer (src/scene/Renderer.h:14-42) with backend "hip" ---- */
int rt_renderer_create(const char* backend, int device, rt_renderer** out);   /* std::make_unique<...Renderer>(), main.cpp:8 */
void rt_renderer_destroy(rt_renderer*);
/* lights and meshes are accepted until rt_generate (RTHIP_E_STATE afterwards: they are uploaded with the scene) */
int rt_add_directional_light(rt_renderer*, const float direction[3], const float color[3]); /* Renderer::addDirectionalLight */
int rt_add_point_light(rt_renderer*, const float position[3], const float color[3], float attenuation); /* ::addPointLight */
/* Renderer::addModel(PModel, glm::mat4): the mesh is copied, transform is 16 floats column-major */
int rt_add_mesh(rt_renderer*, const rt_mesh*, const float transform[16]);
int rt_generate(rt_renderer*);                                                 /* Renderer::generate */

/* options the reference hard-codes: name in {"maxDepth","aovDepths","tileSize","passesPerCall","chunkMs",
 * "referenceEdgeIndexing","maxLeafTriangles","sahBins","traversalCost","buildThreads","originBoundScale","materials" (1: a hit's albedo / emissive come
 * from its mesh's material instead of the constants of src/cpu/CPUScene.cpp:243-247), "pointShadows" (1: point lights cast shadow
 * rays as in src/metal/Compute.metal:238-254; needs <= 32 lights in all), "deviceBvh" (1: the triangle-level BVH is
 * built on the device, rthip_upload_scene_build, as a Morton LBVH with SAH leaves instead of by the host's binned-SAH builder; 2: the same with PLOC clustering),"countWork","kernelTiming","fusedPrimary","pipelining","overlapShadow",
 * "refillMode" (-1 measured (default), 0 / 1 / 2 / 3 fixed: rthip_set_refill_mode)};
 * scene options must be set before rt_generate */
int rt_set_option(rt_renderer*, const char* name, double value);
int rt_set_tiles(rt_renderer*, const uint32_t* tileIds, uint32_t numTiles);    /* tiles this renderer owns; 0 = all */

int rt_start_render(rt_renderer*, const rthip_camera* camera, uint32_t width, uint32_t height, uint32_t numSamples); /* ::startRender */
int rt_wait_render(rt_renderer*);     /* joins the render thread; returns the render's error code */
int rt_cancel_render(rt_renderer*);   /* running = false + join (Renderer.cpp:13-17); returns within about one chunk ("chunkMs", 50 ms) of device work */
/* synchronous stepping for hosts that run the `samp` loop themselves (src/cpu/CPURenderer.cpp:144): rt_begin_frame
 * is render()'s prologue (:140-143), rt_render_passes runs passes [begin, begin+count) on the calling thread and
 * reports the device time of the call */
int rt_begin_frame(rt_renderer*, const rthip_camera* camera, uint32_t width, uint32_t height, uint32_t numSamples);
int rt_render_passes(rt_renderer*, uint32_t begin, uint32_t count, float* msOut);
/* the same without waiting for the device: passes -- and whole frames, rt_begin_frame included -- queue up behind each other;
 * rt_synchronize (or any read-back call) waits for everything queued so far */
int rt_enqueue_passes(rt_renderer*, uint32_t begin, uint32_t count);
int rt_synchronize(rt_renderer*);
uint32_t rt_get_sample_times(rt_renderer*, float* msOut, uint32_t cap);        /* ::getSampleTimes */
int rt_read_accumulator(rt_renderer*, float* rgbOut);                          /* CPURenderer::accumulator */
int rt_read_image(rt_renderer*, float* rgbOut);                                /* CPURenderer::image */
/* headless stand-in for the reference's window (CPURenderer.cpp:118-136 uploads `image` to a GL texture every frame): the
 * tonemapped image of the current frame as a binary PPM (P6, 8 bits per channel, values clamped to [0,1]) */
int rt_save_image(rt_renderer*, const char* path, uint32_t width, uint32_t height);
int rt_write_ppm(const char* path, const float* rgb, uint32_t width, uint32_t height); /* the encoder alone (any row-major RGB floats) */
int rt_read_aov(rt_renderer*, uint32_t samp, uint32_t depth, uint32_t* triIdOut, uint8_t* shadowOut);
int rt_get_counters(rt_renderer*, rthip_counters* out);
int rt_get_kernel_times(rt_renderer*, float msOut[6], uint32_t launchesOut[6]);

/* the flattened scene (valid after rt_generate, owned by the renderer) */
int rt_get_flat(rt_renderer*, const rtflat_info** info, const rtflat_model_node** modelNodes, const rtflat_node** nodes,
                const rtflat_tri** tris, double* buildSeconds);
/* the shading tables handed to rthip_upload_materials (NULL / 0 unless the scene option "materials" was set) */
int rt_get_flat_shading(rt_renderer*, const rtflat_shade** shade, const rtflat_material** materials, uint32_t* numMaterials,
                        const float** texels, uint64_t* numTexels);
/* host seconds of the flattener and device milliseconds of the BVH build (0 unless "deviceBvh" was set) */
int rt_get_build_times(rt_renderer*, double* hostSeconds, double* deviceMs);
/* the low-level context, for device-pointer level integration (multi-GPU gather) and probes */
rthip_ctx* rt_context(rt_renderer*);
/* multi-GPU assembly, one renderer per GPU / process (forwards to rthip_comm_init / rthip_gather; the id comes from
 * rthip_comm_unique_id on one rank) */
int rt_comm_init(rt_renderer*, const unsigned char id[RTHIP_COMM_ID_BYTES], int rank, int worldSize);
int rt_gather(rt_renderer*, float* recvDev, uint64_t floatsPerRank, int root);

#ifdef __cplusplus
}
#endif
#endif
