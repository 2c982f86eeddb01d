/*
 * TEST INFRASTRUCTURE -- C API of the CPU oracle (liboracle_cpu.so).
 *
 * The oracle is a CPU restatement of the reference's src/cpu path
 * (Dynamitos/RayTracer: src/cpu/CPURenderer.cpp, src/cpu/CPUScene.cpp,
 * src/scene/{Scene.cpp,AABB.h}, src/util/{Model.cpp,BRDF.cpp}, src/ThreadPool.cpp).
 * The reference itself does not compile in this image (glm / assimp / GLFW /
 * GLEW / ImGui are absent and libstdc++ 11 lacks std::views::zip), see
 * DESIGN.md "Oracle".  PARITY UNPINNED by the reference: it ships no tests,
 * golden vectors or fixtures for this path; the pins are the
 * hand-derived known-answer tests in tests/test_oracle_kat.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (raytracer_amd/) never does.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#include "../include/rtflat.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_camera
{
  float position[3];
  float target[3];
  float sensorSize[2]; /* Camera.h:8 default (0.036, 0.024) */
  float S_O, f, A;     /* Camera.h:9-11 */
} orc_camera;

typedef struct orc_render_params
{
  uint32_t width, height, numSamples; /* RenderParameter, Renderer.h:7-12 */
  uint32_t maxDepth;                  /* reference constant 12 (CPUScene.cpp:15) */
  uint32_t x0, y0, x1, y1;            /* pixel window that is rendered; ray generation always uses width/height */
  uint32_t sampBegin, sampEnd;        /* sample passes to run, [sampBegin, sampEnd) */
  uint32_t numWorkers;                /* 0 = hardware_concurrency()-2 (ThreadPool.h:29) */
  uint32_t aovDepths;                 /* path depths for which AOVs are written (0 = none) */
  uint32_t useFlat;                   /* 0 = variant A: model tree + linear scan (reference-faithful)
                                         1 = variant B: walk the flattened arrays given to orc_set_flat */
  uint32_t pad;
} orc_render_params;

typedef struct orc_counters
{
  uint64_t extendPrimary;   /* closest-hit rays at depth 0 */
  uint64_t extendSecondary; /* closest-hit rays at depth >= 1 (including the one the reference traces and discards at the depth cap) */
  uint64_t shadow;          /* any-hit rays */
  uint64_t nodesVisited;    /* variant B: 64-B node records fetched (model nodes + BVH nodes) */
  uint64_t trisTested;      /* triangle tests executed */
  uint64_t nonFiniteAccepts;/* triangle tests accepted with a NaN t or barycentric: outside the parity domain */
  uint64_t hits[16];        /* closest hits found per depth (depth >= 15 pooled) */
} orc_counters;

/* fixEdgeOffset: 0 reproduces the reference's edgesPool indexing (indicesOffset + 2i, CPUScene.cpp:157-158,205-206);
 *                1 indexes 2*indicesOffset + 2i as Scene.cpp:39-40 lays the pool out. Only differs for 2+ models. */
orc_scene* orc_scene_create(int fixEdgeOffset);
void orc_scene_destroy(orc_scene*);
void orc_add_directional_light(orc_scene*, const float dir[3], const float color[3]);
void orc_add_point_light(orc_scene*, const float pos[3], const float color[3], float attenuation);
/* positions: nPos*3 floats; texcoords: nPos*2 floats or NULL (-> 0,0 as ModelLoader.cpp:29);
 * indices: nTris*3 u32; transform: 16 floats column-major (glm::mat4).  Runs ModelLoader's AABB
 * accumulation (ModelLoader.cpp:18-39) then Scene::addModel -> Model::transform. */
void orc_add_model(orc_scene*, const float* positions, uint32_t nPos, const float* texcoords, const uint32_t* indices,
                   uint32_t nTris, const float transform[16]);
/* Shading beyond the reference's CPU path.  A model's material (src/util/Material.h:7-14): kind 0 =
 * Material::evaluate as it stands (src/util/Material.cpp:4-12: albedo = (u, v, 0)), kind 1 = factor [* texture sample]
 * (src/util/Texture.cpp:3-8) for albedo and emissive; textures are w*h RGB float triples or NULL.  Call after
 * orc_add_model, any time before rendering. */
void orc_set_model_material(orc_scene*, uint32_t modelIndex, uint32_t kind, const float baseColor[3], const float emissive[3],
                            const float* albedoTexels, uint32_t albedoW, uint32_t albedoH, const float* emissiveTexels,
                            uint32_t emissiveW, uint32_t emissiveH);
/* materialsOn: hits take albedo / emissive from their model's material instead of the constants of
 * src/cpu/CPUScene.cpp:243-247; pointShadows: point lights cast shadow rays as in src/metal/Compute.metal:238-254.
 * Both 0 (the default) = the reference's CPU path as it stands. */
void orc_set_shading(orc_scene*, int materialsOn, int pointShadows);
/* variant B's view of the materials: the flattened shading tables (include/rtflat.h), owned by the caller */
void orc_set_flat_shading(orc_scene*, const rtflat_shade* shade, const rtflat_material* materials, const float* texels);
void orc_generate(orc_scene*);

/* pool read-back for cross-checking the product's own Scene::generate / flattener */
uint32_t orc_num_tris(const orc_scene*);
uint32_t orc_num_models(const orc_scene*);
void orc_get_tri_operands(const orc_scene*, uint32_t triId, float p0[3], float e0[3], float e1[3], float n[3]);
/* model tree in pre-order: for node k writes lo[3],hi[3] into boxes[6k..], children indices (or 0xffffffff) into
 * topo[2k..], and the model index (or 0xffffffff for inner nodes) into modelOf[k]. returns node count. */
uint32_t orc_get_model_tree(const orc_scene*, float* boxes, uint32_t* topo, uint32_t* modelOf, uint32_t cap);

/* variant B input: arrays written by the product's flattener; the caller keeps them alive */
void orc_set_flat(orc_scene*, const rtflat_model_node* modelNodes, const rtflat_node* nodes, const rtflat_tri* tris,
                  const rtflat_info* info);

/* accumulator / image: width*height*3 floats, row-major (CPURenderer.cpp:197-198); zeroed here when sampBegin == 0.
 * aovTriId / aovShadow: [numSamples][aovDepths][height][width]; triId 0xffffffff = miss / path dead;
 *   shadow bit l = directional light l occluded, 0xff = no hit.  May be NULL when aovDepths == 0.
 * sampleTimes: numSamples floats (ms, CPURenderer.cpp:204-205) or NULL.  Returns 0. */
int orc_render(orc_scene*, const orc_camera*, const orc_render_params*, float* accumulator, float* image,
               uint32_t* aovTriId, uint8_t* aovShadow, float* sampleTimes, orc_counters* counters);

/* single-function probes for known-answer tests */
void orc_rand01(uint32_t x, uint32_t y, uint32_t z, float out[3]);                 /* CPURenderer.cpp:111-116 */
void orc_primary_ray(const orc_camera*, uint32_t width, uint32_t height, uint32_t w, uint32_t h, uint32_t samp,
                     float origin[3], float dir[3], float rnd[3]);                 /* CPURenderer.cpp:158-192 */
void orc_sincos(float x, float* s, float* c);                                      /* rt_math.h sincos_det */
void orc_bounce_dir(const float normalLight[3], const float rnd[3], float out[3]); /* CPUScene.cpp:54-60 */
int orc_aabb_intersects(const float lo[3], const float hi[3], const float o[3], const float d[3], float tmin, float tmax);
/* closest hit / any hit of one ray through variant A (flat=0) or B (flat=1); returns triId or 0xffffffff, t in *tOut */
uint32_t orc_closest_hit(orc_scene*, int flat, const float o[3], const float d[3], float tmin, float tmax, float* tOut);
int orc_any_hit(orc_scene*, int flat, const float o[3], const float d[3], float tmin, float tmax);

#ifdef __cplusplus
}
#endif
#endif
