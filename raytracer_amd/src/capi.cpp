// capi.cpp -- the C ABI of include/raytracer_c.h over the C++ Renderer interface.
#include <cstdio>
#include "raytracer_c.h"

#include "hip/HIPRenderer.h"
#include "util/ModelLoader.h"
#include "util/Procedural.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>

using namespace rt;

struct rt_mesh
{
  PModel model;
};

struct rt_renderer
{
  std::unique_ptr<HIPRenderer> hip;
  bool generated = false;
};

namespace
{
rt_mesh* wrap(PModel m)
{
  if (!m)
    return nullptr;
  auto* h = new rt_mesh();
  h->model = std::move(m);
  return h;
}
PModel cloneModel(const Model& src)
{
  auto m = std::make_unique<Model>();
  m->boundingBox = src.boundingBox;
  m->positions = src.positions;
  m->texCoords = src.texCoords;
  m->normals = src.normals;
  m->indices = src.indices;
  m->material = src.material; // textures are shared, not copied
  return m;
}
} // namespace

extern "C" {

rt_mesh* rt_mesh_cube(void) { return wrap(makeCube()); }
rt_mesh* rt_mesh_soup(uint32_t numTriangles, uint64_t seed, float edgeRadius)
{
  return numTriangles ? wrap(makeTriangleSoup(numTriangles, seed, edgeRadius)) : nullptr;
}
rt_mesh* rt_mesh_heightfield(uint32_t quadsPerSide, uint64_t seed, float amplitude)
{
  return quadsPerSide ? wrap(makeHeightField(quadsPerSide, seed, amplitude)) : nullptr;
}
rt_mesh* rt_mesh_load_obj(const char* path)
{
  if (!path)
    return nullptr;
  std::vector<PModel> models = ModelLoader::loadModel(path);
  if (models.empty())
    return nullptr;
  return wrap(std::move(models[0]));
}
uint32_t rt_mesh_load_all(const char* path, rt_mesh** out, uint32_t capacity)
{
  if (!path)
    return 0;
  std::vector<PModel> models = ModelLoader::loadModel(path);
  const uint32_t n = (uint32_t)models.size();
  for (uint32_t k = 0; out && k < n && k < capacity; ++k)
    out[k] = wrap(std::move(models[k]));
  return n;
}
rt_mesh* rt_mesh_from_arrays(const float* positions, uint32_t numPositions, const float* texCoords, const uint32_t* indices,
                             uint32_t numTriangles)
{
  if (!positions || !indices || !numPositions || !numTriangles)
    return nullptr;
  for (uint32_t k = 0; k < 3 * numTriangles; ++k)
    if (indices[k] >= numPositions)
      return nullptr;
  return wrap(Model::fromArrays(positions, numPositions, texCoords, indices, numTriangles));
}
uint32_t rt_mesh_num_positions(const rt_mesh* m) { return m ? (uint32_t)m->model->positions.size() : 0; }
uint32_t rt_mesh_num_triangles(const rt_mesh* m) { return m ? (uint32_t)m->model->indices.size() : 0; }
void rt_mesh_copy(const rt_mesh* m, float* positions3, uint32_t* indices3)
{
  if (!m)
    return;
  if (positions3)
    std::memcpy(positions3, m->model->positions.data(), m->model->positions.size() * sizeof(vec3));
  if (indices3)
    std::memcpy(indices3, m->model->indices.data(), m->model->indices.size() * sizeof(uvec3));
}
int rt_mesh_set_material(rt_mesh* m, uint32_t kind, const float baseColor[3], const float emissive[3], const float* albedoTexels, uint32_t albedoW,
                         uint32_t albedoH, const float* emissiveTexels, uint32_t emissiveW, uint32_t emissiveH)
{
  if (!m || !baseColor || !emissive || kind > 1u)
    return RTHIP_E_ARG;
  if ((albedoTexels && (!albedoW || !albedoH)) || (emissiveTexels && (!emissiveW || !emissiveH)))
    return RTHIP_E_ARG;
  if ((uint64_t)albedoW * albedoH > (1u << 28) || (uint64_t)emissiveW * emissiveH > (1u << 28))
    return RTHIP_E_LIMIT;
  auto texture = [](const float* rgb, uint32_t w, uint32_t h) -> PTexture {
    if (!rgb)
      return nullptr;
    auto t = std::make_shared<Texture>();
    t->width = (int)w, t->height = (int)h;
    t->textureData.resize((size_t)w * h);
    for (size_t k = 0; k < t->textureData.size(); ++k)
      t->textureData[k] = vec3{rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]};
    return t;
  };
  Material& mt = m->model->material;
  mt.kind = kind == 0u ? MaterialKind::ReferenceStub : MaterialKind::Factors;
  mt.baseColor = vec3{baseColor[0], baseColor[1], baseColor[2]};
  mt.emissive = vec3{emissive[0], emissive[1], emissive[2]};
  mt.albedoTexture = texture(albedoTexels, albedoW, albedoH);
  mt.emissiveTexture = texture(emissiveTexels, emissiveW, emissiveH);
  return 0;
}
int rt_mesh_get_material(const rt_mesh* m, uint32_t* kind, float baseColor[3], float emissive[3])
{
  if (!m)
    return RTHIP_E_ARG;
  const Material& mt = m->model->material;
  if (kind)
    *kind = (uint32_t)mt.kind;
  if (baseColor)
    baseColor[0] = mt.baseColor.x, baseColor[1] = mt.baseColor.y, baseColor[2] = mt.baseColor.z;
  if (emissive)
    emissive[0] = mt.emissive.x, emissive[1] = mt.emissive.y, emissive[2] = mt.emissive.z;
  return 0;
}
void rt_mesh_copy_texcoords(const rt_mesh* m, float* texCoords2)
{
  if (m && texCoords2)
    std::memcpy(texCoords2, m->model->texCoords.data(), m->model->texCoords.size() * sizeof(vec2));
}
void rt_mesh_destroy(rt_mesh* m) { delete m; }

int rt_renderer_create(const char* backend, int device, rt_renderer** out)
{
  if (!backend || !out)
    return RTHIP_E_ARG;
  // "hip": the MI355X backend, fails without a device.  "hip-offline": same host code with no device, for
  // inspecting the flattened scene; every render call fails with RTHIP_E_NODEVICE.  There is no CPU backend in
  // this library (the CPU path lives in the test oracle).
  const bool offline = std::strcmp(backend, "hip-offline") == 0;
  if (!offline && std::strcmp(backend, "hip") != 0)
    return RTHIP_E_ARG;
  auto r = std::make_unique<rt_renderer>();
  r->hip = std::make_unique<HIPRenderer>(offline ? -1 : device);
  if (r->hip->lastError())
    return r->hip->lastError();
  *out = r.release();
  return 0;
}
void rt_renderer_destroy(rt_renderer* r) { delete r; }

int rt_add_directional_light(rt_renderer* r, const float d[3], const float c[3])
{
  if (!r || !d || !c)
    return RTHIP_E_ARG;
  if (r->generated)
    return RTHIP_E_STATE; // lights are uploaded with the scene: one added later would silently never shine
  DirectionalLight L;
  L.direction = vec3{d[0], d[1], d[2]};
  L.color = vec3{c[0], c[1], c[2]};
  r->hip->addDirectionalLight(L);
  return 0;
}
int rt_add_point_light(rt_renderer* r, const float p[3], const float c[3], float attenuation)
{
  if (!r || !p || !c)
    return RTHIP_E_ARG;
  if (r->generated)
    return RTHIP_E_STATE;
  PointLight L;
  L.position = vec3{p[0], p[1], p[2]};
  L.color = vec3{c[0], c[1], c[2]};
  L.attenuation = attenuation;
  r->hip->addPointLight(L);
  return 0;
}
int rt_add_mesh(rt_renderer* r, const rt_mesh* m, const float transform[16])
{
  if (!r || !m || !transform)
    return RTHIP_E_ARG;
  if (r->generated)
    return RTHIP_E_STATE;
  mat4 t;
  std::memcpy(&t, transform, sizeof(t));
  r->hip->addModel(cloneModel(*m->model), t);
  return 0;
}
int rt_generate(rt_renderer* r)
{
  if (!r)
    return RTHIP_E_ARG;
  if (r->generated)
    return RTHIP_E_STATE; // the reference's generate() is one-shot too ("todo: clear everything", Scene.cpp:20)
  r->hip->generate();
  r->generated = true;
  return r->hip->getScene()->lastError();
}

int rt_set_option(rt_renderer* r, const char* name, double value)
{
  if (!r || !name)
    return RTHIP_E_ARG;
  const std::string n(name);
  HIPRenderOptions& o = r->hip->options;
  FlattenOptions& fo = r->hip->getScene()->options;
  const uint32_t u = value < 0 ? 0u : (uint32_t)value;
  if (n == "maxDepth")
    o.maxDepth = u;
  else if (n == "aovDepths")
    o.aovDepths = u;
  else if (n == "tileSize")
    o.tileSize = u;
  else if (n == "passesPerCall")
    o.passesPerCall = u;
  else if (n == "chunkMs")
    o.chunkMs = (float)std::max(0.1, value);
  else if (n == "referenceEdgeIndexing")
    fo.referenceEdgeIndexing = value != 0;
  else if (n == "maxLeafTriangles")
    fo.maxLeafTriangles = u;
  else if (n == "sahBins")
    fo.sahBins = u;
  else if (n == "traversalCost")
    fo.traversalCost = (float)std::max(0.0, value);
  else if (n == "buildThreads")
    fo.threads = u;
  else if (n == "deviceBvh")
  {
    fo.deviceBuild = value != 0;
    fo.deviceBuildQuality = value == 2 ? 1 : 0; // 1: Morton LBVH (default), 2: PLOC over the Morton order
  }
  else if (n == "materials")
    r->hip->getScene()->shading.materials = value != 0;
  else if (n == "pointShadows")
    r->hip->getScene()->shading.pointShadows = value != 0;
  else if (n == "originBoundScale")
    fo.originBoundScale = (float)value;
  else if (n == "countWork")
    return r->hip->context() ? rthip_set_work_counting(r->hip->context(), value != 0) : RTHIP_E_NODEVICE;
  else if (n == "fusedPrimary")
    return r->hip->context() ? rthip_set_fused_primary(r->hip->context(), value != 0) : RTHIP_E_NODEVICE;
  else if (n == "pipelining")
    return r->hip->context() ? rthip_set_pipelining(r->hip->context(), value != 0) : RTHIP_E_NODEVICE;
  else if (n == "refillMode")
    return r->hip->context() ? rthip_set_refill_mode(r->hip->context(), value < 0 ? -1 : (int)std::min(3.0, value)) : RTHIP_E_NODEVICE;
  else if (n == "overlapShadow")
    return r->hip->context() ? rthip_set_overlap_shadow(r->hip->context(), value != 0) : RTHIP_E_NODEVICE;
  else if (n == "kernelTiming")
    return r->hip->context() ? rthip_set_kernel_timing(r->hip->context(), value != 0) : RTHIP_E_NODEVICE;
  else
    return RTHIP_E_ARG;
  return 0;
}
int rt_set_tiles(rt_renderer* r, const uint32_t* tileIds, uint32_t numTiles)
{
  if (!r || (numTiles && !tileIds))
    return RTHIP_E_ARG;
  r->hip->options.tiles.assign(tileIds, tileIds + numTiles);
  return 0;
}

static Camera toCamera(const rthip_camera* c)
{
  Camera cam;
  cam.position = vec3{c->position[0], c->position[1], c->position[2]};
  cam.target = vec3{c->target[0], c->target[1], c->target[2]};
  cam.sensorSize = vec2{c->sensorSize[0], c->sensorSize[1]};
  cam.S_O = c->S_O, cam.f = c->f, cam.A = c->A;
  return cam;
}
int rt_start_render(rt_renderer* r, const rthip_camera* c, uint32_t width, uint32_t height, uint32_t numSamples)
{
  if (!r || !c)
    return RTHIP_E_ARG;
  if (!r->generated)
    return RTHIP_E_STATE;
  r->hip->startRender(toCamera(c), RenderParameter{width, height, numSamples});
  return 0;
}
int rt_begin_frame(rt_renderer* r, const rthip_camera* c, uint32_t width, uint32_t height, uint32_t numSamples)
{
  if (!r || !c)
    return RTHIP_E_ARG;
  if (!r->generated)
    return RTHIP_E_STATE;
  r->hip->cancelRender();
  return r->hip->beginFrameRender(toCamera(c), RenderParameter{width, height, numSamples});
}
int rt_render_passes(rt_renderer* r, uint32_t begin, uint32_t count, float* msOut)
{
  if (!r)
    return RTHIP_E_ARG;
  return r->hip->renderPasses(begin, count, msOut);
}
int rt_enqueue_passes(rt_renderer* r, uint32_t begin, uint32_t count) { return r ? r->hip->enqueuePasses(begin, count) : RTHIP_E_ARG; }
int rt_synchronize(rt_renderer* r) { return r ? r->hip->synchronize() : RTHIP_E_ARG; }
int rt_wait_render(rt_renderer* r)
{
  if (!r)
    return RTHIP_E_ARG;
  r->hip->waitRender();
  return r->hip->lastError();
}
int rt_cancel_render(rt_renderer* r)
{
  if (!r)
    return RTHIP_E_ARG;
  r->hip->cancelRender();
  return 0;
}
uint32_t rt_get_sample_times(rt_renderer* r, float* msOut, uint32_t cap)
{
  if (!r)
    return 0;
  const std::vector<float>& t = r->hip->getSampleTimes();
  const uint32_t n = (uint32_t)t.size();
  if (msOut)
    std::memcpy(msOut, t.data(), sizeof(float) * std::min(n, cap));
  return n;
}
int rt_read_accumulator(rt_renderer* r, float* rgb) { return r ? r->hip->readAccumulator(rgb) : RTHIP_E_ARG; }
int rt_read_image(rt_renderer* r, float* rgb) { return r ? r->hip->readImage(rgb) : RTHIP_E_ARG; }
int rt_write_ppm(const char* path, const float* rgb, uint32_t width, uint32_t height)
{
  if (!path || !rgb || !width || !height)
    return RTHIP_E_ARG;
  std::FILE* f = std::fopen(path, "wb");
  if (!f)
    return RTHIP_E_ARG;
  std::fprintf(f, "P6\n%u %u\n255\n", width, height);
  std::vector<unsigned char> row((size_t)width * 3);
  for (uint32_t y = 0; y < height; ++y)
  {
    for (size_t k = 0; k < row.size(); ++k)
    {
      const float v = rgb[(size_t)y * row.size() + k];
      row[k] = (unsigned char)(v >= 1.0f ? 255 : v > 0.0f ? (int)(v * 255.0f + 0.5f) : 0); // NaN -> 0
    }
    if (std::fwrite(row.data(), 1, row.size(), f) != row.size())
    {
      std::fclose(f);
      return RTHIP_E_ARG;
    }
  }
  return std::fclose(f) == 0 ? 0 : RTHIP_E_ARG;
}
int rt_save_image(rt_renderer* r, const char* path, uint32_t width, uint32_t height)
{
  if (!r || !path || !width || !height)
    return RTHIP_E_ARG;
  if (width != r->hip->frameWidth() || height != r->hip->frameHeight())
    return RTHIP_E_ARG; // the read-back below fills the CURRENT frame's width * height pixels
  std::vector<float> rgb((size_t)width * height * 3, 0.0f);
  const int rc = r->hip->readImage(rgb.data());
  return rc ? rc : rt_write_ppm(path, rgb.data(), width, height);
}
int rt_read_aov(rt_renderer* r, uint32_t samp, uint32_t depth, uint32_t* triIdOut, uint8_t* shadowOut)
{
  if (!r)
    return RTHIP_E_ARG;
  return r->hip->context() ? rthip_read_aov(r->hip->context(), samp, depth, triIdOut, shadowOut) : RTHIP_E_NODEVICE;
}
int rt_get_counters(rt_renderer* r, rthip_counters* out)
{
  if (!r)
    return RTHIP_E_ARG;
  return r->hip->context() ? rthip_get_counters(r->hip->context(), out) : RTHIP_E_NODEVICE;
}
int rt_get_kernel_times(rt_renderer* r, float msOut[6], uint32_t launchesOut[6])
{
  if (!r)
    return RTHIP_E_ARG;
  return r->hip->context() ? rthip_get_kernel_times(r->hip->context(), msOut, launchesOut) : RTHIP_E_NODEVICE;
}
int rt_get_flat(rt_renderer* r, const rtflat_info** info, const rtflat_model_node** modelNodes, const rtflat_node** nodes,
                const rtflat_tri** tris, double* buildSeconds)
{
  if (!r || !r->generated)
    return RTHIP_E_STATE;
  const FlatScene& f = r->hip->getScene()->flattened();
  if (info)
    *info = &f.info;
  if (modelNodes)
    *modelNodes = f.modelNodes.data();
  if (nodes)
    *nodes = f.nodes.data();
  if (tris)
    *tris = f.tris.data();
  if (buildSeconds)
    *buildSeconds = f.buildSeconds;
  return 0;
}
int rt_get_flat_shading(rt_renderer* r, const rtflat_shade** shade, const rtflat_material** materials, uint32_t* numMaterials, const float** texels,
                        uint64_t* numTexels)
{
  if (!r || !r->generated)
    return RTHIP_E_STATE;
  const HIPScene::FlatShading& fs = r->hip->getScene()->flattenedShading();
  if (shade)
    *shade = fs.shade.empty() ? nullptr : fs.shade.data();
  if (materials)
    *materials = fs.materials.empty() ? nullptr : fs.materials.data();
  if (numMaterials)
    *numMaterials = (uint32_t)fs.materials.size();
  if (texels)
    *texels = fs.texels.empty() ? nullptr : fs.texels.data();
  if (numTexels)
    *numTexels = fs.texels.size() / 3;
  return 0;
}
rthip_ctx* rt_context(rt_renderer* r) { return r ? r->hip->context() : nullptr; }
int rt_get_build_times(rt_renderer* r, double* hostSeconds, double* deviceMs)
{
  if (!r || !r->generated)
    return RTHIP_E_STATE;
  const FlatScene& f = r->hip->getScene()->flattened();
  if (hostSeconds)
    *hostSeconds = f.buildSeconds;
  if (deviceMs)
    *deviceMs = f.deviceBuildMs;
  return 0;
}
int rt_comm_init(rt_renderer* r, const unsigned char id[RTHIP_COMM_ID_BYTES], int rank, int worldSize)
{
  return (r && id) ? r->hip->commInit(id, rank, worldSize) : RTHIP_E_ARG;
}
int rt_gather(rt_renderer* r, float* recvDev, uint64_t floatsPerRank, int root) { return r ? r->hip->gather(recvDev, floatsPerRank, root) : RTHIP_E_ARG; }
}
