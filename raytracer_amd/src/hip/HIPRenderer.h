// HIPRenderer.h -- the MI355X backend behind the reference's Renderer interface, a sibling of CPURenderer
// (src/cpu/CPURenderer.h:9-39): it forwards the scene mutators to its HIPScene and implements render() by
// driving the HIP kernels through the C ABI of include/rthip.h.  No HIP header is included on the host side.
#pragma once
#include "hip/HIPScene.h"
#include "scene/Renderer.h"

namespace rt
{
// knobs the reference hard-codes or does not have
struct HIPRenderOptions
{
  uint32_t maxDepth = 12;       // src/cpu/CPUScene.cpp:15
  uint32_t aovDepths = 0;       // record hit-id / shadow AOVs for parity checks
  uint32_t tileSize = 64;
  std::vector<uint32_t> tiles;  // empty = whole image; otherwise the tiles this renderer owns (multi-GPU sharding)
  // sample passes handed to the device per call; `running` is polled and sampleTimes appended between calls
  // (src/cpu/CPURenderer.cpp:146,204-205).  0 = adaptive: the first call runs one pass, later calls as many as fit in
  // about chunkMs of device time, so cancelRender()/startRender() return within that bound whatever the sample budget
  uint32_t passesPerCall = 0;
  float chunkMs = 50.0f;
};

class HIPRenderer : public Renderer
{
public:
  explicit HIPRenderer(int device = 0);
  ~HIPRenderer() override;

  void addPointLight(PointLight point) override { scene->addPointLight(point); }
  void addDirectionalLight(DirectionalLight dir) override { scene->addDirectionalLight(dir); }
  void addModel(PModel model, mat4 transform) override { scene->addModel(std::move(model), transform); }
  void addModels(std::vector<PModel> models, mat4 transform) override { scene->addModels(std::move(models), transform); }
  void generate() override { scene->generate(); }

  // the reference's backends draw a window here; this one is headless
  void beginFrame() override {}
  void update() override {}

  HIPRenderOptions options;
  HIPScene* getScene() { return scene; }
  rthip_ctx* context() { return ctx; }
  int lastError() const { return error; }
  // row-major width*height*3, valid after waitRender()
  int readAccumulator(float* rgb) { return ctx ? rthip_read_accumulator(ctx, rgb) : RTHIP_E_NODEVICE; }
  int readImage(float* rgb) { return ctx ? rthip_read_image(ctx, rgb) : RTHIP_E_NODEVICE; }

  // synchronous stepping for hosts that run the sample loop themselves (bench, parity tests): beginFrameRender
  // does what render() does before its loop, renderPasses runs passes [begin, begin+count) and returns device ms
  int beginFrameRender(Camera cam, RenderParameter params);
  int renderPasses(uint32_t begin, uint32_t count, float* ms);
  // the same without waiting: passes (and whole frames, beginFrameRender included) queue up behind each other on the
  // context's stream; synchronize() waits for everything queued so far
  int enqueuePasses(uint32_t begin, uint32_t count);
  int synchronize();
  uint32_t frameWidth() const { return frame.width; }
  uint32_t frameHeight() const { return frame.height; }

  // multi-GPU assembly (one renderer per GPU / process): joins an RCCL communicator and gathers the tile-major accumulators
  // of all ranks on `root` (include/rthip.h: rthip_comm_init, rthip_gather)
  int commInit(const unsigned char* id, int rank, int worldSize) { return ctx ? (error = rthip_comm_init(ctx, id, rank, worldSize)) : RTHIP_E_NODEVICE; }
  int gather(float* recvDev, uint64_t floatsPerRank, int root) { return ctx ? (error = rthip_gather(ctx, recvDev, floatsPerRank, root)) : RTHIP_E_NODEVICE; }

protected:
  void render(Camera cam, RenderParameter params) override;

private:
  rthip_ctx* ctx = nullptr;
  HIPScene* scene = nullptr;
  int error = 0;
  RenderParameter frame{0, 0, 0}; // of the last beginFrameRender
};
} // namespace rt
