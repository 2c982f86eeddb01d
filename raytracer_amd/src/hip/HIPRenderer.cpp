#include "hip/HIPRenderer.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace rt
{
HIPRenderer::HIPRenderer(int device)
{
  // device < 0: offline mode -- the scene can be flattened and inspected, rendering fails with RTHIP_E_NODEVICE
  if (device >= 0)
  {
    error = rthip_create(device, &ctx);
    if (error)
      std::fprintf(stderr, "HIPRenderer: %s\n", rthip_error_string(error));
  }
  scene = new HIPScene(ctx);
}

HIPRenderer::~HIPRenderer()
{
  cancelRender();
  delete scene;
  if (ctx)
    rthip_destroy(ctx);
}

int HIPRenderer::beginFrameRender(Camera cam, RenderParameter params)
{
  if (!ctx || scene->lastError())
  {
    error = ctx ? scene->lastError() : RTHIP_E_NODEVICE;
    std::fprintf(stderr, "HIPRenderer: cannot render: %s\n", rthip_error_string(error));
    return error;
  }
  rthip_camera c;
  std::memcpy(c.position, &cam.position, sizeof(c.position));
  std::memcpy(c.target, &cam.target, sizeof(c.target));
  c.sensorSize[0] = cam.sensorSize.x, c.sensorSize[1] = cam.sensorSize.y;
  c.S_O = cam.S_O, c.f = cam.f, c.A = cam.A;
  rthip_frame_desc frame{};
  frame.width = params.width, frame.height = params.height, frame.numSamples = params.numSamples;
  frame.maxDepth = options.maxDepth;
  frame.aovDepths = options.aovDepths;
  frame.tileSize = options.tileSize;
  frame.numTiles = (uint32_t)options.tiles.size();
  frame.tileIds = options.tiles.empty() ? nullptr : options.tiles.data();
  error = rthip_begin(ctx, &c, &frame);
  if (error == RTHIP_E_BOUNDS)
  {
    // The padding of the BVH boxes was sized for ray origins within originBoundScale scene diagonals of the scene's centre
    // (scene/BvhFlattener.h) and this camera stands further out.  The reference has no such limit: rebuild the acceleration
    // data for a bound that holds this camera (half as far again, so that a camera backing away does not rebuild every
    // frame) and start the frame again.  Anything still queued on the device finishes first (the upload synchronises).
    const rtflat_info& info = scene->flattened().info;
    double d2 = 0, diag2 = 0;
    for (int k = 0; k < 3; ++k)
    {
      const double ctr = 0.5 * ((double)info.sceneLo[k] + (double)info.sceneHi[k]), ext = (double)info.sceneHi[k] - (double)info.sceneLo[k];
      d2 += ((double)c.position[k] - ctr) * ((double)c.position[k] - ctr);
      diag2 += ext * ext;
    }
    if (diag2 > 0 && std::isfinite(d2))
    {
      scene->options.originBoundScale = (float)std::max<double>(scene->options.originBoundScale * 2.0, 1.5 * std::sqrt(d2 / diag2));
      scene->createRayTracingHierarchy();
      error = scene->lastError() ? scene->lastError() : rthip_begin(ctx, &c, &frame);
    }
  }
  if (error)
    std::fprintf(stderr, "HIPRenderer: rthip_begin: %s\n", rthip_error_string(error));
  else
    this->frame = params;
  return error;
}

int HIPRenderer::renderPasses(uint32_t begin, uint32_t count, float* ms)
{
  if (!ctx)
    return error = RTHIP_E_NODEVICE;
  error = rthip_render_passes(ctx, begin, count, ms);
  if (error)
    std::fprintf(stderr, "HIPRenderer: rthip_render_passes: %s\n", rthip_error_string(error));
  return error;
}

int HIPRenderer::enqueuePasses(uint32_t begin, uint32_t count)
{
  if (!ctx)
    return error = RTHIP_E_NODEVICE;
  error = rthip_enqueue_passes(ctx, begin, count);
  if (error)
    std::fprintf(stderr, "HIPRenderer: rthip_enqueue_passes: %s\n", rthip_error_string(error));
  return error;
}

int HIPRenderer::synchronize() { return ctx ? (error = rthip_synchronize(ctx)) : (error = RTHIP_E_NODEVICE); }

// Takes the place of CPURenderer::render (src/cpu/CPURenderer.cpp:138-207).  The device keeps the accumulator.  `running`
// is polled before every call (:146) and a call is bounded to about options.chunkMs of device work, so a cancel returns
// promptly even with main.cpp's 10 000 spp; sampleTimes gets one entry per pass (:204-205): the call's device
// milliseconds divided over the passes it ran.
void HIPRenderer::render(Camera cam, RenderParameter params)
{
  if (beginFrameRender(cam, params))
    return;
  uint32_t samp = 0;
  uint32_t n = options.passesPerCall ? options.passesPerCall : 1u;
  while (samp < params.numSamples)
  {
    if (!running) // src/cpu/CPURenderer.cpp:146
      return;
    n = std::min(n, params.numSamples - samp);
    float ms = 0.0f;
    if (renderPasses(samp, n, &ms))
      return;
    for (uint32_t k = 0; k < n; ++k)
      sampleTimes.push_back(ms / float(n));
    samp += n;
    if (!options.passesPerCall)
    {
      const float perPass = std::max(ms / float(n), 1e-3f);
      n = (uint32_t)std::max(1.0f, std::min(256.0f, options.chunkMs / perPass));
    }
  }
}
} // namespace rt
