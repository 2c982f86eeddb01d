// Renderer.h -- what a backend has to be.  The method names, argument lists and threading contract are those of the
// reference's Renderer (src/scene/Renderer.h:14-42) so that `HIPRenderer` can sit next to `CPURenderer`/`GPURenderer`
// behind the choice made in src/main.cpp:8; everything else (layout, helpers, lifetime handling) is this project's.
#pragma once
#include "scene/Scene.h"
#include "util/Camera.h"
#include <atomic>
#include <thread>
#include <vector>

namespace rt
{
// Frame size and sample budget of one render (reference: RenderParameter, src/scene/Renderer.h:7-12).
struct RenderParameter
{
  uint32_t width, height;
  uint32_t numSamples;
};

class Renderer
{
public:
  Renderer();
  virtual ~Renderer(); // joins a render in flight: a library must not leak its thread the way the reference's process may

  // ---- rendering, driven from the caller's thread -------------------------------------------------------------------
  // Stops a render in flight, then runs render(cam, params) on a thread of its own (src/scene/Renderer.cpp:10-21).
  void startRender(Camera cam, RenderParameter params);
  void waitRender();   // headless addition: join the render thread (the reference's UI loop never needs to)
  void cancelRender(); // headless addition: clear `running` and join
  virtual void beginFrame() = 0; // main-thread hooks of the reference's UI loop; no-ops for a headless backend
  virtual void update() = 0;

  // ---- scene set-up, forwarded by a backend to its Scene subclass ---------------------------------------------------
  virtual void addDirectionalLight(DirectionalLight dir) = 0;
  virtual void addPointLight(PointLight point) = 0;
  virtual void addModel(PModel model, mat4 transform) = 0;
  virtual void addModels(std::vector<PModel> models, mat4 transform) = 0;
  virtual void generate() = 0;

  // ---- per-sample-pass timings of the current render (CPURenderer.cpp:204-205 fills them; milliseconds) --------------
  const std::vector<float>& getSampleTimes() const { return sampleTimes; }
  float getLastSampleTime() const;
  float getAverageSampleTime() const;

protected:
  // The body of a render: must poll `running` at least once per sample pass (src/cpu/CPURenderer.cpp:146) and append to
  // sampleTimes as it goes.
  virtual void render(Camera cam, RenderParameter params) = 0;

  std::vector<float> sampleTimes;
  std::atomic_bool running{false};
  std::thread worker;
};
} // namespace rt
