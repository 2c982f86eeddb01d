// Renderer.h -- the backend interface (reference: src/scene/Renderer.h:14-42).
// A backend forwards the scene mutators to its Scene subclass and implements render(); startRender()
// launches render() on a worker thread exactly as the reference does, and `running` is the cancellation
// flag render() must poll at least once per sample pass (src/cpu/CPURenderer.cpp:146).
#pragma once
#include "scene/Scene.h"
#include "util/Camera.h"
#include <atomic>
#include <numeric>
#include <thread>
#include <vector>

namespace rt
{
struct RenderParameter // src/scene/Renderer.h:7-12
{
  uint32_t width;
  uint32_t height;
  uint32_t numSamples;
};

class Renderer
{
public:
  Renderer();
  virtual ~Renderer();
  virtual void addPointLight(PointLight point) = 0;
  virtual void addDirectionalLight(DirectionalLight dir) = 0;
  virtual void addModel(PModel model, mat4 transform) = 0;
  virtual void addModels(std::vector<PModel> models, mat4 transform) = 0;
  virtual void generate() = 0;
  void startRender(Camera cam, RenderParameter params);
  // headless addition: block until the render thread has finished (the reference's UI loop never needs to)
  void waitRender();
  void cancelRender();
  const std::vector<float>& getSampleTimes() const { return sampleTimes; }
  float getLastSampleTime() const { return sampleTimes.empty() ? 0.0f : sampleTimes.back(); }
  float getAverageSampleTime() const
  {
    return std::accumulate(sampleTimes.begin(), sampleTimes.end(), 0.0f) / sampleTimes.size();
  }
  // main thread
  virtual void beginFrame() = 0;
  virtual void update() = 0;

protected:
  virtual void render(Camera cam, RenderParameter params) = 0;
  std::thread worker;
  std::atomic_bool running{false};
  std::vector<float> sampleTimes;
};
} // namespace rt
