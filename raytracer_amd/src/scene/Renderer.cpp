#include "scene/Renderer.h"
#include <numeric>

namespace rt
{
Renderer::Renderer() = default;

Renderer::~Renderer()
{
  // the reference leaks its worker (the process never exits cleanly); a library must not
  cancelRender();
}

// src/scene/Renderer.cpp:10-21: stop and join a render in flight, then start the new one on its own thread
void Renderer::startRender(Camera cam, RenderParameter params)
{
  cancelRender();
  sampleTimes.clear();
  running = true;
  worker = std::thread([this, cam, params] {
    render(cam, params);
    running = false;
  });
}

float Renderer::getLastSampleTime() const { return sampleTimes.empty() ? 0.0f : sampleTimes.back(); }

float Renderer::getAverageSampleTime() const
{
  // (the reference divides by size() unconditionally and yields NaN before the first pass; an empty history reads 0 here)
  return sampleTimes.empty() ? 0.0f : std::accumulate(sampleTimes.begin(), sampleTimes.end(), 0.0f) / (float)sampleTimes.size();
}

void Renderer::waitRender()
{
  if (worker.joinable())
    worker.join();
}

void Renderer::cancelRender()
{
  running = false;
  waitRender();
}
} // namespace rt
