#include "scene/Renderer.h"

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
