// Camera.h -- thin-lens camera parameters handed to Renderer::startRender.
// Same fields, order and defaults as the reference's Camera (src/util/Camera.h:4-12).
#pragma once
#include "util/Math.h"

namespace rt
{
struct Camera
{
  vec3 position{0, 0, 0};
  vec3 target{0, 0, 0};
  vec2 sensorSize{0.036f, 0.024f}; // 36 x 24 mm
  float S_O = 20.0f;               // object distance
  float f = 0.7f;                  // focal length
  float A = 0.35f;                 // aperture
};
} // namespace rt
