// Material.h -- what a model looks like: the data side of the reference's Material / Texture classes
// (src/util/Material.h:7-14, src/util/Texture.h:8-16).  The reference declares them, gives Material::evaluate a stub body
// (src/util/Material.cpp:4-12: albedo = texture coordinates) and never calls it; its intersector hard-codes the BRDF
// inputs instead (src/cpu/CPUScene.cpp:243-247).  This backend wires them into the path when the scene option
// "materials" is set: the evaluation happens on the device (csrc/rt_device.h: materialEvaluate / textureSample), this
// header only carries the inputs there.
#pragma once
#include "util/Math.h"
#include <memory>
#include <vector>

namespace rt
{
struct Texture // src/util/Texture.h:8-16
{
  std::vector<vec3> textureData; // row-major, width * height
  int width = 0;
  int height = 0;
};
using PTexture = std::shared_ptr<Texture>;

enum class MaterialKind : uint32_t
{
  ReferenceStub = 0, // Material::evaluate as it stands in the reference: albedo = (u, v, 0), emissive = 0
  Factors = 1,       // albedo = baseColor [* albedoTexture sample], emissive = emissive [* emissiveTexture sample]
};

struct Material // src/util/Material.h:7-14 + the factors glTF files carry
{
  MaterialKind kind = MaterialKind::Factors;
  vec3 baseColor{0.0f, 1.0f, 0.0f}; // defaults = the constants of CPUScene.cpp:245-246: a model without a material of its own
  vec3 emissive{0.0f, 0.0f, 0.0f};  // looks exactly as it does with materials switched off
  PTexture albedoTexture;
  PTexture emissiveTexture;
};
} // namespace rt
