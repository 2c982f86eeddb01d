// Procedural.h -- seeded synthetic scenes for the benchmark configurations.
// They stand in for the reference's assimp loader (src/util/ModelLoader.cpp:8-43), which cannot be built
// here: every generator returns a Model with the same fields a loaded mesh would have.
#pragma once
#include "util/Model.h"

namespace rt
{
// Unit cube, corners at +-1, 6 quads as 12 triangles wound so that cross(e0,e1) points outward, vertices
// duplicated per face (24) the way an FBX import produces them.  Stands in for res/models/cube.fbx.
PModel makeCube();

// `numTriangles` independent triangles: first vertex uniform in [-1,1]^3, the two edge vectors uniform in a ball
// of radius `edgeRadius`; zero-area draws are repeated.  PCG32 stream seeded with `seed`.
PModel makeTriangleSoup(uint32_t numTriangles, uint64_t seed, float edgeRadius = 0.02f);

// Height field over [-1,1]^2 (x,z) with `quadsPerSide`^2 quads (2 triangles each), heights from 4 octaves of
// seeded value noise, amplitude `amplitude`.
PModel makeHeightField(uint32_t quadsPerSide, uint64_t seed, float amplitude = 0.15f);

struct Pcg32
{
  uint64_t state, inc;
  explicit Pcg32(uint64_t seed, uint64_t stream = 0x5851f42d4c957f2dull);
  uint32_t next();
  float nextFloat(); // [0,1), 24 bits
};
} // namespace rt
