#include "util/Procedural.h"

namespace rt
{
Pcg32::Pcg32(uint64_t seed, uint64_t stream) : state(0), inc((stream << 1u) | 1u)
{
  next();
  state += seed;
  next();
}
uint32_t Pcg32::next()
{
  const uint64_t old = state;
  state = old * 6364136223846793005ull + inc;
  const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  const uint32_t rot = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
float Pcg32::nextFloat() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }

static void finishModel(Model& m)
{
  m.texCoords.assign(m.positions.size(), vec2{0.0f, 0.0f});
  m.normals.assign(m.positions.size(), vec3{0.0f, 0.0f, 0.0f});
  for (const vec3& p : m.positions)
    m.boundingBox.adjust(p);
}

PModel makeCube()
{
  // The geometry of res/models/cube.fbx as data: its 8 control points and 6 quads (Objects/Geometry: Vertices,
  // PolygonVertexIndex), expanded the way ModelLoader expands a binary FBX -- one vertex per polygon corner in polygon
  // order, each quad split (0,1,2), (0,2,3) -- so triangle ids match a load of the file itself
  // (tests/test_model_loader_cpu.py compares the two where the file exists).  All windings face outward.
  auto m = std::make_unique<Model>();
  const float points[8][3] = {{1, 1, -1}, {1, -1, -1}, {1, 1, 1}, {1, -1, 1}, {-1, 1, -1}, {-1, -1, -1}, {-1, 1, 1}, {-1, -1, 1}};
  const int quads[6][4] = {{0, 4, 6, 2}, {3, 2, 6, 7}, {7, 6, 4, 5}, {5, 1, 3, 7}, {1, 0, 2, 3}, {5, 4, 0, 1}};
  for (const auto& q : quads)
  {
    const uint32_t base = (uint32_t)m->positions.size();
    for (int c : q)
      m->positions.push_back(vec3{points[c][0], points[c][1], points[c][2]});
    m->indices.push_back(uvec3{base, base + 1, base + 2});
    m->indices.push_back(uvec3{base, base + 2, base + 3});
  }
  finishModel(*m);
  return m;
}

PModel makeTriangleSoup(uint32_t numTriangles, uint64_t seed, float edgeRadius)
{
  auto m = std::make_unique<Model>();
  m->positions.reserve((size_t)numTriangles * 3);
  m->indices.reserve(numTriangles);
  Pcg32 rng(seed);
  auto inBall = [&]() {
    for (;;)
    {
      const vec3 v{2.0f * rng.nextFloat() - 1.0f, 2.0f * rng.nextFloat() - 1.0f, 2.0f * rng.nextFloat() - 1.0f};
      if (dot(v, v) <= 1.0f)
        return v * edgeRadius;
    }
  };
  for (uint32_t t = 0; t < numTriangles; ++t)
  {
    const vec3 c{2.0f * rng.nextFloat() - 1.0f, 2.0f * rng.nextFloat() - 1.0f, 2.0f * rng.nextFloat() - 1.0f};
    vec3 a, b, n;
    do
    {
      a = inBall();
      b = inBall();
      n = cross(a, b);
    } while (dot(n, n) < 1e-24f); // |cross| < 1e-12: redraw
    const uint32_t base = 3 * t;
    m->positions.push_back(c);
    m->positions.push_back(c + a);
    m->positions.push_back(c + b);
    m->indices.push_back(uvec3{base, base + 1, base + 2});
  }
  finishModel(*m);
  return m;
}

namespace
{
inline float latticeValue(uint32_t x, uint32_t y, uint32_t seed)
{
  uint32_t h = x * 0x9E3779B1u ^ (y * 0x85EBCA77u) ^ seed;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
inline float valueNoise(float x, float y, uint32_t seed)
{
  const float fx = std::floor(x), fy = std::floor(y);
  const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
  float tx = x - fx, ty = y - fy;
  tx = tx * tx * (3.0f - 2.0f * tx);
  ty = ty * ty * (3.0f - 2.0f * ty);
  const float v00 = latticeValue(ix, iy, seed), v10 = latticeValue(ix + 1, iy, seed);
  const float v01 = latticeValue(ix, iy + 1, seed), v11 = latticeValue(ix + 1, iy + 1, seed);
  const float a = v00 + (v10 - v00) * tx, b = v01 + (v11 - v01) * tx;
  return a + (b - a) * ty;
}
} // namespace

PModel makeHeightField(uint32_t q, uint64_t seed, float amplitude)
{
  auto m = std::make_unique<Model>();
  const uint32_t side = q + 1;
  m->positions.resize((size_t)side * side);
  m->indices.reserve((size_t)q * q * 2);
  const uint32_t s32 = (uint32_t)(seed ^ (seed >> 32));
  for (uint32_t j = 0; j < side; ++j)
    for (uint32_t i = 0; i < side; ++i)
    {
      const float u = (float)i / (float)q, v = (float)j / (float)q;
      float hgt = 0.0f, amp = 1.0f, freq = 4.0f;
      for (uint32_t o = 0; o < 4; ++o)
      {
        hgt += amp * (valueNoise(u * freq, v * freq, s32 + o * 0x01000193u) - 0.5f);
        amp *= 0.5f;
        freq *= 2.0f;
      }
      m->positions[(size_t)j * side + i] = vec3{2.0f * u - 1.0f, amplitude * hgt, 2.0f * v - 1.0f};
    }
  for (uint32_t j = 0; j < q; ++j)
    for (uint32_t i = 0; i < q; ++i)
    {
      const uint32_t a = j * side + i, b = a + 1, c = a + side, d = c + 1;
      // wound so that the face normals point up (+y)
      m->indices.push_back(uvec3{a, c, b});
      m->indices.push_back(uvec3{b, c, d});
    }
  finishModel(*m);
  return m;
}
} // namespace rt
