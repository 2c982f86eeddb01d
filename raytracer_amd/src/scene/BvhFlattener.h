// BvhFlattener.h -- writes the scene's acceleration data into the flat 64-byte records of include/rtflat.h.
//
//  1. The reference's model-level tree (src/cpu/CPUScene.cpp:67-105) is rebuilt with the same pairing loop and
//     the same tie behaviour and stored, topology unchanged, as rtflat_model_node[] in pre-order.
//  2. Below every model leaf a triangle-level BVH2 is built (binned SAH).  The reference has nothing here: it
//     scans all triangles of the model (CPUScene.cpp:143,191).  Child boxes are padded outward so that the
//     added culling can only ever skip triangles the reference's own test would reject.
//  3. Triangles become self-contained rtflat_tri records in BVH leaf order.
#pragma once
#include "rtflat.h"
#include "scene/Scene.h"
#include <vector>

namespace rt
{
struct FlattenOptions
{
  // true : e0/e1 of triangle i of a model are read at edgesPool[indicesOffset + 2i (+1)], exactly where the
  //        reference's intersector reads them (CPUScene.cpp:157-158,205-206) -- only right for the first model;
  // false: read at edgesPool[2*indicesOffset + 2i (+1)], where Scene::generate stored them (Scene.cpp:39-40).
  bool referenceEdgeIndexing = true;
  uint32_t maxLeafTriangles = 4; // <= RTFLAT_MAX_LEAF_TRIS
  uint32_t sahBins = 16;
  // SAH cost of visiting an inner node relative to one triangle test: a run of up to maxLeafTriangles triangles stays a
  // leaf when n * area <= traversalCost * area + (cost of the best split)
  float traversalCost = 1.0f;
  float originBoundScale = 4.0f; // ray origins are promised to stay within this many scene diagonals of the centre
  float padUlps = 16.0f;         // box padding in units of 2^-24 * (largest coordinate magnitude a ray can see)
  float cullRel = 1.0f / 4096.0f;
  uint32_t threads = 0;          // 0 = hardware concurrency, capped at 16
  // true: stop before step 2 -- the triangle-level BVH is built on the device (rthip_upload_scene_build); the records of
  // step 3 are then written in the order of the scene's pools and the node array holds a single childless node
  bool deviceBuild = false;
  int deviceBuildQuality = 0;  // rthip_set_build_quality: 0 = Morton LBVH, 1 = PLOC (option deviceBvh = 1 / 2)
};

struct FlatScene
{
  std::vector<rtflat_model_node> modelNodes;
  std::vector<rtflat_node> nodes;
  std::vector<rtflat_tri> tris;
  rtflat_info info{};
  double buildSeconds = 0.0;   // host time of flattenScene
  double deviceBuildMs = 0.0;  // device time of the BVH build when it ran there (0 otherwise)
};

struct ScenePoolsView
{
  const std::vector<ModelReference>* refs;
  const std::vector<AABB>* modelBoxes; // Model::boundingBox after transform, one per model
  const std::vector<vec3>* positionPool;
  const std::vector<uvec3>* indicesPool;
  const std::vector<vec3>* edgesPool;
  const std::vector<vec3>* faceNormalsPool;
};

FlatScene flattenScene(const ScenePoolsView& pools, const FlattenOptions& options);
} // namespace rt
