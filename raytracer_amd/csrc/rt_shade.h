// rt_shade.h: frame set-up, shading, compaction, accumulation, tonemap / untile and the counter fold -- part of the gfx950 kernel library (see rthip.hip for the pipeline overview and DESIGN.md section 5).
#pragma once
#include "rt_kernel_common.h"

namespace rtk
{
// ------------------------------------------------------------------------------------------------ set-up kernels
// also clears the frame's counter totals (one dispatch less in the frame prologue)
__global__ void k_camera(rthip_camera cam, CameraConsts* out, rthip_counters* totals)
{
  if (totals && blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < sizeof(rthip_counters) / sizeof(uint64_t); i += blockDim.x)
      reinterpret_cast<uint64_t*>(totals)[i] = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *out = rtd::cameraSetup(cam);
}

// stand-alone ray generation (used when primary-ray fusion is switched off, and as the A/B baseline for it)
__global__ __launch_bounds__(kBlock) void k_generate(FrameGeom g, const CameraConsts* camp, PathBuffers pb, uint32_t sampBegin,
                                                     uint32_t numPaths)
{
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= numPaths)
    return;
  uint32_t s, i, w, h;
  sampleOf(g, p, s, i);
  if (!pixelOf(g, i, w, h))
  {
    pb.live[p] = kDeadZero;
    return;
  }
  const CameraConsts k = *camp;
  const uint32_t samp = sampBegin + s;
  const vec3 rnd = rtd::rand01(w, h, samp);
  vec3 o, d;
  rtd::primaryRay(k, g.width, g.height, w, h, samp, rnd, o, d);
  pb.rayO[p] = make_float4(o.x, o.y, o.z, 0.0f);
  pb.rayD[p] = make_float4(d.x, d.y, d.z, 0.0f);
  pb.live[p] = 0;
}

// ------------------------------------------------------------------------------------------------ shading
struct ShadeArgs
{
  FrameGeom g;
  DeviceScene scene;
  PathBuffers pb;
  const Lights* lights;
  const CameraConsts* cam;
  DeviceCounters* counters;
  uint32_t numSlots, depth, maxDepth, sampBegin;
  uint32_t finalFold;      // 1: only fold the pending lights and finish every path (depth == maxDepth)
  uint32_t primaryInRegs;  // 1: depth-0 rays were never stored; recompute them from (pixel, sample)
  uint32_t maskSplit;      // > 0: this depth's shadow launch wrote mask[] as one byte per light, for this many lights (k_trace<SHADOW>, splitLg)
  uint32_t aovDepths;
  uint32_t* aovTri;        // [S][aovDepths][P]
  uint8_t* aovShadow;      // [S][aovDepths][P]
  const uint32_t* list;    // depth >= 1: the slots alive at `depth`, listCount[depth] of them (nullptr at depth 0: every slot)
  uint32_t* listNext;      // the paths that go on are appended here (count in counters->listCount[depth+1]); nullptr: none kept
};

// One launch per depth.  SPARSE = false (depth 0): every slot of the batch is an item; SPARSE = true (depth >= 1): the items
// are the entries of the slot list the previous depth's launch wrote (its length lives in device memory, so the host never
// has to learn how many paths are still alive).  Paths that go on are appended to the next depth's list: a wave shades a
// run of up to kShadeRun consecutive items, keeps the surviving slots in LDS in item order, and publishes them with ONE atomic,
// so the list keeps the spatial order of the slots (neighbouring entries = neighbouring pixels and samples) and the one
// counter word sees numItems / 1024 atomics per launch instead of one per wave iteration.
// MAT: the scene carries materials (rthip_upload_materials): a hit's albedo / emissive come from its model's material at the
// hit's texture coordinates and the albedo travels with the path (pb.alb) to the light fold one depth later.
template <bool SPARSE, bool MAT>
__global__ __launch_bounds__(kBlock, 8) void k_shade(ShadeArgs a)
{
  __shared__ uint32_t sKeep[kWavesPerBlock][kShadeRun];
  __shared__ uint32_t sAlive[kWavesPerBlock], sHits[kWavesPerBlock];
  uint32_t nAlive = 0, nHits = 0;
  const uint8_t depthTag = (uint8_t)a.depth;
  const uint32_t total = SPARSE ? a.counters->listCount[a.depth] : a.numSlots;
  const uint32_t lane = laneId(), wv = threadIdx.x >> 6;
  uint32_t* keep = sKeep[wv];
  const uint32_t numWaves = gridDim.x * kWavesPerBlock;
  // a list too short to give every wave of the grid a full run is cut into shorter runs (64 items or more), so that the launch
  // is spread over all the waves instead of being a few waves' sixteen passes one after the other (hf10m at 4 spp -5 %, soup1m
  // -1.5 %; a fixed run of 512 or 256 costs the long lists of hf10m 1-8 %)
  const uint32_t runLen = min(kShadeRun, max(64u, (((total + numWaves - 1) / numWaves) + 63u) & ~63u));
  const uint32_t numRuns = (total + runLen - 1) / runLen;
  const bool appending = a.finalFold == 0 && a.listNext != nullptr;
  for (uint32_t run = blockIdx.x * kWavesPerBlock + wv; run < numRuns; run += numWaves)
  {
    uint32_t kept = 0; // wave-uniform
    const uint32_t runBegin = run * runLen, runEnd = min(total, runBegin + runLen);
    uint32_t count = runEnd - runBegin;
    if (!SPARSE)
    {
      // depth 0: only the slots whose camera ray hit something are shaded (the fused primary kernels write the depth-0 bytes:
      // 0 = hit, kDeadZero = miss).  They are gathered first, so that the shading below runs on full waves whatever the share
      // of hits (hf10m: 52 %).  The survivors go back into the same buffer: entry k is written only after the group that
      // holds entry k has been read (kept <= entries read so far; LDS operations of one wave complete in order).
      uint32_t numHit = 0;
      for (uint32_t base = runBegin; base < runEnd; base += 64)
      {
        const uint32_t item = base + lane;
        const bool isHit = item < runEnd && a.pb.live[item] == depthTag;
        const uint64_t m = __ballot(isHit);
        if (isHit)
          keep[numHit + rankIn(m)] = item;
        numHit += (uint32_t)__popcll(m);
      }
      count = numHit;
    }
    for (uint32_t base = 0; base < count; base += 64)
    {
    const uint32_t idx = base + lane;
    const bool mine = idx < count;
    uint32_t p = 0;
    if (mine)
      p = SPARSE ? a.list[runBegin + idx] : keep[idx];
    bool survives = false;
    if (mine)
    {
      uint32_t s, i;
      sampleOf(a.g, p, s, i);
      uint32_t w = 0, h = 0;
      const bool regen = a.depth == 0 && a.primaryInRegs;
      const bool folding = a.finalFold != 0;
      vec3 o{0, 0, 0}, d{0, 0, 1}, rnd{0, 0, 0};
      vec3 radiance{0.0f, 0.0f, 0.0f}, material{1.0f, 1.0f, 1.0f}; // Ray.h:7-8
      if (!regen)
      {
        const float4 ro = a.pb.rayO[p];
        o = vec3{ro.x, ro.y, ro.z};
        if (a.depth > 0)
        {
          const float4 s0 = a.pb.s0[p];
          radiance = vec3{s0.x, s0.y, s0.z};
          // lights of the previous hit, whose position is this ray's origin (CPUScene.cpp:31-49)
          const bool shadowed = a.lights->numDir || (a.scene.pointShadows && a.lights->numPoint);
          uint32_t occl = shadowed ? a.pb.mask[p] : 0u;
          if (a.maskSplit) // this depth's shadow launch wrote one byte per light (k_trace<SHADOW>, splitLg)
            occl = ((occl & 1u) | ((occl >> 7) & 2u) | ((occl >> 14) & 4u) | ((occl >> 21) & 8u)) & ((1u << a.maskSplit) - 1u);
          vec3 albedoPrev = referenceAlbedo();
          if (MAT)
          {
            const float4 al = a.pb.alb[p];
            albedoPrev = vec3{al.x, al.y, al.z};
          }
          radiance = foldLights(a.lights, occl, o, vec3{ro.w, s0.w, a.pb.nz[p]}, radiance, albedoPrev, a.scene.pointShadows);
          if (a.depth - 1 < a.aovDepths)
            a.aovShadow[((size_t)s * a.aovDepths + (a.depth - 1)) * a.g.numLocalPixels + i] = (uint8_t)occl;
        }
      }
      bool finished = folding;
      uint32_t rec = RTFLAT_INVALID;
      float t = FLT_MAX;
      vec3 albedo = referenceAlbedo();        // CPUScene.cpp:245
      vec3 emissive{0.0f, 0.0f, 0.0f};        // CPUScene.cpp:246
      float4 a2 = make_float4(0, 0, 0, 0), a3 = a2; // third and fourth quarter of the hit's triangle record
      if (!finished)
      {
        const uint2 hv = a.pb.hit[p];
        t = __uint_as_float(hv.x);
        rec = hv.y;
        if (!(t < FLT_MAX)) // CPUScene.cpp:10
          finished = true;
        else
        {
          nHits++;
          pixelOf(a.g, i, w, h);
          rnd = rtd::rand01(w, h, a.sampBegin + s);
          if (regen) // the primary ray was never stored: same function, same bits (only paths that hit need it)
            rtd::primaryRay(*a.cam, a.g.width, a.g.height, w, h, a.sampBegin + s, rnd, o, d);
          else
          {
            // direction and throughput are read only now: a ray that missed -- 96 % of hf10m's first bounce -- never needs them
            const float4 rd = a.pb.rayD[p];
            d = vec3{rd.x, rd.y, rd.z};
            if (a.depth > 0)
            {
              const float2 s1 = a.pb.s1[p];
              material = vec3{rd.w, s1.x, s1.y};
            }
          }
          const float4* tp = reinterpret_cast<const float4*>(a.scene.tris + rec);
          a2 = tp[2], a3 = tp[3];
          if (MAT)
          {
            // IntersectionInfo::brdf of this hit (CPUScene.cpp:232-247 with the material in place of the constants)
            const float4 a0 = tp[0], a1 = tp[1];
            const rtflat_shade sh = a.scene.shade[__float_as_uint(a3.x)];
            float u, v;
            rtd::hitTexCoords(vec3{a0.x, a0.y, a0.z}, vec3{a0.w, a1.x, a1.y}, vec3{a1.z, a1.w, a2.x}, o, d, sh.uv, u, v);
            rtd::materialEvaluate(a.scene.materials[sh.material], a.scene.texels, u, v, albedo, emissive);
          }
          const float pr = rt::maxRef(rt::maxRef(albedo.x, albedo.y), albedo.z); // :13
          if (a.depth > 5)                                                      // :19 (depth >= maxDepth never reaches here)
          {
            if (rnd.z >= pr)
              finished = true;
            else
              material = material / pr;
          }
        }
      }
      if (finished)
      {
        if (a.depth == 0)
          a.pb.live[p] = kDeadZero; // a primary miss adds exactly +0 to the accumulator: nothing to store
        else
        {
          a.pb.s0[p] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
          a.pb.live[p] = kDead;
        }
      }
      else
      {
        const vec3 n{a2.y, a2.z, a2.w};
        if (a.depth < a.aovDepths)
          a.aovTri[((size_t)s * a.aovDepths + a.depth) * a.g.numLocalPixels + i] = __float_as_uint(a3.x);
        const float payloadEmissive = a.depth == 0 ? 1.0f : 0.0f;                  // Ray.h:10, CPUScene.cpp:61
        radiance = radiance + material * emissive * payloadEmissive;              // :27
        material = material * albedo;                                             // :28
        const vec3 position = o + d * t;                                          // :238
        a.pb.rayO[p] = make_float4(position.x, position.y, position.z, n.x);
        a.pb.s0[p] = make_float4(radiance.x, radiance.y, radiance.z, n.y);
        a.pb.nz[p] = n.z;
        if (MAT)
          a.pb.alb[p] = make_float4(albedo.x, albedo.y, albedo.z, 0.0f);
        if (a.depth + 1 < a.maxDepth) // the last shaded depth traces no bounce ray: neither direction nor throughput is kept
        {
          const vec3 normalLight = rt::dot(n, d) < 0 ? n : -n;                    // :240
          const vec3 nd = rtd::bounceDirection(normalLight, rnd);                 // :54-60
          a.pb.rayD[p] = make_float4(nd.x, nd.y, nd.z, material.x);
          a.pb.s1[p] = make_float2(material.y, material.z);
        }
        a.pb.live[p] = (uint8_t)(a.depth + 1);
        nAlive++;
        survives = true;
      }
    }
    if (appending)
    {
      const uint64_t m = __ballot(survives);
      if (survives)
        keep[kept + rankIn(m)] = p;
      kept += (uint32_t)__popcll(m);
    }
    }
    // publish the run's survivors: one atomic, then the LDS copy goes out in item order
    // (the buffer is written and read by this wave only; LDS operations of one wave complete in order)
    if (appending && kept)
    {
      uint32_t dst = 0;
      if (lane == 0)
        dst = atomicAdd(&a.counters->listCount[a.depth + 1], kept);
      dst = __builtin_amdgcn_readfirstlane(dst);
      for (uint32_t k = lane; k < kept; k += 64)
        a.listNext[dst + k] = keep[k];
    }
  }
  // block tallies: one atomic per block and counter
  for (int off = 32; off > 0; off >>= 1)
  {
    nAlive += __shfl_down(nAlive, off);
    nHits += __shfl_down(nHits, off);
  }
  if ((threadIdx.x & 63u) == 0)
  {
    sAlive[wv] = nAlive;
    sHits[wv] = nHits;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t ta = 0, th = 0;
    for (uint32_t k = 0; k < kWavesPerBlock; ++k)
    {
      ta += sAlive[k];
      th += sHits[k];
    }
    if (ta)
      atomicAdd(&a.counters->alive[a.depth + 1], (unsigned long long)ta);
    if (th)
      atomicAdd(&a.counters->hits[a.depth < 15 ? a.depth : 15], (unsigned long long)th);
  }
}

// accumulator[pix] += radiance / float(N), sample passes in ascending order (CPURenderer.cpp:197), for the pixels of the batch
// `fresh` = first render call of a frame: the accumulator is taken as zero instead of being loaded (saves clearing it beforehand)
__global__ __launch_bounds__(kBlock) void k_accumulate(FrameGeom g, PathBuffers pb, float* acc, uint32_t numBatchSamples, uint32_t fresh)
{
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= g.batchPixels)
    return;
  const size_t i = (size_t)g.batchPixelBase + j;
  float r = 0, gg = 0, b = 0;
  bool loaded = fresh != 0; // 0 + x == x and 0 + 0 == +0: starting from zero is what the cleared accumulator would give
  const float n = float(g.numSamples);
  auto add = [&](float4 v) {
    if (!loaded)
    {
      r = acc[3 * i], gg = acc[3 * i + 1], b = acc[3 * i + 2];
      loaded = true;
    }
    r += v.x / n;
    gg += v.y / n;
    b += v.z / n;
  };
  if (g.batchLg == 0 && (numBatchSamples & 3u) == 0)
  {
    // one pixel per micro-block: this thread's samples are consecutive slots.  Four at a time, so that every 64-byte line
    // of radiance records (and every word of liveness bytes) is fetched once and used whole -- one record per iteration
    // made 64 lanes revisit 64 different lines four times over, which the L1 does not hold on to.
    const size_t first = (size_t)j * numBatchSamples;
    for (uint32_t s = 0; s < numBatchSamples; s += 4)
    {
      const uint32_t lv = *reinterpret_cast<const uint32_t*>(pb.live + first + s);
      const bool d0 = (lv & 0xffu) == kDead, d1 = ((lv >> 8) & 0xffu) == kDead, d2 = ((lv >> 16) & 0xffu) == kDead, d3 = (lv >> 24) == kDead;
      if (!(d0 || d1 || d2 || d3))
        continue;
      const float4* q = pb.s0 + first + s;
      const float4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
      if (d0)
        add(v0);
      if (d1)
        add(v1);
      if (d2)
        add(v2);
      if (d3)
        add(v3);
    }
  }
  else
    for (uint32_t s = 0; s < numBatchSamples; ++s)
    {
      const size_t p = slotOf(g, s, j);
      // kDeadZero: the sample's radiance is exactly +0 (primary miss) -> adding 0/N leaves the sum unchanged
      if (pb.live[p] != kDead)
        continue;
      add(pb.s0[p]);
    }
  if (loaded)
  {
    acc[3 * i] = r;
    acc[3 * i + 1] = gg;
    acc[3 * i + 2] = b;
  }
}

// image = pow(max(accumulator * resolver, 0), 0.45)  (CPURenderer.cpp:196,198); display only
__global__ __launch_bounds__(kBlock) void k_tonemap(const float* acc, float* img, uint32_t numFloats, float resolver)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < numFloats)
    img[i] = powf(rt::maxRef(acc[i] * resolver, 0.0f), 0.45f);
}

// tile-major -> row-major scatter (multi-GPU assembly and read-back)
__global__ __launch_bounds__(kBlock) void k_untile(FrameGeom g, const float* tileMajor, float* rowMajor, uint32_t numPixels)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= numPixels)
    return;
  uint32_t w, h;
  if (!pixelOf(g, i, w, h))
    return;
  const size_t dst = 3 * ((size_t)h * g.width + w);
  rowMajor[dst] = tileMajor[3 * (size_t)i];
  rowMajor[dst + 1] = tileMajor[3 * (size_t)i + 1];
  rowMajor[dst + 2] = tileMajor[3 * (size_t)i + 2];
}

// adds one batch's tallies to the frame totals (device memory), so a frame needs no host round trip between batches
__global__ void k_fold_counters(const DeviceCounters* batch, rthip_counters* totals)
{
  if (threadIdx.x != 0 || blockIdx.x != 0)
    return;
  // atomics: two batches in flight on two streams may fold at the same time
  auto add = [](uint64_t* dst, unsigned long long v) {
    if (v)
      atomicAdd(reinterpret_cast<unsigned long long*>(dst), v);
  };
  add(&totals->extendPrimary, batch->extendPrimary);
  add(&totals->extendSecondary, batch->extendSecondary);
  add(&totals->shadow, batch->shadow);
  add(&totals->nodesVisited, batch->nodesVisited);
  add(&totals->trisTested, batch->trisTested);
  add(&totals->nonFiniteAccepts, batch->nonFinite);
  add(&totals->nodesVisitedExtend, batch->nodesExtend);
  add(&totals->trisTestedExtend, batch->trisExtend);
  add(&totals->waveNodeSteps, batch->waveNodeSteps);
  add(&totals->waveLeafSteps, batch->waveLeafSteps);
  add(&totals->recordsFetchedExtend, batch->fetchExtend);
  add(&totals->recordsFetchedShadow, batch->fetchShadow);
  for (int i = 0; i < 16; ++i)
    add(&totals->hits[i], batch->hits[i]);
}
} // namespace rtk
