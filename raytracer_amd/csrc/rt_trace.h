// rt_trace.h: the traversal kernel (five template modes over one body) -- part of the gfx950 kernel library (see rthip.hip for the pipeline overview and DESIGN.md section 5).
#pragma once
#include "rt_kernel_common.h"

namespace rtk
{
// ------------------------------------------------------------------------------------------------ traversal
struct TraceArgs
{
  DeviceScene scene;
  PathBuffers pb;
  const Lights* lights;      // SHADOW only
  FrameGeom geom;            // PRIMARY only
  const CameraConsts* cam;   // PRIMARY only
  uint32_t sampBegin;        // PRIMARY only
  uint32_t numSlots;         // dense: S*P work items, item == slot
  const uint32_t* list;      // sparse: work item -> slot; nullptr in dense mode
  const uint32_t* listCount; // sparse: number of work items (device memory)
  uint32_t runLen;           // static mode: work items per wave (multiple of 64, <= kMaxRun)
  uint32_t chunkLen;         // dynamic mode: work items per fetch (multiple of 64); 0 = derive it from the list length on the device
  uint32_t refillAt;         // idle lanes that trigger a refill from the work list (see the refill rule in the traversal loop)
  uint32_t leafTrigger;      // lanes waiting on a leaf that trigger a leaf step once their number has stopped growing ...
  uint32_t leafTriggerHi;    // ... or at once when this many wait
  uint32_t stealSpans;       // spans a wave may pull chunks from: its own first, then its neighbours'
  uint32_t splitLg;          // SHADOW over a list: log2 of the number of directional lights when every (path, light) pair is an item of its own (see k_trace); 0 = a lane tests a path's lights one after the other
  uint32_t* fetch;           // dynamic mode: kFetchShards counters, kFetchStride words apart, zeroed before the launch
  uint32_t depth;            // live[] value this launch serves (dense mode filter)
  uint32_t* spill;           // [spillCap][gridLanes]
  DeviceCounters* counters;
  float tmin, tmax;
  uint8_t* anyOut;           // ANYPROBE
  uint32_t foldFinal;        // SHADOW: 1 = this is the last depth: fold the lights into the radiance and finish the path here
  uint32_t aovDepths;        // SHADOW with foldFinal: shadow-flag AOV
  uint8_t* aovShadow;
};

template <int MODE, bool COUNT>
// (launch bounds: the per-lane modes need 66-79 VGPRs, so six blocks of four waves still fit a CU -- what 24 KB of LDS per block
// admits; asking for six made the register allocator spill 12-28 bytes per lane to stay under 80, asking for four does not)
__global__ __launch_bounds__(kBlock, (MODE == PRIMARY_GROUP ? 5 : 4)) void k_trace(TraceArgs a)
{
  __shared__ uint32_t sStack[kWavesPerBlock][kStackLds][64];
  __shared__ uint32_t sList[kWavesPerBlock][kChunk];

  const uint32_t lane = laneId();
  const uint32_t wv = threadIdx.x >> 6;
  uint32_t(*stack)[64] = sStack[wv];
  uint32_t* list = sList[wv];
  const uint32_t gridLanes = gridDim.x * kBlock;
  uint32_t* spill = a.spill + (blockIdx.x * kBlock + threadIdx.x);

  const rtflat_model_node* __restrict__ modelNodes = a.scene.modelNodes;
  const float4* __restrict__ nodes4 = reinterpret_cast<const float4*>(a.scene.nodes);
  const float4* __restrict__ tris4 = reinterpret_cast<const float4*>(a.scene.tris);
  const uint32_t numModelNodes = a.scene.numModelNodes;

  // work distribution
  //  static mode (chunkLen == 0): wave `wid` owns items [wid*runLen, (wid+1)*runLen) -- used for coherent primary rays
  //    over tiny scenes, where every ray costs the same and 64-slot groups need no refilling;
  //  dynamic mode: a persistent grid; the items are split into kFetchShards contiguous spans (one per XCD label, so that
  //    an XCD's L2 keeps seeing the same image region) and a wave pulls chunkLen items at a time from its span's counter.
  const bool sparse = a.list != nullptr;
  // SHADOW with splitLg: the scattered shadow rays of the deeper depths are few, and a launch lasts as long as its longest lane.
  // One item per (path, light) -- the light in the top two bits of the wave's list entry -- makes twice (four times) as many
  // items of half (a quarter) the length; each writes its own byte of mask[slot], which k_shade folds back into bits.
  const uint32_t splitLg = MODE == SHADOW ? a.splitLg : 0u;
  const uint32_t total = sparse ? (*a.listCount << splitLg) : a.numSlots;
  constexpr bool dynamic = MODE != PRIMARY_GROUP;
  const uint32_t wid = xcdRemap(blockIdx.x, gridDim.x) * kWavesPerBlock + wv;
  uint32_t runNext = dynamic ? 0u : min(total, wid * a.runLen);
  uint32_t runEnd = dynamic ? 0u : min(total, runNext + a.runLen);
  const uint32_t shard = blockIdx.x % kFetchShards;
  // list launches: the host does not know how many paths are alive, so the chunk size is derived here from the list's
  // length (every resident wave fetches ~8 times when there is that much work, as dynamicGeometry does for dense launches)
  uint32_t chunkLen = a.chunkLen;
  if (dynamic && chunkLen == 0)
  {
    const uint32_t waves = gridDim.x * kWavesPerBlock;
    const uint32_t cl = (((total + waves * 8u - 1) / (waves * 8u)) + 63u) & ~63u;
    chunkLen = max(64u, min(cl, 4u * kChunk));
  }
  const uint32_t chunksPerShard = dynamic ? ((total + kFetchShards - 1) / kFetchShards + chunkLen - 1) / chunkLen : 0u;
  bool exhausted = !dynamic; // dynamic mode: no more chunks anywhere
  uint32_t shardsDone = 0;   // spans found empty so far: the wave's own first, then its neighbours' (work stealing)
  // pulls the next chunk into [runNext, runEnd): from this wave's own span while it lasts, then from the other spans
  // in turn (one attempt per call), so that an XCD whose image region was cheap helps the others out instead of idling
  // through their tail.
  auto fetchChunk = [&]() {
    const uint32_t sh = (shard + shardsDone) % kFetchShards;
    uint32_t c = 0;
    if (lane == 0)
      c = atomicAdd(a.fetch + sh * kFetchStride, 1u);
    c = __builtin_amdgcn_readfirstlane(c);
    const uint32_t base = (sh * chunksPerShard + c) * chunkLen;
    if (c < chunksPerShard && base < total)
    {
      runNext = base;
      runEnd = min(total, base + chunkLen);
    }
    else if (++shardsDone >= a.stealSpans) // this span is empty: the next call tries the next one
      exhausted = true;
  };

  // wave-uniform work-list state
  uint32_t listCount = 0, listNext = 0;
  uint32_t prevStalled = 0; // lanes that were waiting on a leaf at the previous step

  // per-lane ray state
  bool active = false;
  bool groupValid = false; // PRIMARY: this lane carries a camera sample of the current group
  uint32_t slot = 0;
  vec3 o{0, 0, 0}, d{0, 0, 1}, invD{0, 0, 0}, oi{0, 0, 0};
  float best = FLT_MAX, cull = FLT_MAX;
  uint32_t bestRec = RTFLAT_INVALID;
  uint32_t cur = kNone;          // inner node index, a second leaf waiting for the leaf step, kDone or kNone
  uint32_t mk = 0;               // model-tree cursor (pre-order index)
  uint32_t sp = 0, spLo = 0;     // stack entries [spLo, sp) are in LDS, [0, spLo) spilled
  uint32_t postponed = 0xfffffffeu; // a leaf reference parked while the lane keeps walking inner nodes (kNone = none)
  bool needPop = false;          // the lane needs a new `cur` from its stack / the model tree
  uint32_t lightIdx = 0;         // SHADOW: which light this lane is testing (directional lights first, then -- SHADOW_PT -- point lights)
  uint32_t lightEnd = 0;         // SHADOW: one past the last light this lane tests for its path
  uint32_t occluded = 0;         // SHADOW: result bits
  float prevNx = 0.0f;           // SHADOW: rayO.w, the face normal x of the hit the shadow rays leave from
  vec3 hitPos{0, 0, 0}, hitN{0, 0, 0}; // SHADOW_PT: that hit's position and face normal (a point light's ray starts off the surface)
  float tmn = a.tmin, tmx = a.tmax;    // SHADOW_PT: per-lane t range (the launch's for directional lights, [0,1] for point lights)
  const uint32_t numShadowLights = isShadow(MODE) ? a.lights->numDir + (MODE == SHADOW_PT ? a.lights->numPoint : 0u) : 0u;
  uint32_t nRays = 0, nNodes = 0, nTris = 0, nNodeSteps = 0, nLeafSteps = 0;

  // SHADOW: every light of this lane's path has been tested.  Normally the occlusion bits go to mask[] and the next
  // shade launch folds the lights; at the last depth the fold happens right here (same function, same bits), which
  // saves that launch and its re-read of the path state.
  // points the lane's ray at light l from the hit it is testing
  auto aimAtLight = [&](uint32_t l) {
    if (MODE == SHADOW_PT && l >= a.lights->numDir)
    {
      const rthip_point_light& L = a.lights->point[l - a.lights->numDir];
      d = vec3{L.position[0], L.position[1], L.position[2]} - hitPos; // Compute.metal:240,243: not normalised, the light is at t = 1
      o = hitPos + hitN * 1e-3f;                                      // :242
      tmn = 0.0f, tmx = 1.0f;                                         // :244 (min_distance defaults to 0)
      cull = rt::fma(tmx, 1.0f + a.scene.cullRel, a.scene.cullAbs);   // boxes beyond the light cannot hold an accepted triangle
    }
    else
    {
      const rthip_dir_light& L = a.lights->dir[l];
      d = vec3{-L.direction[0], -L.direction[1], -L.direction[2]}; // Ray(position, -d.direction), CPUScene.cpp:34
      if (MODE == SHADOW_PT)
      {
        o = hitPos;
        tmn = a.tmin, tmx = a.tmax;
        cull = FLT_MAX;
      }
    }
    invD = rtd::padInvDir(d);
    oi = o * invD;
  };
  auto finishShadow = [&]() {
    if (a.foldFinal)
    {
      const float4 s0 = a.pb.s0[slot];
      const vec3 from = MODE == SHADOW_PT ? hitPos : o;
      vec3 albedo = referenceAlbedo();
      if (a.scene.materialsOn)
      {
        const float4 al = a.pb.alb[slot];
        albedo = vec3{al.x, al.y, al.z};
      }
      const vec3 radiance = foldLights(a.lights, occluded, from, vec3{prevNx, s0.w, a.pb.nz[slot]}, vec3{s0.x, s0.y, s0.z}, albedo, a.scene.pointShadows);
      if (a.depth - 1 < a.aovDepths)
      {
        uint32_t sidx, i;
        sampleOf(a.geom, slot, sidx, i);
        a.aovShadow[((size_t)sidx * a.aovDepths + (a.depth - 1)) * a.geom.numLocalPixels + i] = (uint8_t)occluded;
      }
      a.pb.s0[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
      a.pb.live[slot] = kDead;
    }
    else if (splitLg)
      reinterpret_cast<uint8_t*>(a.pb.mask)[(size_t)slot * 4 + (lightEnd - 1u)] = (uint8_t)((occluded >> (lightEnd - 1u)) & 1u);
    else
      a.pb.mask[slot] = occluded;
  };

  for (;;)
  {
    if (MODE == PRIMARY_GROUP)
    {
      // ---------------- coherent primary rays over a tiny scene: take the run 64 slots (one 8x8 pixel block) at a time,
      // generate the rays in registers at full occupancy, traverse until the whole group is done, store the 64 hit
      // records together
      if (__ballot(active) == 0)
      {
        // a primary miss contributes exactly +0: its slot is marked finished here and costs the shade kernel nothing
        if (groupValid)
        {
          const bool hitSomething = bestRec != RTFLAT_INVALID;
          if (hitSomething)
            a.pb.hit[slot] = make_uint2(__float_as_uint(best), bestRec);
          a.pb.live[slot] = hitSomething ? (uint8_t)0 : kDeadZero;
        }
        groupValid = false;
        if (runNext >= runEnd)
          break;
        slot = runNext + lane;
        runNext += 64;
        uint32_t sidx, i, w, h;
        sampleOf(a.geom, slot, sidx, i);
        const bool inImage = pixelOf(a.geom, i, w, h);
        if (slot < runEnd && !inImage)
          a.pb.live[slot] = kDeadZero; // padding pixel of a tile that overhangs the image
        if (slot < runEnd && inImage)
        {
          const uint32_t samp = a.sampBegin + sidx;
          rtd::primaryRay(*a.cam, a.geom.width, a.geom.height, w, h, samp, rtd::rand01(w, h, samp), o, d);
          invD = rtd::padInvDir(d);
          oi = o * invD;
          best = FLT_MAX;
          cull = FLT_MAX;
          bestRec = RTFLAT_INVALID;
          cur = kNone;
          postponed = kNone;
          needPop = true;
          mk = 0;
          sp = spLo = 0;
          active = true;
          groupValid = true;
          nRays++;
        }
        continue;
      }
    }
    else if (MODE == PRIMARY)
    {
      // ---------------- primary rays over a real scene: neighbouring pixels still walk very different numbers of nodes,
      // so idle lanes are refilled with the next consecutive slots of the chunk once half the wave is idle
      const uint64_t idleMask = __ballot(!active);
      const uint32_t numIdle = __popcll(idleMask);
      if (numIdle >= a.refillAt)
      {
        if (runNext >= runEnd && !exhausted)
          fetchChunk();
        if (runNext < runEnd)
        {
          const uint32_t r = rankIn(idleMask);
          const uint32_t avail = runEnd - runNext;
          if (!active && r < avail)
          {
            slot = runNext + r;
            uint32_t sidx, i, w, h;
            sampleOf(a.geom, slot, sidx, i);
            if (!pixelOf(a.geom, i, w, h))
              a.pb.live[slot] = kDeadZero; // padding pixel of a tile that overhangs the image
            else
            {
              const uint32_t samp = a.sampBegin + sidx;
              rtd::primaryRay(*a.cam, a.geom.width, a.geom.height, w, h, samp, rtd::rand01(w, h, samp), o, d);
              invD = rtd::padInvDir(d);
              oi = o * invD;
              best = FLT_MAX;
              cull = FLT_MAX;
              bestRec = RTFLAT_INVALID;
              cur = kNone;
              postponed = kNone;
              needPop = true;
              mk = 0;
              sp = spLo = 0;
              active = true;
              nRays++;
            }
          }
          runNext += min(numIdle, avail);
        }
      }
      if (__ballot(active) == 0)
      {
        if (exhausted && runNext >= runEnd)
          break;
        continue;
      }
    }
    else
    {
      // ---------------- work list: scan + compact the next chunk of the run when the list has run dry
      const uint64_t idleMask = __ballot(!active);
      const uint32_t numIdle = __popcll(idleMask);
      const bool refilling = numIdle >= a.refillAt;
      if (listNext >= listCount && refilling && runNext >= runEnd && !exhausted)
        fetchChunk();
      if (listNext >= listCount && runNext < runEnd && refilling)
      {
        const uint32_t chunkEnd = min(runEnd, runNext + kChunk);
        uint32_t n = 0;
        for (uint32_t base = runNext; base < chunkEnd; base += 64)
        {
          const uint32_t item = base + lane;
          bool isLive = item < chunkEnd;
          uint32_t s = item;
          if (sparse)
          {
            if (isLive)
              s = splitLg ? (a.list[item >> splitLg] | ((item & ((1u << splitLg) - 1u)) << 30)) : a.list[item];
          }
          else
            isLive = isLive && (a.pb.live[item] == (uint8_t)a.depth);
          const uint64_t m = __ballot(isLive);
          if (isLive)
            list[n + rankIn(m)] = s;
          n += __popcll(m);
        }
        runNext = chunkEnd;
        listCount = n;
        listNext = 0;
        // the list is written and read by this wave only; LDS operations of one wave complete in order
      }
      // ---------------- hand list entries to idle lanes (ballot + mbcnt compaction)
      const uint32_t avail = listCount - listNext;
      if (avail > 0 && refilling)
      {
        const uint32_t r = rankIn(idleMask);
        const bool take = !active && r < avail;
        if (take)
        {
          slot = list[listNext + r];
          uint32_t firstLight = 0;
          if (splitLg)
          {
            firstLight = slot >> 30;
            slot &= 0x3fffffffu;
          }
          const float4 ro = a.pb.rayO[slot];
          o = vec3{ro.x, ro.y, ro.z};
          best = FLT_MAX;
          cull = FLT_MAX;
          if (isShadow(MODE))
          {
            lightIdx = firstLight;
            lightEnd = splitLg ? firstLight + 1u : numShadowLights;
            occluded = 0;
            prevNx = ro.w;
            if (MODE == SHADOW_PT)
            {
              hitPos = o;
              hitN = vec3{ro.w, a.pb.s0[slot].w, a.pb.nz[slot]};
            }
            aimAtLight(lightIdx);
          }
          else
          {
            const float4 rd = a.pb.rayD[slot];
            d = vec3{rd.x, rd.y, rd.z};
            invD = rtd::padInvDir(d);
            oi = o * invD;
          }
          bestRec = RTFLAT_INVALID;
          cur = kNone;
          postponed = kNone;
          needPop = true;
          mk = 0;
          sp = spLo = 0;
          active = true;
          nRays++;
        }
        listNext += min(numIdle, avail);
      }
      if (__ballot(active) == 0)
      {
        if (exhausted && runNext >= runEnd && listNext >= listCount)
          break;
        continue;
      }
    }

    const bool refillPossible = (MODE == PRIMARY_GROUP) ? false
                                : (MODE == PRIMARY)     ? (runNext < runEnd || !exhausted)
                                                        : (listNext < listCount || runNext < runEnd || !exhausted);
    // ---------------- traversal: a wave alternates NODE steps (every lane that holds an inner node tests its two child
    // boxes) with LEAF steps (every lane that holds a postponed leaf runs the reference's triangle test on it).  A lane
    // that reaches a leaf parks it in `postponed` and keeps walking inner nodes; the leaf step is taken only once
    // kLeafTrigger lanes cannot go on without it (they met a second leaf, or have nothing else left) or no lane can do a
    // node step.  Compared with a plain while-while loop, where every lane waits at its leaf until the slowest lane has
    // found one, this keeps most lanes busy in both kinds of step.
    for (;;)
    {
      // --- resolve: lanes that need a new `cur` pop their stack, or walk on to the next model leaf when it is empty
      while (active && needPop)
      {
        uint32_t got;
        if (sp != 0)
        {
          if (sp == spLo)
          {
            spLo--;
            stack[spLo % kStackLds][lane] = spill[(size_t)spLo * gridLanes];
          }
          sp--;
          got = stack[sp % kStackLds][lane];
        }
        else
        {
          // stack empty: next model leaf whose box chain passes the reference's test, in DFS order
          // (CPUScene.cpp:123-137 / 107-121: both children always, box test with the caller's tmin/tmax)
          got = kDone;
          while (mk < numModelNodes)
          {
            const rtflat_model_node* mn = modelNodes + mk;
            if (COUNT)
              nNodes++;
            // AABB.h:49 divides by the direction: invD is that quotient unless it was clamped (axis-parallel rays, rare),
            // so the three IEEE divisions are repeated only for a wave that holds such a ray
            vec3 invRef = invD;
            if (fabsf(invD.x) == 1e18f || fabsf(invD.y) == 1e18f || fabsf(invD.z) == 1e18f)
              invRef = rt::rcp3(d);
            if (!rtd::refSlab(mn->lo, mn->hi, o, invRef, MODE == SHADOW_PT ? tmn : a.tmin, MODE == SHADOW_PT ? tmx : a.tmax))
            {
              mk += mn->subtreeSize; // skip the subtree
              continue;
            }
            mk++;
            if (mn->triCount > 0)
            {
              got = mn->blas;
              break;
            }
          }
        }
        if (got != kDone && (int)got < 0 && postponed == kNone)
          postponed = got; // a leaf: park it and keep looking for inner-node work
        else
        {
          cur = got;
          needPop = false;
        }
      }
      // --- rays that have nothing left
      if (active && cur == kDone && postponed == kNone)
      {
        bool done = true;
        if (isShadow(MODE))
        {
          // not occluded for this light; the next light restarts the walk
          lightIdx++;
          if (lightIdx < lightEnd)
          {
            aimAtLight(lightIdx);
            mk = 0;
            needPop = true;
            cur = kNone;
            nRays++;
            done = false;
          }
          else
            finishShadow();
        }
        else if (MODE == CLOSEST)
          a.pb.hit[slot] = make_uint2(__float_as_uint(best), bestRec);
        else if (MODE == PRIMARY)
        {
          // a primary miss contributes exactly +0: its slot is marked finished here and costs the shade kernel nothing
          const bool hitSomething = bestRec != RTFLAT_INVALID;
          if (hitSomething)
            a.pb.hit[slot] = make_uint2(__float_as_uint(best), bestRec);
          a.pb.live[slot] = hitSomething ? (uint8_t)0 : kDeadZero;
        }
        else if (MODE == ANYPROBE)
          a.anyOut[slot] = 0;
        // PRIMARY_GROUP: the group's hit records are stored together once every lane has finished
        if (done)
          active = false;
      }
      const bool atNode = active && !needPop && (int)cur >= 0;
      const bool stalled = active && !needPop && postponed != kNone && !((int)cur >= 0);
      const uint64_t nodeMask = __ballot(atNode || (active && needPop));
      const uint64_t stalledMask = __ballot(stalled);
      const uint32_t numStalled = __popcll(stalledMask);
      // A leaf step costs as much as two node steps whatever the number of lanes that take part.  Coherent rays reach their
      // leaves within a step or two of each other, so it pays to let the waiting lanes pile up (hf10m at 64 spp: -6 % frame
      // time with 24 instead of 4); incoherent rays trickle in one by one and waiting only idles lanes (1M soup: +18 %).
      // So: step at once when many wait, and otherwise as soon as a few wait and the node step brought (almost) nobody new.
      const bool leafStep = numStalled >= a.leafTriggerHi || (numStalled >= a.leafTrigger && numStalled <= prevStalled + 1u) || (nodeMask == 0 && numStalled > 0);
      prevStalled = leafStep ? 0u : numStalled;
      if (leafStep)
      {
        if (COUNT)
          nLeafSteps++;
        // --- LEAF step: the reference's triangle test on every record of every parked leaf
        if (active && postponed != kNone)
        {
          const uint32_t first = rtflat_leaf_first(postponed), count = rtflat_leaf_count(postponed);
          postponed = kNone;
          bool anyAccepted = false;
          for (uint32_t k = 0; k < count; ++k)
          {
            const float4* tp = tris4 + (size_t)(first + k) * 4;
            const float4 a0 = tp[0], a1 = tp[1], a2 = tp[2];
            if (COUNT)
              nTris++;
            float t;
            const int acc = rtd::triangleTest(vec3{a0.x, a0.y, a0.z}, vec3{a0.w, a1.x, a1.y}, vec3{a1.z, a1.w, a2.x}, o, d, MODE == SHADOW_PT ? tmn : a.tmin,
                                              MODE == SHADOW_PT ? tmx : a.tmax, t);
            if (acc == 2)
              atomicAdd(&a.counters->nonFinite, 1ull);
            if (acc != 1)
              continue;
            if (isClosest(MODE))
            {
              bool better = t < best;
              if (!better && t == best)
              {
                // tie: the triangle later in (DFS model order, index in model) wins (CPUScene.cpp:231, :136)
                const uint32_t keyNew = a.scene.tris[first + k].orderKey, keyOld = a.scene.tris[bestRec].orderKey;
                better = keyNew > keyOld;
              }
              if (better)
              {
                best = t;
                bestRec = first + k;
                cull = rt::fma(best, 1.0f + a.scene.cullRel, a.scene.cullAbs);
              }
            }
            else
            {
              anyAccepted = true;
              break;
            }
          }
          if (!isClosest(MODE) && anyAccepted)
          {
            // first accepted triangle ends an any-hit ray (testModel returns true, CPUScene.cpp:182)
            if (isShadow(MODE))
            {
              occluded |= 1u << lightIdx;
              lightIdx++;
              if (lightIdx < lightEnd)
              {
                aimAtLight(lightIdx);
                mk = 0;
                sp = spLo = 0;
                cur = kNone;
                needPop = true;
                nRays++;
              }
              else
              {
                finishShadow();
                active = false;
              }
            }
            else
            {
              a.anyOut[slot] = 1;
              active = false;
            }
          }
          else if (!needPop && cur != kDone && (int)cur < 0)
          {
            // the lane was holding a second leaf: park it now and look for more work
            postponed = cur;
            cur = kNone;
            needPop = true;
          }
        }
        continue;
      }
      if (nodeMask == 0)
        break; // nothing left to traverse in this wave: back to the work list
      // Leave for a refill once enough lanes are idle and there is work to hand out.  How many is enough depends on the rays
      // and is the launch's choice (a.refillAt): coherent ones -- a pixel's samples over a terrain, shadow rays toward one
      // light -- finish within a few steps of each other, and handing new rays to the first finishers only puts the wave's
      // lanes out of step for good (hf10m at 64 spp: -10 % frame time with 56 instead of 16, the bunny -11 %); incoherent
      // ones finish one by one and waiting only idles lanes (1M soup: +20 % with 56, +39 % with whole groups).  The host
      // measures which way a scene leans (rthip.hip: calibrateRefill).
      // (an active lane is at a node, needs a pop or waits on a leaf: the idle ones are the rest -- no third ballot)
      if (refillPossible && 64u - (uint32_t)__popcll(nodeMask | stalledMask) >= a.refillAt)
        break;
      // --- NODE step: one 64-byte node = four quantised child boxes
      if (COUNT)
        nNodeSteps++;
      if (atNode)
      {
        const float4* np = nodes4 + (size_t)cur * 4;
        const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
        if (COUNT)
          nNodes++;
        const rtd::WideNode wn = rtd::wideNodeSetup(w0, w2, w3, invD, oi);
        float en0, ex0, en1, ex1, en2, ex2, en3, ex3;
        rtd::wideSlabPair<0>(wn, en0, ex0, en1, ex1);
        rtd::wideSlabPair<2>(wn, en2, ex2, en3, ex3);
        constexpr bool kClosest = isClosest(MODE) || MODE == SHADOW_PT; // modes whose rays have a far limit worth culling against
        // sort the (at most four) children that were hit by entry distance; unused slots have inverted boxes
        uint32_t k0 = ((en0 <= ex0) && (!kClosest || en0 <= cull)) ? __float_as_uint(en0) : 0xffffffffu;
        uint32_t k1 = ((en1 <= ex1) && (!kClosest || en1 <= cull)) ? __float_as_uint(en1) : 0xffffffffu;
        uint32_t k2 = ((en2 <= ex2) && (!kClosest || en2 <= cull)) ? __float_as_uint(en2) : 0xffffffffu;
        uint32_t k3 = ((en3 <= ex3) && (!kClosest || en3 <= cull)) ? __float_as_uint(en3) : 0xffffffffu;
        uint32_t r0 = __float_as_uint(w1.x), r1 = __float_as_uint(w1.y), r2 = __float_as_uint(w1.z), r3 = __float_as_uint(w1.w);
        rtd::orderPair(k0, r0, k1, r1);
        rtd::orderPair(k2, r2, k3, r3);
        rtd::orderPair(k0, r0, k2, r2);
        rtd::orderPair(k1, r1, k3, r3);
        rtd::orderPair(k1, r1, k2, r2);
        // push the farther ones, farthest first, so that the nearest of them is popped first.  Misses sort to the
        // end, so the valid flags are monotone (v1 >= v2 >= v3) and the three references can be written without
        // branches: r3 at sp, r2 at sp+v3, r1 at sp+v3+v2 -- a reference that is not valid is overwritten by the next
        // one or lands above the new top.  One spill check per step makes room for all three.
        if (sp + 3 - spLo > kStackLds)
        {
          do
          {
            spill[(size_t)spLo * gridLanes] = stack[spLo % kStackLds][lane];
            spLo++;
          } while (sp + 3 - spLo > kStackLds);
        }
        const uint32_t v3 = k3 != 0xffffffffu, v2 = k2 != 0xffffffffu, v1 = k1 != 0xffffffffu;
        stack[sp % kStackLds][lane] = r3;
        stack[(sp + v3) % kStackLds][lane] = r2;
        stack[(sp + v3 + v2) % kStackLds][lane] = r1;
        sp += v3 + v2 + v1;
        if (k0 == 0xffffffffu)
          needPop = true;
        else if ((int)r0 < 0 && postponed == kNone)
        {
          postponed = r0; // a leaf: park it, go on with the stack
          needPop = true;
        }
        else
          cur = r0; // an inner node, or a second leaf (the lane now waits for the leaf step)
        if (needPop)
          cur = kNone;
      }
    }
    // the wave is out of traversal work (or wants a refill): PRIMARY_GROUP and the list modes pick up from here
  }

  // ---------------- tallies: one atomic per wave and counter
  for (int off = 32; off > 0; off >>= 1)
  {
    nRays += __shfl_down(nRays, off);
    if (COUNT)
    {
      nNodes += __shfl_down(nNodes, off);
      nTris += __shfl_down(nTris, off);
    }
  }
  if (lane == 0)
  {
    if (nRays)
    {
      unsigned long long* dst =
          isShadow(MODE) ? &a.counters->shadow : ((isPrimary(MODE) || a.depth == 0) ? &a.counters->extendPrimary : &a.counters->extendSecondary);
      atomicAdd(dst, (unsigned long long)nRays);
    }
    if (COUNT)
    {
      atomicAdd(&a.counters->waveNodeSteps, (unsigned long long)nNodeSteps);
      atomicAdd(&a.counters->waveLeafSteps, (unsigned long long)nLeafSteps);
    }
    if (COUNT && (nNodes || nTris)) // every lane fetches the records its ray visits
      atomicAdd(isClosest(MODE) ? &a.counters->fetchExtend : &a.counters->fetchShadow, (unsigned long long)nNodes + nTris);
    if (COUNT && nNodes)
    {
      atomicAdd(&a.counters->nodesVisited, (unsigned long long)nNodes);
      if (isClosest(MODE))
        atomicAdd(&a.counters->nodesExtend, (unsigned long long)nNodes);
    }
    if (COUNT && nTris)
    {
      atomicAdd(&a.counters->trisTested, (unsigned long long)nTris);
      if (isClosest(MODE))
        atomicAdd(&a.counters->trisExtend, (unsigned long long)nTris);
    }
  }
}
} // namespace rtk
