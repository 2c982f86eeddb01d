// rt_packet.h: packet traversal -- a wave walks the tree ONCE for its 64 rays -- part of the gfx950 kernel library (see rthip.hip for the pipeline overview and DESIGN.md section 5).
#pragma once
#include "rt_trace.h"

namespace rtk
{
// Wave-uniform loads: one scalar load brings a whole 64-byte record (node, triangle, model node) into 16 SGPRs -- one request
// per wave into the scalar cache instead of 64 lanes asking the vector memory path for the same line.  Written as inline
// assembly because the compiler only selects scalar loads where it can prove that no store of the kernel may alias, which
// it cannot here; the scene arrays are written by no kernel.  (Loads only: nothing is ever written through the scalar cache.)
typedef int record16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ record16 loadRecord(const void* p)
{
  record16 r;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  return r;
}
typedef int record4 __attribute__((ext_vector_type(4)));
// a 64-byte record and a 16-byte one, issued together, one wait
__device__ __forceinline__ record16 loadRecordAnd4(const void* p, const void* q, record4& four)
{
  record16 r;
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r), "=&s"(four) : "s"(p), "s"(q) : "memory");
  return r;
}
__device__ __forceinline__ float4 recordQuarter(const record16& r, int q)
{
  return make_float4(__int_as_float(r[4 * q]), __int_as_float(r[4 * q + 1]), __int_as_float(r[4 * q + 2]), __int_as_float(r[4 * q + 3]));
}

// A wave-wide stack held in one VGPR per word: entry i lives in lane i (v_writelane / v_readlane), so a push or a pop is one
// instruction per word, costs no LDS and -- unlike the per-lane stacks of k_trace -- is shared by the 64 rays.
// One stack entry = three words (reference, lane mask low / high) at the same lane index.  (v_writelane with two different SGPR
// operands would break the constant-bus limit, so the lane select goes through M0 -- set once for the three writes and put back.)
__device__ __forceinline__ void laneStackPush(uint32_t& regRef, uint32_t& regLo, uint32_t& regHi, uint32_t ref, uint64_t mask, uint32_t index)
{
  uint32_t keepM0;
  asm volatile("s_mov_b32 %3, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0\n\ts_mov_b32 m0, %3"
               : "+v"(regRef), "+v"(regLo), "+v"(regHi), "=&s"(keepM0)
               : "s"(__builtin_amdgcn_readfirstlane(ref)), "s"(__builtin_amdgcn_readfirstlane((uint32_t)mask)),
                 "s"(__builtin_amdgcn_readfirstlane((uint32_t)(mask >> 32))), "s"(__builtin_amdgcn_readfirstlane(index)));
}
__device__ __forceinline__ uint32_t laneStackRead(uint32_t reg, uint32_t index) { return (uint32_t)__builtin_amdgcn_readlane((int)reg, (int)index); }
// Lane masks without vector instructions.  The compiler turns `__ballot(a <= b)` into compare + select + compare and a per-lane
// `(mask >> lane) & 1` into two ANDs and a 64-bit compare -- 11 of a node visit's ~75 vector instructions, in a kernel whose
// vector ALU port is taken 0.8-0.9 of the time.  A VOP3 compare writes the lane mask straight into an SGPR pair (inactive lanes
// read as 0, NaN compares false, as `<=` does), and inverse_ballot hands a wave-uniform mask back as a per-lane condition.
__device__ __forceinline__ uint64_t laneMaskLe(float a, float b)
{
  uint64_t m;
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}
// the seven 16-byte pieces of a FatNode a wave needs: near and far planes per axis (offsets chosen by the direction signs) and the
// child references; issued together, one wait
struct FatPieces
{
  record4 nx, fx, ny, fy, nz, fz, ch;
};
__device__ __forceinline__ FatPieces loadFat(const void* p, uint32_t offNX, uint32_t offFX, uint32_t offNY, uint32_t offFY, uint32_t offNZ, uint32_t offFZ)
{
  FatPieces f;
  asm volatile("s_load_dwordx4 %0, %7, %8\n\t"
               "s_load_dwordx4 %1, %7, %9\n\t"
               "s_load_dwordx4 %2, %7, %10\n\t"
               "s_load_dwordx4 %3, %7, %11\n\t"
               "s_load_dwordx4 %4, %7, %12\n\t"
               "s_load_dwordx4 %5, %7, %13\n\t"
               "s_load_dwordx4 %6, %7, 0x60\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(f.nx), "=&s"(f.fx), "=&s"(f.ny), "=&s"(f.fy), "=&s"(f.nz), "=&s"(f.fz), "=&s"(f.ch)
               : "s"(p), "s"(offNX), "s"(offFX), "s"(offNY), "s"(offFY), "s"(offNZ), "s"(offFZ)
               : "memory");
  return f;
}
template <int K>
__device__ __forceinline__ void fatSlab(const FatPieces& f, vec3 invD, vec3 noi, float& entry, float& exit)
{
  const float nx = rt::fma(__int_as_float(f.nx[K]), invD.x, noi.x), ny = rt::fma(__int_as_float(f.ny[K]), invD.y, noi.y), nz = rt::fma(__int_as_float(f.nz[K]), invD.z, noi.z);
  const float fx = rt::fma(__int_as_float(f.fx[K]), invD.x, noi.x), fy = rt::fma(__int_as_float(f.fy[K]), invD.y, noi.y), fz = rt::fma(__int_as_float(f.fz[K]), invD.z, noi.z);
  entry = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
  exit = fminf(fminf(fx, fy), fz);
}

constexpr uint32_t kPacketStack = 64; // entries of the wave stack: the host only picks this kernel when 3 * (BVH depth) fits

// k_trace for COHERENT rays (MODE = PRIMARY: camera rays, generated here; SHADOW: any-hit toward every directional light
// from the stored origins).  The 64 work items of a group -- with 64 samples per pixel, one pixel's samples -- are traced
// together: the wave visits one node at a time, every lane tests ITS ray against the node's four child boxes, a child is
// entered when any lane hits it (the lanes that do are remembered in a 64-bit mask pushed with the reference), and a leaf's
// triangles are tested by the lanes that reached it.  Node and triangle records are fetched once per wave by scalar loads,
// the stack is wave-uniform, and there is no per-lane control flow besides the masks, so a node visit costs about a third
// of k_trace's instructions -- as long as the rays really do go the same way; the host measures that (calibrateRefill).
//
// Every lane still applies exactly the tests k_trace applies to its own ray: the reference's slab test on the model boxes
// (and only the triangles of models whose box chain ITS ray passes), the conservative quantised-box test with its own
// culling distance, the reference's triangle test, the orderKey tie rule.  A lane may meet triangles in a different order
// than in k_trace and skip or visit different subtrees, but closest hits are an arg-min with a total tie order and any-hits
// an OR, so the results are the same bits.
template <int MODE, bool COUNT>
__global__ __launch_bounds__(kBlock, 8) void k_trace_packet(TraceArgs a)
{
  static_assert(MODE == PRIMARY || MODE == SHADOW || MODE == SHADOW_PT, "packet traversal serves camera rays and shadow rays");
  constexpr bool kClosest = MODE == PRIMARY;
  constexpr bool kShadow = MODE == SHADOW || MODE == SHADOW_PT;
  const uint32_t lane = laneId();
  const rtflat_model_node* __restrict__ modelNodes = a.scene.modelNodes;
  const float4* __restrict__ nodes4 = reinterpret_cast<const float4*>(a.scene.nodes);
  const float4* __restrict__ tris4 = reinterpret_cast<const float4*>(a.scene.tris);
  const FatNode* __restrict__ fat = a.scene.fat;
  const uint32_t numModelNodes = a.scene.numModelNodes;

  const bool sparse = a.list != nullptr;
  const uint32_t total = sparse ? *a.listCount : a.numSlots;
  // work distribution as in k_trace's dynamic mode: kFetchShards contiguous spans (one per XCD label), chunks of whole groups
  const uint32_t shard = blockIdx.x % kFetchShards;
  const uint32_t waves = gridDim.x * kWavesPerBlock;
  const uint32_t chunkLen = max(64u, min((((total + waves * 8u - 1) / (waves * 8u)) + 63u) & ~63u, 2048u));
  const uint32_t chunksPerShard = ((total + kFetchShards - 1) / kFetchShards + chunkLen - 1) / chunkLen;
  uint32_t shardsDone = 0;
  uint32_t nRays = 0, nNodes = 0, nTris = 0, nNodeSteps = 0, nLeafSteps = 0, nFetched = 0; // nFetched: records loaded by the wave
  uint32_t stackRef = 0, stackLo = 0, stackHi = 0; // the wave stack (see laneStackPush)

  while (shardsDone < kFetchShards)
  {
    const uint32_t sh = (shard + shardsDone) % kFetchShards;
    uint32_t c = 0;
    if (lane == 0)
      c = atomicAdd(a.fetch + sh * kFetchStride, 1u);
    c = __builtin_amdgcn_readfirstlane(c);
    const uint32_t chunkBegin = (sh * chunksPerShard + c) * chunkLen;
    if (!(c < chunksPerShard && chunkBegin < total))
    {
      shardsDone++; // this span is empty: help with the next one
      continue;
    }
    const uint32_t chunkEnd = min(total, chunkBegin + chunkLen);
    for (uint32_t groupBegin = chunkBegin; groupBegin < chunkEnd; groupBegin += 64)
    {
      // ---------------- the group's rays
      const uint32_t item = groupBegin + lane;
      bool valid = item < chunkEnd;
      uint32_t slot = item;
      if (sparse && valid)
        slot = a.list[item];
      vec3 o{0, 0, 0}, d{0, 0, 1};
      vec3 hitPos{0, 0, 0}, hitN{0, 0, 0}; // SHADOW_PT: the hit the shadow rays leave from (a point light's ray starts off the surface)
      float prevNx = 0.0f;
      if (MODE == PRIMARY)
      {
        if (valid)
        {
          uint32_t sidx, i, w, h;
          sampleOf(a.geom, slot, sidx, i);
          if (!pixelOf(a.geom, i, w, h))
          {
            a.pb.live[slot] = kDeadZero; // padding pixel of a tile that overhangs the image
            valid = false;
          }
          else
          {
            const uint32_t samp = a.sampBegin + sidx;
            rtd::primaryRay(*a.cam, a.geom.width, a.geom.height, w, h, samp, rtd::rand01(w, h, samp), o, d);
          }
        }
      }
      else if (valid)
      {
        const float4 ro = a.pb.rayO[slot];
        o = vec3{ro.x, ro.y, ro.z};
        prevNx = ro.w;
        if (MODE == SHADOW_PT)
        {
          hitPos = o;
          hitN = vec3{ro.w, a.pb.s0[slot].w, a.pb.nz[slot]};
        }
      }
      uint32_t occluded = 0;
      float best = FLT_MAX;
      uint32_t bestRec = RTFLAT_INVALID;
      const uint32_t numPasses = MODE == PRIMARY ? 1u : a.lights->numDir + (MODE == SHADOW_PT ? a.lights->numPoint : 0u);
      for (uint32_t pass = 0; pass < numPasses; ++pass)
      {
        float tmn = a.tmin, tmx = a.tmax;
        const bool pointPass = MODE == SHADOW_PT && pass >= a.lights->numDir; // wave-uniform
        if (pointPass)
        {
          // Compute.metal:238-254, as k_trace<SHADOW_PT> casts it: toward the light, not normalised (the light is at t = 1), from
          // just off the surface, t in [0, 1]
          const rthip_point_light& L = a.lights->point[pass - a.lights->numDir];
          d = vec3{L.position[0], L.position[1], L.position[2]} - hitPos;
          o = hitPos + hitN * 1e-3f;
          tmn = 0.0f, tmx = 1.0f;
        }
        else if (kShadow)
        {
          const rthip_dir_light& L = a.lights->dir[pass];
          d = vec3{-L.direction[0], -L.direction[1], -L.direction[2]}; // Ray(position, -d.direction), CPUScene.cpp:34
          if (MODE == SHADOW_PT)
            o = hitPos;
        }
        if (valid)
          nRays++;
        const vec3 invD = rtd::padInvDir(d);
        const vec3 oi = o * invD;
        // Do the wave's rays agree on the sign of every direction component?  Then a child's near and far planes are the same
        // for all of them and come straight out of the FatNode (see rt_kernel_common.h); otherwise this pass decodes the
        // 64-byte records as before.
        const uint64_t validMask = __ballot(valid);
        const uint64_t sxm = __ballot(valid && invD.x < 0.0f), sym = __ballot(valid && invD.y < 0.0f), szm = __ballot(valid && invD.z < 0.0f);
        const bool fatOk = fat != nullptr && (sxm == 0 || sxm == validMask) && (sym == 0 || sym == validMask) && (szm == 0 || szm == validMask);
        const uint32_t offNX = sxm ? 16u : 0u, offFX = sxm ? 0u : 16u, offNY = sym ? 48u : 32u, offFY = sym ? 32u : 48u, offNZ = szm ? 80u : 64u,
                       offFZ = szm ? 64u : 80u;
        const vec3 noi{-oi.x, -oi.y, -oi.z};
        // AABB.h:49 divides by the direction: invD is that quotient unless it was clamped (axis-parallel rays)
        vec3 invRef = invD;
        if (fabsf(invD.x) == 1e18f || fabsf(invD.y) == 1e18f || fabsf(invD.z) == 1e18f)
          invRef = rt::rcp3(d);
        float cull = pointPass ? rt::fma(tmx, 1.0f + a.scene.cullRel, a.scene.cullAbs) : FLT_MAX; // boxes beyond a point light hold no accepted triangle
        best = FLT_MAX;
        bestRec = RTFLAT_INVALID;
        // (wave-uniform) this pass's per-light triangle operands, if the scene has them
        const float4* operands = (kShadow && !pointPass && a.scene.shadowOperands) ? a.scene.shadowOperands + (size_t)pass * a.scene.numTris : nullptr;
        bool live = valid; // this lane's ray is still looking (an any-hit ray stops at its first accepted triangle)
        uint64_t liveMask = __ballot(live); // the same, wave-wide

        // ---------------- model tree, pre-order, the whole wave at one node; a lane is inside node mk iff mk >= skipUntil
        uint32_t skipUntil = 0;
        for (uint32_t mk = 0; mk < numModelNodes;)
        {
          // rtflat_model_node: lo[3] hi[3] left right blas triCount indicesOffset orderBase modelIndex subtreeSize pad[2]
          const record16 mn = loadRecord(modelNodes + mk);
          if (COUNT)
            nFetched++;
          const uint32_t subtree = (uint32_t)mn[13], triCount = (uint32_t)mn[9];
          bool passed = false;
          if (live && mk >= skipUntil)
          {
            if (COUNT)
              nNodes++;
            const float lo[3] = {__int_as_float(mn[0]), __int_as_float(mn[1]), __int_as_float(mn[2])};
            const float hi[3] = {__int_as_float(mn[3]), __int_as_float(mn[4]), __int_as_float(mn[5])};
            passed = rtd::refSlab(lo, hi, o, invRef, tmn, tmx); // CPUScene.cpp:109 / :125, the caller's tmin / tmax
            if (!passed)
              skipUntil = mk + subtree;
          }
          const uint64_t inside = __ballot(passed);
          if (inside == 0)
          {
            mk += subtree; // no ray of the wave goes in
            continue;
          }
          mk++;
          if (triCount == 0)
            continue;
          // ---------------- the model's triangle BVH
          uint32_t sp = 0;
          uint32_t cur = (uint32_t)mn[8];
          uint64_t curMask = inside;
          for (;;)
          {
            cur = __builtin_amdgcn_readfirstlane(cur);
            const uint64_t mineMask = curMask & liveMask; // the lanes whose ray is at this node and still looking
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(mineMask);
            if (rtflat_is_leaf(cur))
            {
              if (COUNT)
                nLeafSteps++;
              const uint32_t first = rtflat_leaf_first(cur), count = rtflat_leaf_count(cur);
              for (uint32_t k = 0; k < count; ++k)
              {
                // shadow rays toward a directional light: cross(d, e1) and 1 / dot(., e0) of the reference's test are the same for
                // every ray of the scene and come with the record (DeviceScene::shadowOperands): a third of the test's instructions
                record4 pre{0, 0, 0, 0};
                const record16 tr = operands ? loadRecordAnd4(tris4 + (size_t)(first + k) * 4, operands + first + k, pre) : loadRecord(tris4 + (size_t)(first + k) * 4);
                if (COUNT)
                  nFetched++;
                const float4 a0 = recordQuarter(tr, 0), a1 = recordQuarter(tr, 1), a2 = recordQuarter(tr, 2);
                if (mine && live)
                {
                  if (COUNT)
                    nTris++;
                  float t;
                  const int acc = operands ? rtd::triangleTestWith(vec3{a0.x, a0.y, a0.z}, vec3{a0.w, a1.x, a1.y}, vec3{a1.z, a1.w, a2.x}, o, d,
                                                                   vec3{__int_as_float(pre[0]), __int_as_float(pre[1]), __int_as_float(pre[2])}, __int_as_float(pre[3]), tmn, tmx, t)
                                           : rtd::triangleTest(vec3{a0.x, a0.y, a0.z}, vec3{a0.w, a1.x, a1.y}, vec3{a1.z, a1.w, a2.x}, o, d, tmn, tmx, t);
                  if (acc == 2)
                    atomicAdd(&a.counters->nonFinite, 1ull);
                  if (acc == 1)
                  {
                    if (kClosest)
                    {
                      bool better = t < best;
                      if (!better && t == best)
                      {
                        // tie: the triangle later in (DFS model order, index in model) wins (CPUScene.cpp:231, :136)
                        const uint32_t keyNew = (uint32_t)tr[13], keyOld = a.scene.tris[bestRec].orderKey;
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
                      live = false; // first accepted triangle ends an any-hit ray (testModel returns true, CPUScene.cpp:182)
                  }
                }
              }
              if (!kClosest)
              {
                liveMask = __ballot(live);
                if (liveMask == 0)
                  break; // every ray of the wave is occluded
              }
            }
            else
            {
              if (COUNT)
                nNodeSteps++;
              if (COUNT && mine)
                nNodes++;
              // every lane runs the box tests (a wave instruction costs the same for one lane as for 64; idle lanes carry a dummy
              // ray); the masks keep the lanes that are at this node.  Unused slots have inverted boxes.
              float en0, en1, en2, en3, ex0, ex1, ex2, ex3;
              uint32_t r0, r1, r2, r3;
              if (fatOk)
              {
                const FatPieces f = loadFat(fat + cur, offNX, offFX, offNY, offFY, offNZ, offFZ);
                if (COUNT)
                  nFetched += 2; // 128 bytes
                fatSlab<0>(f, invD, noi, en0, ex0);
                fatSlab<1>(f, invD, noi, en1, ex1);
                fatSlab<2>(f, invD, noi, en2, ex2);
                fatSlab<3>(f, invD, noi, en3, ex3);
                r0 = (uint32_t)f.ch[0], r1 = (uint32_t)f.ch[1], r2 = (uint32_t)f.ch[2], r3 = (uint32_t)f.ch[3];
              }
              else
              {
                const record16 nd = loadRecord(nodes4 + (size_t)cur * 4);
                if (COUNT)
                  nFetched++;
                const float4 w0 = recordQuarter(nd, 0), w2 = recordQuarter(nd, 2), w3 = recordQuarter(nd, 3);
                const rtd::WideNode wn = rtd::wideNodeSetup(w0, w2, w3, invD, oi);
                rtd::wideSlabPair<0>(wn, en0, ex0, en1, ex1);
                rtd::wideSlabPair<2>(wn, en2, ex2, en3, ex3);
                r0 = (uint32_t)nd[4], r1 = (uint32_t)nd[5], r2 = (uint32_t)nd[6], r3 = (uint32_t)nd[7];
              }
              uint64_t m0 = laneMaskLe(en0, ex0) & mineMask, m1 = laneMaskLe(en1, ex1) & mineMask, m2 = laneMaskLe(en2, ex2) & mineMask,
                       m3 = laneMaskLe(en3, ex3) & mineMask;
              if (kClosest || pointPass)
                m0 &= laneMaskLe(en0, cull), m1 &= laneMaskLe(en1, cull), m2 &= laneMaskLe(en2, cull), m3 &= laneMaskLe(en3, cull);
              if (!kClosest)
              {
                // Any-hit rays need no order among the children (an OR over the triangles: CPUScene.cpp:107-121 visits both
                // subtrees whatever happens), and on the scalar unit -- which this kernel keeps busier than the vector ALU since
                // the float planes (profiles/r03_hf10m_pmc.json: 45.5 G scalar against 40.9 G vector instructions) -- the
                // ordering is the largest item of a visit: four first-lane distance read-outs and a five-exchange network of
                // branches and moves.  The children are taken in slot order: the lowest hit one next, the others on the stack
                // (A/B of two library builds on one box: hf10m 155.7 -> 149.6 ms, bunny 1.42 -> 1.35 ms; round 2, when the vector
                // ALU was the busier unit, measured no difference).
                uint32_t next = RTFLAT_INVALID;
                uint64_t nextMask = 0;
                auto take = [&](uint32_t r, uint64_t m) {
                  if (m)
                  {
                    if (next != RTFLAT_INVALID)
                    {
                      laneStackPush(stackRef, stackLo, stackHi, next, nextMask, sp);
                      sp++;
                    }
                    next = r, nextMask = m;
                  }
                };
                take(r3, m3), take(r2, m2), take(r1, m1), take(r0, m0);
                if (next != RTFLAT_INVALID)
                {
                  cur = next;
                  curMask = nextMask;
                  continue;
                }
              }
              else
              {
              // children in the order of their entry distance along the first ray that hits each (any ray would do: the order
              // only decides how soon the culling distances shrink)
              uint32_t k0 = m0 ? (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(en0), (int)__builtin_ctzll(m0)) : 0xffffffffu;
              uint32_t k1 = m1 ? (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(en1), (int)__builtin_ctzll(m1)) : 0xffffffffu;
              uint32_t k2 = m2 ? (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(en2), (int)__builtin_ctzll(m2)) : 0xffffffffu;
              uint32_t k3 = m3 ? (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(en3), (int)__builtin_ctzll(m3)) : 0xffffffffu;
              auto order = [](uint32_t& ka, uint32_t& ra, uint64_t& ma, uint32_t& kb, uint32_t& rb, uint64_t& mb) {
                if (kb < ka)
                {
                  const uint32_t tk = ka, tr = ra;
                  const uint64_t tm = ma;
                  ka = kb, ra = rb, ma = mb;
                  kb = tk, rb = tr, mb = tm;
                }
              };
              order(k0, r0, m0, k1, r1, m1);
              order(k2, r2, m2, k3, r3, m3);
              order(k0, r0, m0, k2, r2, m2);
              order(k1, r1, m1, k3, r3, m3);
              order(k1, r1, m1, k2, r2, m2);
              // the farther ones go on the stack, farthest first
              if (k3 != 0xffffffffu)
              {
                laneStackPush(stackRef, stackLo, stackHi, r3, m3, sp);
                sp++;
              }
              if (k2 != 0xffffffffu)
              {
                laneStackPush(stackRef, stackLo, stackHi, r2, m2, sp);
                sp++;
              }
              if (k1 != 0xffffffffu)
              {
                laneStackPush(stackRef, stackLo, stackHi, r1, m1, sp);
                sp++;
              }
              if (k0 != 0xffffffffu)
              {
                cur = r0;
                curMask = m0;
                continue;
              }
              }
            }
            if (sp == 0)
              break;
            sp--;
            cur = laneStackRead(stackRef, sp);
            curMask = (uint64_t)laneStackRead(stackLo, sp) | ((uint64_t)laneStackRead(stackHi, sp) << 32);
          }
          if (!kClosest && __ballot(live) == 0)
            break;
        }
        if (kShadow && valid && !live)
          occluded |= 1u << pass;
      }

      // ---------------- results, as k_trace writes them
      if (valid)
      {
        if (MODE == PRIMARY)
        {
          // a primary miss contributes exactly +0: its slot is marked finished here and costs the shade kernel nothing
          const bool hitSomething = bestRec != RTFLAT_INVALID;
          if (hitSomething)
            a.pb.hit[slot] = make_uint2(__float_as_uint(best), bestRec);
          a.pb.live[slot] = hitSomething ? (uint8_t)0 : kDeadZero;
        }
        else if (a.foldFinal)
        {
          // last depth: fold the lights into the radiance and finish the path here (same function as k_shade: same bits)
          const float4 s0 = a.pb.s0[slot];
          vec3 albedo = referenceAlbedo();
          if (a.scene.materialsOn)
          {
            const float4 al = a.pb.alb[slot];
            albedo = vec3{al.x, al.y, al.z};
          }
          const vec3 radiance = foldLights(a.lights, occluded, MODE == SHADOW_PT ? hitPos : o, vec3{prevNx, s0.w, a.pb.nz[slot]}, vec3{s0.x, s0.y, s0.z}, albedo,
                                           a.scene.pointShadows);
          if (a.depth - 1 < a.aovDepths)
          {
            uint32_t sidx, i;
            sampleOf(a.geom, slot, sidx, i);
            a.aovShadow[((size_t)sidx * a.aovDepths + (a.depth - 1)) * a.geom.numLocalPixels + i] = (uint8_t)occluded;
          }
          a.pb.s0[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
          a.pb.live[slot] = kDead;
        }
        else
          a.pb.mask[slot] = occluded;
      }
    }
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
      atomicAdd(kShadow ? &a.counters->shadow : &a.counters->extendPrimary, (unsigned long long)nRays);
    if (COUNT)
    {
      atomicAdd(&a.counters->waveNodeSteps, (unsigned long long)nNodeSteps);
      atomicAdd(&a.counters->waveLeafSteps, (unsigned long long)nLeafSteps);
      atomicAdd(kClosest ? &a.counters->fetchExtend : &a.counters->fetchShadow, (unsigned long long)nFetched);
      if (nNodes)
      {
        atomicAdd(&a.counters->nodesVisited, (unsigned long long)nNodes);
        if (kClosest)
          atomicAdd(&a.counters->nodesExtend, (unsigned long long)nNodes);
      }
      if (nTris)
      {
        atomicAdd(&a.counters->trisTested, (unsigned long long)nTris);
        if (kClosest)
          atomicAdd(&a.counters->trisExtend, (unsigned long long)nTris);
      }
    }
  }
}
} // namespace rtk
