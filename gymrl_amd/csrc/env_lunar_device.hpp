// env_lunar_device.hpp — device side of env_lunar.hip (shared with the persistent rollout kernel).
// env_lunar.hip — LunarLander-v3 (discrete, no wind) batched stepper for gfx950.
//
// Replaces env.reset()/env.step() of gym.make("LunarLander-v3") as called from
// ppo_lunarlander.py:200,211,222 and ppo_full_lunarlander.py:466,474,496.  The
// arithmetic behind that call is gymnasium's lunar_lander.py on top of Box2D 2.3
// (both third-party, absent from the reference tree, unpinned: SURVEY.md §8c.2),
// so this file is a purpose-built rigid-body solver that follows their published
// structure: 3 dynamic bodies (hull + 2 legs), 2 revolute joints with motor +
// limits, polygon-vs-terrain-edge manifolds (SAT + clipping, 0.98/0.001
// hysteresis), sequential impulses with warm starting and the 2-point block
// solver, 180 velocity / <= 60 position iterations per 1/50 s step, Box2D's
// sleep rule (the +100 "landed" terminal), gymnasium's reward shaping.
//
// Mapping: FOUR lanes per env (a DPP quad), 16 envs per wavefront, one solver wavefront per
// workgroup (N/16 workgroups: 256 at N = 4096, one per CU).  Every lane of a quad carries
// the three bodies and both joints in VGPRs; the contact work is split by role: lane 0 owns
// the hull's terrain contacts, lanes 1/2 the legs' (lane 3 mirrors lane 0 and never stores)
// — the three bodies' contact solves only touch their own body (the terrain is static), so
// they run side by side in one instruction stream.  In the 180 velocity sweeps the joint solve
// runs ACROSS the quad instead: lane r holds component r (x, y, angular) of each 3-vector,
// cross products and Solve33's sums are DPP-operand arithmetic in the reference's order; steps
// with a contact transpose between the two layouts once per sweep.  Per-lane manifolds and
// solver scratch live in LDS columns ([word][lane]).
// Bound: instruction issue.  One wave on a SIMD issues one instruction per ~5 clocks whatever
// it is (tools/ubench/valu_issue.hip), so a step costs its instruction count; HBM sees the
// 576 B of SoA state per env once per step — or, in the persistent rollout kernels, once per
// launch (world_park keeps the worlds in LDS between steps).
//
// Determinism: IEEE f32 +,-,*,/,sqrt and explicit fmaf only (-ffp-contract=off),
// det_sincosf for rotations, Philox for every random draw — the CPU oracle's
// independent restatement reproduces every bit.
#pragma once
#include "env_common.hpp"

namespace gymrl {
namespace lunar {


// ------------------------------------------------------------------ constants
constexpr float kScale = 30.0f;
constexpr float kW = 600.0f / kScale;            // 20
constexpr float kH = 400.0f / kScale;            // 13.333
constexpr float kHelipadY = kH / 4.0f;
constexpr float kLegDown = 18.0f / kScale;
constexpr float kLegAway = 20.0f / kScale;
constexpr float kDt = 1.0f / 50.0f;
constexpr int kVelIters = 180, kPosIters = 60, kMaxSteps = 1000;
// kLinearSlopSqMax: the largest float whose correctly rounded square root is <= kLinearSlop (sqrtf is monotonic, so
// sqrtf(x) <= kLinearSlop  <=>  x <= kLinearSlopSqMax, bit for bit; tests/test_lunar_constants.py re-derives it)
constexpr float kLinearSlopSqMax = 0x1.a36e30p-16f;
constexpr float kLinearSlop = 0.005f, kAngularSlop = 2.0f / 180.0f * 3.14159265359f;
constexpr float kPolyRadius = 2.0f * kLinearSlop;           // b2_polygonRadius
constexpr float kMaxLinCorr = 0.2f, kMaxAngCorr = 8.0f / 180.0f * 3.14159265359f;
constexpr float kBaumgarte = 0.2f;
constexpr float kMaxTranslation = 2.0f, kMaxRotation = 0.5f * 3.14159265359f;
constexpr float kTimeToSleep = 0.5f, kLinSleepTol = 0.01f, kAngSleepTol = 2.0f / 180.0f * 3.14159265359f;
constexpr float kMotorTorque = 40.0f;

// body indices: 0 = hull, 1 = legs[0] (i = -1, right), 2 = legs[1] (i = +1, left)
// mass data from Box2D's polygon ComputeMass (density 5 hull / 1 legs), precomputed in double
__device__ constexpr float kInvM[3] = {0.20761245674740486f, 14.0625f, 14.0625f};
__device__ constexpr float kInvI[3] = {1.2757043935679302f, 558.3639705882352f, 558.3639705882352f};
constexpr float kHullLcY = 0.10130718954248369f;            // hull local centre of mass (x = 0)
// hull polygon, CCW hull order Box2D's Set() produces, and its outward normals
__device__ constexpr float kHullVx[6] = {17.f / 30, 17.f / 30, 14.f / 30, -14.f / 30, -17.f / 30, -17.f / 30};
__device__ constexpr float kHullVy[6] = {-10.f / 30, 0.f, 17.f / 30, 17.f / 30, 0.f, -10.f / 30};
__device__ constexpr float kHullNx[6] = {1.0f, 0.9847835588179369f, 0.0f, -0.9847835588179369f, -1.0f, 0.0f};
__device__ constexpr float kHullNy[6] = {0.0f, 0.17378533390904766f, 1.0f, 0.17378533390904766f, 0.0f, -1.0f};
__device__ constexpr float kLegVx[4] = {-2.f / 30, 2.f / 30, 2.f / 30, -2.f / 30};
__device__ constexpr float kLegVy[4] = {-8.f / 30, -8.f / 30, 8.f / 30, 8.f / 30};
__device__ constexpr float kLegNx[4] = {0.0f, 1.0f, 0.0f, -1.0f};
__device__ constexpr float kLegNy[4] = {-1.0f, 0.0f, 1.0f, 0.0f};

typedef float v2f __attribute__((ext_vector_type(2)));
struct Body { float cx, cy, a, vx, vy, w; };   // centre of mass, angle, velocities

struct Joint {              // revolute joint hull(A) -> leg(B)
  float ix, iy, iz, im;     // accumulated impulse (x, y, limit), motor impulse
  int state;                // 0 inactive, 1 at lower, 2 at upper
};

struct ContactPt { float lpx, lpy; uint32_t key; float ni, ti; };   // local point, feature key, impulses
struct Manifold {
  int count;                // 0, 1, 2
  int faceB;                // 0: reference face on the terrain edge (A), 1: on the polygon (B)
  float lnx, lny, lpx, lpy; // local normal / plane point (A frame == world for the static terrain)
  ContactPt p[2];
};

// Manifolds (warm-start impulses included) and the solver's per-contact scratch live in
// LDS during a step, one dword column per lane ([word][64 lanes]: bank = lane, conflict
// free); contacts are rare, so this keeps the always-hot state (bodies, joints) in VGPRs.
constexpr int kMfWords = 16;   // count, faceB, lnx, lny, lpx, lpy, 2 x {lpx, lpy, key, ni, ti}
constexpr int kVcWords = 17;   // count, nx, ny, rx0, ry0, rx1, ry1, nm0, nm1, tm0, tm1, K(3), K^-1(3)
constexpr int kLdsWords = 2 * kMfWords + 2 * kVcWords;   // per lane: its own body's two slots
constexpr int kQuad = 4;                                  // lanes per env
constexpr int kEnvsPerBlock = kEnvBlock / kQuad;          // 16

// Section timers of the probe build (make prof): 100 MHz ticks summed per workgroup by its first lane.
#ifdef GYMRL_LUNAR_PROF
#define LUNAR_PROF_MARK(var) const unsigned long long var = wall_clock64()
#define LUNAR_PROF_ADD(lds, slot, t0, t1) do { if ((lds).prof && (threadIdx.x & 63) == 0) (lds).prof[slot] += (t1) - (t0); } while (0)
#else
#define LUNAR_PROF_MARK(var)
#define LUNAR_PROF_ADD(lds, slot, t0, t1)
#endif

struct Lds {
  uint32_t* w;   // base + lane
#ifdef GYMRL_LUNAR_PROF
  unsigned long long* prof;
#endif
  uint32_t* park = nullptr;   // persistent kernels: this lane's column of the parking area ([kParkWords][64 lanes]), see world_park
#ifdef GYMRL_LUNAR_PROF
  unsigned long long* envp = nullptr;   // probe build: this lane's env's counters, [3][envn]: steps with a contact of its own, its own
  int envn = 0;                         // position iterations, steps on which the WAVE ran the contact sweeps
#endif
  __device__ __forceinline__ float& mf(int s, int f) const { return reinterpret_cast<float*>(w)[(s * kMfWords + f) * kEnvBlock]; }
  __device__ __forceinline__ uint32_t& mu(int s, int f) const { return w[(s * kMfWords + f) * kEnvBlock]; }
  __device__ __forceinline__ float& vc(int s, int f) const { return reinterpret_cast<float*>(w)[(2 * kMfWords + s * kVcWords + f) * kEnvBlock]; }
  __device__ __forceinline__ uint32_t& vu(int s, int f) const { return w[(2 * kMfWords + s * kVcWords + f) * kEnvBlock]; }
};

// quad broadcast: every lane of a quad reads lane R's value (v_mov_b32 dpp quad_perm:[R,R,R,R]; bound_ctrl: every source
// lane of a quad permutation is valid, and without it the compiler first zeroes the destination — one more instruction)
template <int R>
__device__ __forceinline__ float quad_bcast(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), R * 0x55, 0xF, 0xF, true));
}
// general quad permutation (dpp quad_perm, CTRL = src of lane 0 | lane 1 << 2 | lane 2 << 4 | lane 3 << 6); the compiler folds
// the move into the consuming VALU instruction's first operand (v_mul_f32_dpp ...)
template <int CTRL>
__device__ __forceinline__ float quad_perm(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
constexpr int kRot1 = 0xC9;     // lanes (0,1,2,3) read lanes (1,2,0,3)
constexpr int kRot2 = 0xD2;     // lanes (0,1,2,3) read lanes (2,0,1,3)
constexpr int kSwap01 = 0xE1;   // lanes (0,1,2,3) read lanes (1,0,2,3)
template <int R>
__device__ __forceinline__ uint32_t quad_bcast_u(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, R * 0x55, 0xF, 0xF, true);
}
enum { MF_COUNT = 0, MF_FACEB = 1, MF_LNX = 2, MF_LNY = 3, MF_LPX = 4, MF_LPY = 5, MF_P0 = 6, MF_P1 = 11,
       P_LPX = 0, P_LPY = 1, P_KEY = 2, P_NI = 3, P_TI = 4 };
enum { VC_COUNT = 0, VC_NX = 1, VC_NY = 2, VC_RX0 = 3, VC_RY0 = 4, VC_RX1 = 5, VC_RY1 = 6, VC_NM0 = 7, VC_NM1 = 8,
       VC_TM0 = 9, VC_TM1 = 10, VC_K11 = 11, VC_K12 = 12, VC_K22 = 13, VC_I11 = 14, VC_I12 = 15, VC_I22 = 16 };

struct World {
  Body b[3];
  float sleep[3];
  Joint j[2];
  int edge0;                // terrain edge index of slot 0 of THIS lane's body (slot 1 = edge0 + 1)
  uint32_t touching;        // bit (10*body + edge): pair was touching after the last Collide
  float ty[11];             // smoothed terrain heights at x = 0, 2, ..., 20
  uint32_t flags;           // bit0 leg0 ground contact, bit1 leg1, bit2 game_over, bit3 has prev_shaping, bit4 asleep
  float prev_shaping;
};

__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fmaxf(lo, fminf(x, hi)); }

// joint geometry per leg L (0: i=-1, 1: i=+1)
__device__ __forceinline__ float leg_sign(int L) { return L == 0 ? -1.0f : 1.0f; }
__device__ __forceinline__ float joint_lower(int L) { return L == 0 ? 0.4f : -0.9f; }
__device__ __forceinline__ float joint_upper(int L) { return L == 0 ? 0.9f : -0.4f; }

// ------------------------------------------------------------------ collision
// b2CollideEdgeAndPolygon for an edge without adjacent vertices, edge body static at
// the identity transform.  NV/V*/N* describe the polygon in its body frame; (px,py,qs,qc)
// is the polygon body's transform; (ccx, ccy) its centroid in world coordinates.
template <int NV>
__device__ __forceinline__ void collide_edge_polygon(Manifold& mf, const float* __restrict__ VX,
                                                     const float* __restrict__ VY,
                                                     const float* __restrict__ NX,
                                                     const float* __restrict__ NY, float px, float py,
                                                     float qs, float qc, float ccx, float ccy,
                                                     float v1x, float v1y, float v2x, float v2y) {
  mf.count = 0;
  float wx[NV], wy[NV], wnx[NV], wny[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    wx[i] = (qc * VX[i] - qs * VY[i]) + px;
    wy[i] = (qs * VX[i] + qc * VY[i]) + py;
    wnx[i] = qc * NX[i] - qs * NY[i];
    wny[i] = qs * NX[i] + qc * NY[i];
  }
  float ex = v2x - v1x, ey = v2y - v1y;
  const float el = sqrtf(ex * ex + ey * ey);
  const float inv = 1.0f / el;
  ex *= inv; ey *= inv;
  const float n1x = ey, n1y = -ex;
  const float offset1 = dot2(n1x, n1y, ccx - v1x, ccy - v1y);
  const bool front = offset1 >= 0.0f;
  const float Nx = front ? n1x : -n1x, Ny = front ? n1y : -n1y;
  const float radius = 2.0f * kPolyRadius;

  // edge axis
  float esep = 3.4e38f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = dot2(Nx, Ny, wx[i] - v1x, wy[i] - v1y);
    if (s < esep) esep = s;
  }
  if (esep > radius) return;
  // polygon axes
  float psep = -3.4e38f;
  int pidx = -1;
  bool separated = false;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float nx = -wnx[i], ny = -wny[i];
    const float s1 = dot2(nx, ny, wx[i] - v1x, wy[i] - v1y);
    const float s2 = dot2(nx, ny, wx[i] - v2x, wy[i] - v2y);
    const float s = fminf(s1, s2);
    if (!separated) {
      if (s > radius) { separated = true; }
      else if (s > psep) { psep = s; pidx = i; }
    }
  }
  if (separated) return;
  const bool use_edge = (pidx < 0) || !(psep > 0.98f * esep + 0.001f);

  // incident edge (ie0, ie1) and reference face
  float i0x, i0y, i1x, i1y;
  uint32_t id0, id1;
  float rv1x, rv1y, rv2x, rv2y, rnx, rny;
  int ri1, ri2;
  if (use_edge) {
    int best = 0;
    float bestv = dot2(Nx, Ny, wnx[0], wny[0]);
#pragma unroll
    for (int i = 1; i < NV; ++i) {
      const float v = dot2(Nx, Ny, wnx[i], wny[i]);
      if (v < bestv) { bestv = v; best = i; }
    }
    const int b2 = best + 1 < NV ? best + 1 : 0;
    i0x = wx[0]; i0y = wy[0]; i1x = wx[0]; i1y = wy[0];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i == best) { i0x = wx[i]; i0y = wy[i]; }
      if (i == b2) { i1x = wx[i]; i1y = wy[i]; }
    }
    // id bytes: indexA | indexB<<8 | typeA<<16 | typeB<<24   (type: 0 vertex, 1 face)
    id0 = 0u | ((uint32_t)best << 8) | (1u << 16) | (0u << 24);
    id1 = 0u | ((uint32_t)b2 << 8) | (1u << 16) | (0u << 24);
    if (front) { ri1 = 0; ri2 = 1; rv1x = v1x; rv1y = v1y; rv2x = v2x; rv2y = v2y; rnx = n1x; rny = n1y; }
    else { ri1 = 1; ri2 = 0; rv1x = v2x; rv1y = v2y; rv2x = v1x; rv2y = v1y; rnx = -n1x; rny = -n1y; }
  } else {
    i0x = v1x; i0y = v1y; i1x = v2x; i1y = v2y;
    id0 = 0u | ((uint32_t)pidx << 8) | (0u << 16) | (1u << 24);
    id1 = 0u | ((uint32_t)pidx << 8) | (0u << 16) | (1u << 24);
    ri1 = pidx; ri2 = pidx + 1 < NV ? pidx + 1 : 0;
    rv1x = wx[0]; rv1y = wy[0]; rv2x = wx[0]; rv2y = wy[0]; rnx = wnx[0]; rny = wny[0];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i == ri1) { rv1x = wx[i]; rv1y = wy[i]; rnx = wnx[i]; rny = wny[i]; }
      if (i == ri2) { rv2x = wx[i]; rv2y = wy[i]; }
    }
  }
  const float sn1x = rny, sn1y = -rnx;
  const float so1 = dot2(sn1x, sn1y, rv1x, rv1y);
  const float so2 = dot2(-sn1x, -sn1y, rv2x, rv2y);

  // b2ClipSegmentToLine x2
  float c0x, c0y, c1x, c1y; uint32_t cid0, cid1; int np = 0;
  {
    const float d0 = dot2(sn1x, sn1y, i0x, i0y) - so1;
    const float d1 = dot2(sn1x, sn1y, i1x, i1y) - so1;
    c0x = c0y = c1x = c1y = 0.0f; cid0 = cid1 = 0u;
    if (d0 <= 0.0f) { c0x = i0x; c0y = i0y; cid0 = id0; np = 1; }
    if (d1 <= 0.0f) { if (np == 0) { c0x = i1x; c0y = i1y; cid0 = id1; } else { c1x = i1x; c1y = i1y; cid1 = id1; } ++np; }
    if (d0 * d1 < 0.0f) {
      const float t = d0 / (d0 - d1);
      const float nxp = i0x + t * (i1x - i0x), nyp = i0y + t * (i1y - i0y);
      const uint32_t nid = (uint32_t)ri1 | (((id0 >> 8) & 0xFFu) << 8) | (0u << 16) | (1u << 24);
      if (np == 0) { c0x = nxp; c0y = nyp; cid0 = nid; } else { c1x = nxp; c1y = nyp; cid1 = nid; }
      ++np;
    }
  }
  if (np < 2) return;
  float e0x, e0y, e1x, e1y; uint32_t eid0, eid1; np = 0;
  {
    const float d0 = dot2(-sn1x, -sn1y, c0x, c0y) - so2;
    const float d1 = dot2(-sn1x, -sn1y, c1x, c1y) - so2;
    e0x = e0y = e1x = e1y = 0.0f; eid0 = eid1 = 0u;
    if (d0 <= 0.0f) { e0x = c0x; e0y = c0y; eid0 = cid0; np = 1; }
    if (d1 <= 0.0f) { if (np == 0) { e0x = c1x; e0y = c1y; eid0 = cid1; } else { e1x = c1x; e1y = c1y; eid1 = cid1; } ++np; }
    if (d0 * d1 < 0.0f) {
      const float t = d0 / (d0 - d1);
      const float nxp = c0x + t * (c1x - c0x), nyp = c0y + t * (c1y - c0y);
      const uint32_t nid = (uint32_t)ri2 | (((cid0 >> 8) & 0xFFu) << 8) | (0u << 16) | (1u << 24);
      if (np == 0) { e0x = nxp; e0y = nyp; eid0 = nid; } else { e1x = nxp; e1y = nyp; eid1 = nid; }
      ++np;
    }
  }
  if (np < 2) return;

  mf.faceB = use_edge ? 0 : 1;
  if (use_edge) { mf.lnx = rnx; mf.lny = rny; mf.lpx = rv1x; mf.lpy = rv1y; }
  else {
    mf.lnx = NX[0]; mf.lny = NY[0]; mf.lpx = VX[0]; mf.lpy = VY[0];
#pragma unroll
    for (int i = 0; i < NV; ++i) if (i == ri1) { mf.lnx = NX[i]; mf.lny = NY[i]; mf.lpx = VX[i]; mf.lpy = VY[i]; }
  }
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float qx = k == 0 ? e0x : e1x, qy = k == 0 ? e0y : e1y;
    const uint32_t qid = k == 0 ? eid0 : eid1;
    const float sep = dot2(rnx, rny, qx - rv1x, qy - rv1y);
    if (sep <= radius) {
      float lx, ly; uint32_t key;
      if (use_edge) {   // store in polygon-local coordinates: MulT(xf, q)
        const float dx = qx - px, dy = qy - py;
        lx = qc * dx + qs * dy; ly = -qs * dx + qc * dy;
        key = qid;
      } else {          // store in edge (world) frame, id with A/B swapped
        lx = qx; ly = qy;
        key = ((qid >> 8) & 0xFFu) | ((qid & 0xFFu) << 8) | (((qid >> 24) & 0xFFu) << 16) | (((qid >> 16) & 0xFFu) << 24);
      }
      if (cnt == 0) { mf.p[0].lpx = lx; mf.p[0].lpy = ly; mf.p[0].key = key; }
      else { mf.p[1].lpx = lx; mf.p[1].lpy = ly; mf.p[1].key = key; }
      ++cnt;
    }
  }
  mf.count = cnt;
}
__device__ __forceinline__ void body_xf(const Body& b, float lcy, float& px, float& py, float& qs, float& qc) {
  det_sincosf(b.a, &qs, &qc);
  // p = c - R * localCenter, localCenter = (0, lcy)
  px = b.cx - (qc * 0.0f - qs * lcy);
  py = b.cy - (qs * 0.0f + qc * lcy);
}

// role -> owned body: lanes 0/3 the hull, 1 legs[0], 2 legs[1]
__device__ __forceinline__ float sel3(int mb, float a, float b, float c) { return mb == 0 ? a : (mb == 1 ? b : c); }
__device__ __forceinline__ Body select_body(const Body (&B)[3], int mb) {
  Body r;   // field-wise selects: keeps the body array in VGPRs (no address-taken struct copy)
  r.cx = sel3(mb, B[0].cx, B[1].cx, B[2].cx); r.cy = sel3(mb, B[0].cy, B[1].cy, B[2].cy);
  r.a = sel3(mb, B[0].a, B[1].a, B[2].a);
  r.vx = sel3(mb, B[0].vx, B[1].vx, B[2].vx); r.vy = sel3(mb, B[0].vy, B[1].vy, B[2].vy);
  r.w = sel3(mb, B[0].w, B[1].w, B[2].w);
  return r;
}

// Every lane of the quad adopts lane r's copy of body r (velocities or positions).
__device__ __forceinline__ void quad_share_velocity(Body (&B)[3], const Body& mine) {
  B[0].vx = quad_bcast<0>(mine.vx); B[0].vy = quad_bcast<0>(mine.vy); B[0].w = quad_bcast<0>(mine.w);
  B[1].vx = quad_bcast<1>(mine.vx); B[1].vy = quad_bcast<1>(mine.vy); B[1].w = quad_bcast<1>(mine.w);
  B[2].vx = quad_bcast<2>(mine.vx); B[2].vy = quad_bcast<2>(mine.vy); B[2].w = quad_bcast<2>(mine.w);
}
__device__ __forceinline__ void quad_share_position(Body (&B)[3], const Body& mine) {
  B[0].cx = quad_bcast<0>(mine.cx); B[0].cy = quad_bcast<0>(mine.cy); B[0].a = quad_bcast<0>(mine.a);
  B[1].cx = quad_bcast<1>(mine.cx); B[1].cy = quad_bcast<1>(mine.cy); B[1].a = quad_bcast<1>(mine.a);
  B[2].cx = quad_bcast<2>(mine.cx); B[2].cy = quad_bcast<2>(mine.cy); B[2].a = quad_bcast<2>(mine.a);
}

// ------------------------------------------------------------------ one physics step
// Applies the engine impulses for `action`, runs Collide + Solve, updates flags.
// `role` = lane & 3; mb = body this lane owns for contact work.
__device__ __forceinline__ void world_step(World& W, const Lds& lds, int role, int action, float disp0,
                                           float disp1, float fx, float fy, float& m_power, float& s_power) {
  Body(&B)[3] = W.b;
  const int mb = role == 3 ? 0 : role;
  const bool is_hull = mb == 0;
  const float mB_ = is_hull ? kInvM[0] : kInvM[1], iB_ = is_hull ? kInvI[0] : kInvI[1];
  const float lcy_ = is_hull ? kHullLcY : 0.0f;
  const float fr_ = is_hull ? sqrtf(0.1f * 0.1f) : sqrtf(0.2f * 0.1f);     // b2MixFriction with the terrain
  LUNAR_PROF_MARK(pt0);
  // ---- engines (gymnasium LunarLander.step) ----
  m_power = 0.0f; s_power = 0.0f;
  {
    float sn, cs;
    det_sincosf(B[0].a, &sn, &cs);
    const float tipx = sn, tipy = cs, sidex = -cs, sidey = sn;
    const float posx = B[0].cx - (cs * 0.0f - sn * kHullLcY);
    const float posy = B[0].cy - (sn * 0.0f + cs * kHullLcY);
    if (action == 2) {
      m_power = 1.0f;
      const float ox = tipx * (4.0f / kScale + 2.0f * disp0) + sidex * disp1;
      const float oy = -tipy * (4.0f / kScale + 2.0f * disp0) - sidey * disp1;
      const float ipx = posx + ox, ipy = posy + oy;
      const float Ix = -ox * 13.0f * m_power, Iy = -oy * 13.0f * m_power;
      B[0].vx += kInvM[0] * Ix; B[0].vy += kInvM[0] * Iy;
      B[0].w += kInvI[0] * cross2(ipx - B[0].cx, ipy - B[0].cy, Ix, Iy);
    }
    if (action == 1 || action == 3) {
      const float dir = (float)(action - 2);
      s_power = 1.0f;
      const float ox = tipx * disp0 + sidex * (3.0f * disp1 + dir * 12.0f / kScale);
      const float oy = -tipy * disp0 - sidey * (3.0f * disp1 + dir * 12.0f / kScale);
      const float ipx = posx + ox - tipx * 17.0f / kScale;
      const float ipy = posy + oy + tipy * 14.0f / kScale;
      const float Ix = -ox * 0.6f * s_power, Iy = -oy * 0.6f * s_power;
      B[0].vx += kInvM[0] * Ix; B[0].vy += kInvM[0] * Iy;
      B[0].w += kInvI[0] * cross2(ipx - B[0].cx, ipy - B[0].cy, Ix, Iy);
    }
  }

  // ---- Collide (own body): manifolds from the current transform, warm-start matching ----
  uint32_t my_touch = 0u;          // bit e: own body touches terrain edge e
  bool my_contact = false;
  {
    const Body me = select_body(B, mb);
    float px, py, qs, qc;
    body_xf(me, lcy_, px, py, qs, qc);
    // polygon x-extent -> candidate terrain edges (2 m wide each)
    float minx = 3.4e38f, maxx = -3.4e38f, miny = 3.4e38f;
    if (is_hull) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float x = (qc * kHullVx[i] - qs * kHullVy[i]) + px, y = (qs * kHullVx[i] + qc * kHullVy[i]) + py;
        minx = fminf(minx, x); maxx = fmaxf(maxx, x); miny = fminf(miny, y);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x = (qc * kLegVx[i] - qs * kLegVy[i]) + px, y = (qs * kLegVx[i] + qc * kLegVy[i]) + py;
        minx = fminf(minx, x); maxx = fmaxf(maxx, x); miny = fminf(miny, y);
      }
    }
    int e_lo = (int)floorf((minx - 2.0f * kPolyRadius) * 0.5f);
    int e_hi = (int)floorf((maxx + 2.0f * kPolyRadius) * 0.5f);
    e_lo = e_lo < 0 ? 0 : e_lo;
    e_hi = e_hi > 9 ? 9 : e_hi;
    // previous step's points of this body (for warm starting), before the slots are rewritten
    int ocount[2]; uint32_t okey[2][2]; float oni[2][2], oti[2][2];
#pragma unroll
    for (int os = 0; os < 2; ++os) {
      ocount[os] = (int)lds.mu(os, MF_COUNT);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        okey[os][q] = lds.mu(os, MF_P0 + 5 * q + P_KEY);
        oni[os][q] = lds.mf(os, MF_P0 + 5 * q + P_NI);
        oti[os][q] = lds.mf(os, MF_P0 + 5 * q + P_TI);
      }
    }
    const int old_e0 = W.edge0;
    W.edge0 = e_lo;
#pragma nounroll
    for (int s = 0; s < 2; ++s) {
      Manifold mf;
      mf.count = 0; mf.faceB = 0; mf.lnx = mf.lny = mf.lpx = mf.lpy = 0.0f;
      mf.p[0] = ContactPt{0.0f, 0.0f, 0u, 0.0f, 0.0f}; mf.p[1] = mf.p[0];
      const int e = e_lo + s;
      if (e <= e_hi && e >= 0 && e <= 9) {
        float y1 = W.ty[0], y2 = W.ty[1];
#pragma unroll
        for (int k = 0; k < 10; ++k) if (k == e) { y1 = W.ty[k]; y2 = W.ty[k + 1]; }
        const float x1 = 2.0f * (float)e, x2 = 2.0f * (float)(e + 1);
        if (miny - 2.0f * kPolyRadius <= fmaxf(y1, y2)) {
          const float ccx = is_hull ? me.cx : px, ccy = is_hull ? me.cy : py;  // centroid == centre of mass
          if (is_hull) collide_edge_polygon<6>(mf, kHullVx, kHullVy, kHullNx, kHullNy, px, py, qs, qc, ccx, ccy, x1, y1, x2, y2);
          else collide_edge_polygon<4>(mf, kLegVx, kLegVy, kLegNx, kLegNy, px, py, qs, qc, ccx, ccy, x1, y1, x2, y2);
        }
        // warm start: impulses of the same (edge, feature) from the previous step
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          mf.p[k].ni = 0.0f; mf.p[k].ti = 0.0f;
          if (k < mf.count) {
            bool found = false;   // b2Contact::Update: first old point with the same feature id
#pragma unroll
            for (int os = 0; os < 2; ++os) {
              if (old_e0 + os == e) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                  if (!found && q < ocount[os] && okey[os][q] == mf.p[k].key) {
                    mf.p[k].ni = oni[os][q]; mf.p[k].ti = oti[os][q]; found = true;
                  }
              }
            }
          }
        }
        if (mf.count > 0) { my_touch |= 1u << e; my_contact = true; }
      }
      lds.mu(s, MF_COUNT) = (uint32_t)mf.count; lds.mu(s, MF_FACEB) = (uint32_t)mf.faceB;
      lds.mf(s, MF_LNX) = mf.lnx; lds.mf(s, MF_LNY) = mf.lny;
      lds.mf(s, MF_LPX) = mf.lpx; lds.mf(s, MF_LPY) = mf.lpy;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        lds.mf(s, MF_P0 + 5 * k + P_LPX) = mf.p[k].lpx; lds.mf(s, MF_P0 + 5 * k + P_LPY) = mf.p[k].lpy;
        lds.mu(s, MF_P0 + 5 * k + P_KEY) = mf.p[k].key;
        lds.mf(s, MF_P0 + 5 * k + P_NI) = mf.p[k].ni; lds.mf(s, MF_P0 + 5 * k + P_TI) = mf.p[k].ti;
      }
    }
  }
  // contact listener (gymnasium ContactDetector) on the quad's combined touching mask
  const uint32_t touching_now = quad_bcast_u<0>(my_touch) | (quad_bcast_u<1>(my_touch) << 10) |
                                (quad_bcast_u<2>(my_touch) << 20);
  const bool any_contact = touching_now != 0u;       // uniform across the quad
  {
    const uint32_t began = touching_now & ~W.touching, ended = W.touching & ~touching_now;
    if (began & 0x3FFu) W.flags |= 4u;                       // hull touched the moon: game over
#pragma unroll
    for (int L = 0; L < 2; ++L) {
#pragma unroll
      for (int e = 0; e < 10; ++e) {
        const uint32_t bit = 1u << (10 * (L + 1) + e);
        if (began & bit) W.flags |= (1u << L);
        if (ended & bit) W.flags &= ~(1u << L);
      }
    }
    W.touching = touching_now;
  }

  LUNAR_PROF_MARK(pt1);
  LUNAR_PROF_ADD(lds, 0, pt0, pt1);          // engines + collide + listener
  // ---- Solve (b2Island::Solve) ----
  const float h = kDt;
  // integrate velocities: gravity (0,-10) + the pending reset force on the hull
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float Fx = b == 0 ? fx : 0.0f, Fy = b == 0 ? fy : 0.0f;
    B[b].vx += h * (0.0f + kInvM[b] * Fx);
    B[b].vy += h * (-10.0f + kInvM[b] * Fy);
  }

  // contact velocity constraints + warm start (own body), kept in VGPRs for the sweeps
  int vcn[2] = {0, 0};
  float vcf[2][16], vim[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int f = 0; f < 16; ++f) vcf[s][f] = 0.0f;
    vim[s][0] = vim[s][1] = vim[s][2] = vim[s][3] = 0.0f;
  }
  if (any_contact) {
    Body me = select_body(B, mb);
    if (my_contact) {
      float px, py, qs, qc;
      body_xf(me, lcy_, px, py, qs, qc);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int count = (int)lds.mu(s, MF_COUNT);
        int vcount = count;
        if (count > 0) {
          const bool faceB = lds.mu(s, MF_FACEB) != 0u;
          const float lnx = lds.mf(s, MF_LNX), lny = lds.mf(s, MF_LNY);
          const float lpx = lds.mf(s, MF_LPX), lpy = lds.mf(s, MF_LPY);
          float nx, ny, wpx[2], wpy[2];
          if (!faceB) {
            nx = lnx; ny = lny;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const float qx = lds.mf(s, MF_P0 + 5 * k + P_LPX), qy = lds.mf(s, MF_P0 + 5 * k + P_LPY);
              const float clx = (qc * qx - qs * qy) + px, cly = (qs * qx + qc * qy) + py;
              const float d = kPolyRadius - dot2(clx - lpx, cly - lpy, nx, ny);
              const float cAx = clx + d * nx, cAy = cly + d * ny;
              const float cBx = clx - kPolyRadius * nx, cBy = cly - kPolyRadius * ny;
              wpx[k] = 0.5f * (cAx + cBx); wpy[k] = 0.5f * (cAy + cBy);
            }
          } else {
            nx = qc * lnx - qs * lny; ny = qs * lnx + qc * lny;
            const float ppx = (qc * lpx - qs * lpy) + px, ppy = (qs * lpx + qc * lpy) + py;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const float clx = lds.mf(s, MF_P0 + 5 * k + P_LPX), cly = lds.mf(s, MF_P0 + 5 * k + P_LPY);
              const float d = kPolyRadius - dot2(clx - ppx, cly - ppy, nx, ny);
              const float cBx = clx + d * nx, cBy = cly + d * ny;
              const float cAx = clx - kPolyRadius * nx, cAy = cly - kPolyRadius * ny;
              wpx[k] = 0.5f * (cAx + cBx); wpy[k] = 0.5f * (cAy + cBy);
            }
            nx = -nx; ny = -ny;
          }
          const float tx = ny, ty = -nx;
          float rx[2], ry[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            rx[k] = wpx[k] - me.cx; ry[k] = wpy[k] - me.cy;
            const float rn = cross2(rx[k], ry[k], nx, ny);
            const float kn = mB_ + iB_ * rn * rn;
            vcf[s][VC_NM0 - 1 + k] = kn > 0.0f ? 1.0f / kn : 0.0f;
            const float rt = cross2(rx[k], ry[k], tx, ty);
            const float kt = mB_ + iB_ * rt * rt;
            vcf[s][VC_TM0 - 1 + k] = kt > 0.0f ? 1.0f / kt : 0.0f;
            vcf[s][VC_RX0 - 1 + 2 * k] = rx[k]; vcf[s][VC_RY0 - 1 + 2 * k] = ry[k];
          }
          vcf[s][VC_NX - 1] = nx; vcf[s][VC_NY - 1] = ny;
          if (count == 2) {
            const float rn1 = cross2(rx[0], ry[0], nx, ny), rn2 = cross2(rx[1], ry[1], nx, ny);
            const float k11 = mB_ + iB_ * rn1 * rn1, k22 = mB_ + iB_ * rn2 * rn2, k12 = mB_ + iB_ * rn1 * rn2;
            if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
              float det = k11 * k22 - k12 * k12;
              if (det != 0.0f) det = 1.0f / det;
              vcf[s][VC_K11 - 1] = k11; vcf[s][VC_K12 - 1] = k12; vcf[s][VC_K22 - 1] = k22;
              vcf[s][VC_I11 - 1] = det * k22; vcf[s][VC_I12 - 1] = -det * k12; vcf[s][VC_I22 - 1] = det * k11;
            } else {
              vcount = 1;   // nearly redundant second point: Box2D drops it from the velocity solve
            }
          }
          vim[s][0] = lds.mf(s, MF_P0 + P_NI); vim[s][1] = lds.mf(s, MF_P0 + P_TI);
          vim[s][2] = lds.mf(s, MF_P1 + P_NI); vim[s][3] = lds.mf(s, MF_P1 + P_TI);
          // warm start
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (k < vcount) {
              const float ni = vim[s][2 * k], ti = vim[s][2 * k + 1];
              const float Px = ni * nx + ti * tx, Py = ni * ny + ti * ty;
              me.w += iB_ * cross2(rx[k], ry[k], Px, Py);
              me.vx += mB_ * Px; me.vy += mB_ * Py;
            }
          }
        }
        vcn[s] = vcount;
      }
    }
    quad_share_velocity(B, me);
  }

  // joints: InitVelocityConstraints (island order: joint of legs[1], then legs[0]); replicated per lane
  float jrAx[2], jrAy[2], jrBx[2], jrBy[2];
  float K11[2], K12[2], K13[2], K22[2], K23[2], K33[2], mmass[2];
  {
    float qsA, qcA;
    det_sincosf(B[0].a, &qsA, &qcA);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int L = 1 - jj;
      const int bi = L + 1;
      float qsB, qcB;
      det_sincosf(B[bi].a, &qsB, &qcB);
      const float lax = 0.0f - 0.0f, lay = 0.0f - kHullLcY;
      const float lbx = leg_sign(L) * kLegAway, lby = kLegDown;
      jrAx[L] = qcA * lax - qsA * lay; jrAy[L] = qsA * lax + qcA * lay;
      jrBx[L] = qcB * lbx - qsB * lby; jrBy[L] = qsB * lbx + qcB * lby;
      const float mA = kInvM[0], mB = kInvM[bi], iA = kInvI[0], iB = kInvI[bi];
      K11[L] = mA + mB + jrAy[L] * jrAy[L] * iA + jrBy[L] * jrBy[L] * iB;
      K12[L] = -jrAy[L] * jrAx[L] * iA - jrBy[L] * jrBx[L] * iB;
      K13[L] = -jrAy[L] * iA - jrBy[L] * iB;
      K22[L] = mA + mB + jrAx[L] * jrAx[L] * iA + jrBx[L] * jrBx[L] * iB;
      K23[L] = jrAx[L] * iA + jrBx[L] * iB;
      K33[L] = iA + iB;
      mmass[L] = 1.0f / (iA + iB);
      Joint& J = W.j[L];
      const float ang = B[bi].a - B[0].a;
      if (ang <= joint_lower(L)) { if (J.state != 1) J.iz = 0.0f; J.state = 1; }
      else if (ang >= joint_upper(L)) { if (J.state != 2) J.iz = 0.0f; J.state = 2; }
      else { J.state = 0; J.iz = 0.0f; }
      // warm start (dtRatio = 1)
      const float Px = J.ix, Py = J.iy;
      B[0].vx -= mA * Px; B[0].vy -= mA * Py;
      B[0].w -= iA * (cross2(jrAx[L], jrAy[L], Px, Py) + J.im + J.iz);
      B[bi].vx += mB * Px; B[bi].vy += mB * Py;
      B[bi].w += iB * (cross2(jrBx[L], jrBy[L], Px, Py) + J.im + J.iz);
    }
  }

  LUNAR_PROF_MARK(pt2);
  LUNAR_PROF_ADD(lds, 1, pt1, pt2);          // constraint initialisation + warm start
  // Per-joint invariants of the sweeps, hoisted by hand (same operations, same order as b2Mat33::Solve33 /
  // b2Mat22 inside the sweep): first Cramer cofactor column, the two reciprocal determinants, the joint arms'
  // perpendiculars (w x r = w * (-ry, rx)), the limit state as a mask and the side of the limit as a sign.
  float jpAx[2], jpAy[2], jpBx[2], jpBy[2];
  float c1x[2], c1y[2], c1z[2], det3i[2], det2i[2], lim_sign[2];
  bool at_limit[2];
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    const float a11 = K11[L], a12 = K12[L], a13 = K13[L], a22 = K22[L], a23 = K23[L], a33 = K33[L];
    c1x[L] = a22 * a33 - a23 * a23; c1y[L] = a23 * a13 - a12 * a33; c1z[L] = a12 * a23 - a22 * a13;
    float det = a11 * c1x[L] + a12 * c1y[L] + a13 * c1z[L];
    if (det != 0.0f) det = 1.0f / det;
    det3i[L] = det;
    float d2 = a11 * a22 - a12 * a12;
    if (d2 != 0.0f) d2 = 1.0f / d2;
    det2i[L] = d2;
    jpAx[L] = -jrAy[L]; jpAy[L] = jrAx[L]; jpBx[L] = -jrBy[L]; jpBy[L] = jrBx[L];
    at_limit[L] = W.j[L].state != 0;                 // the state is fixed for the whole velocity solve
    lim_sign[L] = W.j[L].state == 1 ? 1.0f : -1.0f;  // lower limit: release when the impulse turns negative; upper: positive
  }
  // Velocity iterations.  The solver wave issues one VALU instruction per 4 clocks whatever it computes (measured:
  // tools/ubench/valu_latency.hip), so a sweep costs its instruction COUNT: the joint solve is written across the quad —
  // lane r of an env's four lanes holds component r of every 3-vector (x, y, angular) — and one instruction does for
  // (x, y, z) what three did when each lane carried a full copy: body velocities VA / VB, the accumulated impulse JI =
  // (ix, iy, iz), the Cramer cofactors and the 2 x 2 solve are lane vectors, dot products reduce over the quad with DPP
  // operands IN THE ORDER b2Mat33::Solve33 adds them, and each lane applies exactly the operations, on the same values,
  // that b2RevoluteJoint::SolveVelocityConstraints applies to its component (branch-free: the 3 x 3 solution at a limit
  // and the 2 x 2 one are both formed and selected).  ~65 VALU instructions per joint instead of ~105.
  const bool r0 = role == 0, r1 = role == 1, r2 = role == 2;
  auto lane3 = [&](float x, float y, float z) { return r0 ? x : (r1 ? y : (r2 ? z : 0.0f)); };
  float JPA[2], JPB[2], C1[2], P2[2], Q2[2], A2[2], P3[2], Q3[2], A3[2], A3C[2], D2[2], MBv[2], IBz[2];
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    const float a11 = K11[L], a12 = K12[L], a13 = K13[L], a22 = K22[L], a23 = K23[L], a33 = K33[L];
    JPA[L] = lane3(jpAx[L], jpAy[L], 0.0f); JPB[L] = lane3(jpBx[L], jpBy[L], 0.0f);
    C1[L] = lane3(c1x[L], c1y[L], c1z[L]);
    // second / third Cramer columns, rotated so that the three-term sums end in lane 1 / lane 2 in Solve33's order
    P2[L] = lane3(a23, a33, a13); Q2[L] = lane3(a13, a23, a33); A2[L] = lane3(a13, a11, a12);
    P3[L] = lane3(a23, a12, a22); Q3[L] = lane3(a12, a22, a23); A3[L] = lane3(a12, a13, a11);
    A3C[L] = lane3(a13, a23, 0.0f); D2[L] = lane3(a22, a11, 0.0f);
    MBv[L] = lane3(kInvM[L + 1], kInvM[L + 1], kInvI[L + 1]); IBz[L] = r2 ? kInvI[L + 1] : 0.0f;
  }
  const float MAv = lane3(kInvM[0], kInvM[0], kInvI[0]), IAz = r2 ? kInvI[0] : 0.0f;
  float VA = lane3(B[0].vx, B[0].vy, B[0].w);
  float VB[2] = {lane3(B[1].vx, B[1].vy, B[1].w), lane3(B[2].vx, B[2].vy, B[2].w)};
  float JI[2] = {lane3(W.j[0].ix, W.j[0].iy, W.j[0].iz), lane3(W.j[1].ix, W.j[1].iy, W.j[1].iz)};
  float IM[2] = {W.j[0].im, W.j[1].im};            // motor impulse: lane 2's copy is the joint's (the chain below runs on lane 2's angular velocities)
  const float maxImp = h * kMotorTorque;
  // The order of the statements below IS the instruction schedule (a scheduling barrier after each one): an instruction
  // that reads a register through DPP must not follow its producer by less than two issue slots, or the hardware wants
  // an s_nop 1 — which costs this lone wave two of the ~5-clock issue slots every instruction costs it.  The compiler's
  // own order had 29 of them in 125 instructions.
#define SB __builtin_amdgcn_sched_barrier(0)
#define USEL(c, a, b) (__builtin_unpredictable(c) ? (a) : (b))   // a select the compiler may not turn into a branch
  float zb[2];                                     // bcast(iz) * (a13, a23, 0): the limit impulse's share of the 2 x 2 right-hand side
  zb[1] = quad_perm<0xAA>(JI[1]) * A3C[1];
  zb[0] = 0.0f;
  float lim_nan[2] = {at_limit[0] ? lim_sign[0] : __builtin_nanf(""), at_limit[1] ? lim_sign[1] : __builtin_nanf("")};
  asm volatile("" : "+v"(lim_nan[0]), "+v"(lim_nan[1]));     // keep them registers (the compiler would redo the selects in the loop)
  auto joints = [&]() {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int L = 1 - jj;
      const bool lim = at_limit[L];
      // motor (lane 2 carries the angular velocities)
      float t = VB[L] - VA; SB;
      t = t - 0.3f * leg_sign(L); SB;
      float imp = -mmass[L] * t; SB;
      float nw = IM[L] + imp; SB;
      nw = __builtin_amdgcn_fmed3f(nw, -maxImp, maxImp); SB;          // == clampf for lo <= hi, no NaN
      imp = nw - IM[L]; IM[L] = nw; SB;
      const float mb_ = IBz[L] * imp; SB;
      VB[L] = VB[L] + mb_; SB;
      const float ma_ = IAz * imp; SB;
      VA = VA - ma_; SB;
      // Cdot = vB + wB x rB - vA - wA x rA (lanes 0, 1), wB - wA (lane 2: the arms' third components are +0)
      const float xb = quad_perm<0xAA>(VB[L]) * JPB[L]; SB;
      float c = VB[L] + xb; SB;
      const float xa = quad_perm<0xAA>(VA) * JPA[L]; SB;
      c = c - VA; SB;
      c = c - xa; SB;
      // impulse = -K^-1 * Cdot (b2Mat33::Solve33, Cramer): x ends in lane 0, y in lane 1, z in lane 2
      const float u1 = c * C1[L]; SB;
      const float zc = zb[L] - c; SB;
      const float q2 = quad_perm<kRot1>(c) * Q2[L]; SB;
      const float q3 = quad_perm<kRot2>(c) * Q3[L]; SB;
      const float m2 = c * P2[L] - q2; SB;
      const float m3 = c * P3[L] - q3; SB;
      float s1 = quad_perm<kRot1>(u1) + u1; SB;
      const float u2 = A2[L] * m2; SB;
      const float u3 = A3[L] * m3; SB;
      s1 = quad_perm<kRot2>(u1) + s1; SB;
      // right-hand side of the 2 x 2 solution: the point constraint alone (no limit), or with the limit impulse taken back
      const float rr = USEL(lim, zc, -c); SB;
      float s2 = quad_perm<kRot1>(u2) + u2; SB;
      float s3 = quad_perm<kRot1>(u3) + u3; SB;
      const float d2r = D2[L] * rr; SB;
      s2 = quad_perm<kRot2>(u2) + s2; SB;
      s3 = quad_perm<kRot2>(u3) + s3; SB;
      const float r2_ = quad_perm<kSwap01>(rr) * K12[L]; SB;
      float sel = USEL(r1, s2, s3); SB;
      sel = USEL(r0, s1, sel); SB;
      const float nsol = -(det3i[L] * sel); SB;
      const float newJ = JI[L] + nsol; SB;             // lane 2: the limit impulse after this sweep
      const float uu = d2r - r2_; SB;
      const float u = det2i[L] * uu; SB;
      // the limit releases when its impulse turns to the wrong side; `lim_nan` is lim_sign at a limit and NaN without one,
      // so that ONE unordered compare says "no limit, or the limit releases" (no SALU mask arithmetic in the sweep)
      const float rt = quad_perm<0xAA>(newJ) * lim_nan[L]; SB;
      const float altz = USEL(lim, -JI[L], 0.0f); SB;
      const float alt = USEL(r2, altz, u); SB;
      float ip;                                        // (ipx, ipy, ipz) = !(rt >= 0) ? alt : nsol — through VCC: a mask in an SGPR pair costs two wait states
      asm("v_cmp_nge_f32 vcc, %1, 0\n\tv_cndmask_b32_e32 %0, %2, %3, vcc" : "=v"(ip) : "v"(rt), "v"(nsol), "v"(alt) : "vcc"); SB;
      JI[L] = JI[L] + ip; SB;                          // lane 2: newJ, or iz + (-iz) = 0 on release
      zb[1 - L] = quad_perm<0xAA>(JI[1 - L]) * A3C[1 - L]; SB;           // for the OTHER joint's next solve (its JI is long written)
      const float ay = quad_perm<0x55>(ip) * jrAx[L]; SB;
      const float ax = quad_perm<0x00>(ip) * jrAy[L]; SB;
      const float by = quad_perm<0x55>(ip) * jrBx[L]; SB;
      const float bx = quad_perm<0x00>(ip) * jrBy[L]; SB;
      const float crA = ay - ax; SB;
      const float crB = by - bx; SB;
      const float wA = crA + ip; SB;
      const float wB = crB + ip; SB;
      float tA = USEL(r2, wA, ip); SB;
      float tB = USEL(r2, wB, ip); SB;
      tA = MAv * tA; SB;
      tB = MBv[L] * tB; SB;
      VA = VA - tA; SB;
      VB[L] = VB[L] + tB; SB;
    }
  };
#undef SB
#undef USEL
  auto contacts = [&]() {
    // contacts: each lane solves its own body's slots, then the quad exchanges velocities
    if (any_contact) {
      Body me;                                  // only the velocities are touched in a sweep
      me.cx = me.cy = me.a = 0.0f;
      me.vx = sel3(mb, quad_bcast<0>(VA), quad_bcast<0>(VB[0]), quad_bcast<0>(VB[1]));
      me.vy = sel3(mb, quad_bcast<1>(VA), quad_bcast<1>(VB[0]), quad_bcast<1>(VB[1]));
      me.w = sel3(mb, quad_bcast<2>(VA), quad_bcast<2>(VB[0]), quad_bcast<2>(VB[1]));
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int vcount = vcn[s];
        if (vcount > 0) {
          const float nx = vcf[s][VC_NX - 1], ny = vcf[s][VC_NY - 1], tx = ny, ty = -nx;
          const float rx0 = vcf[s][VC_RX0 - 1], ry0 = vcf[s][VC_RY0 - 1], rx1 = vcf[s][VC_RX1 - 1], ry1 = vcf[s][VC_RY1 - 1];
          float ni0 = vim[s][0], ti0 = vim[s][1], ni1 = vim[s][2], ti1 = vim[s][3];
          {  // friction, point 0
            const float dvx = me.vx + (-me.w * ry0), dvy = me.vy + (me.w * rx0);
            const float vt = dot2(dvx, dvy, tx, ty);
            float lam = vcf[s][VC_TM0 - 1] * (-vt);
            const float maxF = fr_ * ni0;
            const float nw = clampf(ti0 + lam, -maxF, maxF);
            lam = nw - ti0; ti0 = nw;
            const float Px = lam * tx, Py = lam * ty;
            me.vx += mB_ * Px; me.vy += mB_ * Py;
            me.w += iB_ * cross2(rx0, ry0, Px, Py);
          }
          if (vcount == 2) {  // friction, point 1
            const float dvx = me.vx + (-me.w * ry1), dvy = me.vy + (me.w * rx1);
            const float vt = dot2(dvx, dvy, tx, ty);
            float lam = vcf[s][VC_TM1 - 1] * (-vt);
            const float maxF = fr_ * ni1;
            const float nw = clampf(ti1 + lam, -maxF, maxF);
            lam = nw - ti1; ti1 = nw;
            const float Px = lam * tx, Py = lam * ty;
            me.vx += mB_ * Px; me.vy += mB_ * Py;
            me.w += iB_ * cross2(rx1, ry1, Px, Py);
          }
          if (vcount == 1) {
            const float dvx = me.vx + (-me.w * ry0), dvy = me.vy + (me.w * rx0);
            const float vn = dot2(dvx, dvy, nx, ny);
            float lam = -vcf[s][VC_NM0 - 1] * vn;
            const float nw = fmaxf(ni0 + lam, 0.0f);
            lam = nw - ni0; ni0 = nw;
            const float Px = lam * nx, Py = lam * ny;
            me.vx += mB_ * Px; me.vy += mB_ * Py;
            me.w += iB_ * cross2(rx0, ry0, Px, Py);
          } else {
            // 2-point block solver (b2ContactSolver::SolveVelocityConstraints)
            const float k11 = vcf[s][VC_K11 - 1], k12 = vcf[s][VC_K12 - 1], k22 = vcf[s][VC_K22 - 1];
            const float a1 = ni0, a2 = ni1;
            const float dv1x = me.vx + (-me.w * ry0), dv1y = me.vy + (me.w * rx0);
            const float dv2x = me.vx + (-me.w * ry1), dv2y = me.vy + (me.w * rx1);
            float vn1 = dot2(dv1x, dv1y, nx, ny), vn2 = dot2(dv2x, dv2y, nx, ny);
            const float b1 = vn1 - (k11 * a1 + k12 * a2);
            const float b2 = vn2 - (k12 * a1 + k22 * a2);
            float x1 = -(vcf[s][VC_I11 - 1] * b1 + vcf[s][VC_I12 - 1] * b2);
            float x2 = -(vcf[s][VC_I12 - 1] * b1 + vcf[s][VC_I22 - 1] * b2);
            bool ok = (x1 >= 0.0f && x2 >= 0.0f);
            if (!ok) {
              x1 = -vcf[s][VC_NM0 - 1] * b1; x2 = 0.0f;
              vn2 = k12 * x1 + b2;
              ok = (x1 >= 0.0f && vn2 >= 0.0f);
            }
            if (!ok) {
              x1 = 0.0f; x2 = -vcf[s][VC_NM1 - 1] * b2;
              vn1 = k12 * x2 + b1;
              ok = (x2 >= 0.0f && vn1 >= 0.0f);
            }
            if (!ok) {
              x1 = 0.0f; x2 = 0.0f;
              ok = (b1 >= 0.0f && b2 >= 0.0f);
            }
            if (ok) {
              const float d1 = x1 - a1, d2 = x2 - a2;
              const float P1x = d1 * nx, P1y = d1 * ny, P2x = d2 * nx, P2y = d2 * ny;
              me.vx += mB_ * (P1x + P2x); me.vy += mB_ * (P1y + P2y);
              me.w += iB_ * (cross2(rx0, ry0, P1x, P1y) + cross2(rx1, ry1, P2x, P2y));
              ni0 = x1; ni1 = x2;
            }
          }
          vim[s][0] = ni0; vim[s][1] = ti0;
          if (vcount == 2) { vim[s][2] = ni1; vim[s][3] = ti1; }
        }
      }
      VA = lane3(quad_bcast<0>(me.vx), quad_bcast<0>(me.vy), quad_bcast<0>(me.w));
      VB[0] = lane3(quad_bcast<1>(me.vx), quad_bcast<1>(me.vy), quad_bcast<1>(me.w));
      VB[1] = lane3(quad_bcast<2>(me.vx), quad_bcast<2>(me.vy), quad_bcast<2>(me.w));
    }
  };
  // a wave without a single contact (most steps of most workgroups) runs the sweeps without the per-sweep test
  if (__builtin_amdgcn_ballot_w64(any_contact) == 0ull) {
#pragma nounroll
    for (int it = 0; it < kVelIters; ++it) joints();
  } else {
#pragma nounroll
    for (int it = 0; it < kVelIters; ++it) { joints(); contacts(); }
  }
  B[0].vx = quad_bcast<0>(VA); B[0].vy = quad_bcast<1>(VA); B[0].w = quad_bcast<2>(VA);
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    B[L + 1].vx = quad_bcast<0>(VB[L]); B[L + 1].vy = quad_bcast<1>(VB[L]); B[L + 1].w = quad_bcast<2>(VB[L]);
    W.j[L].ix = quad_bcast<0>(JI[L]); W.j[L].iy = quad_bcast<1>(JI[L]); W.j[L].iz = quad_bcast<2>(JI[L]);
    W.j[L].im = quad_bcast<2>(IM[L]);
  }
  LUNAR_PROF_MARK(pt3);
  LUNAR_PROF_ADD(lds, 2, pt2, pt3);          // 180 velocity sweeps
#ifdef GYMRL_LUNAR_PROF
  const bool wave_contact = __builtin_amdgcn_ballot_w64(any_contact) != 0ull;   // the path the wave took (not lane 0's own env)
  LUNAR_PROF_ADD(lds, wave_contact ? 6 : 7, pt2, pt3);
#endif
  // b2ContactSolver::StoreImpulses
  if (my_contact) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (vcn[s] > 0) {
        lds.mf(s, MF_P0 + P_NI) = vim[s][0]; lds.mf(s, MF_P0 + P_TI) = vim[s][1];
        if (vcn[s] == 2) { lds.mf(s, MF_P1 + P_NI) = vim[s][2]; lds.mf(s, MF_P1 + P_TI) = vim[s][3]; }
      }
    }
  }

  // integrate positions
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    float tx = h * B[b].vx, ty = h * B[b].vy;
    if (dot2(tx, ty, tx, ty) > kMaxTranslation * kMaxTranslation) {
      const float ratio = kMaxTranslation / sqrtf(dot2(tx, ty, tx, ty));
      B[b].vx *= ratio; B[b].vy *= ratio;
    }
    const float rot = h * B[b].w;
    if (rot * rot > kMaxRotation * kMaxRotation) {
      const float ratio = kMaxRotation / fabsf(rot);
      B[b].w *= ratio;
    }
    B[b].cx += h * B[b].vx; B[b].cy += h * B[b].vy; B[b].a += h * B[b].w;
  }

  LUNAR_PROF_MARK(pt4);
  LUNAR_PROF_ADD(lds, 3, pt3, pt4);          // store impulses + integrate positions
  // position iterations
#ifdef GYMRL_LUNAR_PROF
  int prof_pos_iters = 0;
#endif
  bool position_solved = false;
  float plim_value[2], plim_slop[2], plim_lo[2], plim_hi[2];
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    const bool lower = W.j[L].state == 1;
    plim_value[L] = lower ? joint_lower(L) : joint_upper(L);
    plim_slop[L] = lower ? kAngularSlop : -kAngularSlop;             // C - slop == C + (-slop), exactly
    plim_lo[L] = lower ? -kMaxAngCorr : 0.0f;
    plim_hi[L] = lower ? 0.0f : kMaxAngCorr;
  }
  // The manifolds do not change during the position sweeps: read them from LDS once (a wave runs as many sweeps
  // as its slowest env needs — 19 on average, 49 in the slowest workgroup — and every LDS read in the sweep is
  // a dependent ~100-clock round trip).
  int pcnt[2] = {0, 0};
  bool pfaceB[2] = {false, false};
  float plnx[2], plny[2], plpx[2], plpy[2], pqx[2][2], pqy[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    plnx[s] = plny[s] = plpx[s] = plpy[s] = 0.0f;
    pqx[s][0] = pqx[s][1] = pqy[s][0] = pqy[s][1] = 0.0f;
    if (my_contact) {
      pcnt[s] = (int)lds.mu(s, MF_COUNT);                        // the position solver keeps every manifold point
      pfaceB[s] = lds.mu(s, MF_FACEB) != 0u;
      plnx[s] = lds.mf(s, MF_LNX); plny[s] = lds.mf(s, MF_LNY);
      plpx[s] = lds.mf(s, MF_LPX); plpy[s] = lds.mf(s, MF_LPY);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        pqx[s][k] = lds.mf(s, MF_P0 + 5 * k + P_LPX); pqy[s][k] = lds.mf(s, MF_P0 + 5 * k + P_LPY);
      }
    }
  }
#pragma nounroll
  for (int it = 0; it < kPosIters; ++it) {
    float min_sep = 0.0f;
    if (any_contact) {
      Body me = select_body(B, mb);
      float my_min = 0.0f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int cnt = pcnt[s];
        if (cnt > 0) {
          const bool faceB = pfaceB[s];
          const float lnx = plnx[s], lny = plny[s], lpx = plpx[s], lpy = plpy[s];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (k < cnt) {
              const float qx = pqx[s][k], qy = pqy[s][k];
              float px, py, qs, qc;
              body_xf(me, lcy_, px, py, qs, qc);
              float nx, ny, ptx, pty, sep;
              if (!faceB) {
                nx = lnx; ny = lny;
                const float clx = (qc * qx - qs * qy) + px, cly = (qs * qx + qc * qy) + py;
                sep = dot2(clx - lpx, cly - lpy, nx, ny) - kPolyRadius - kPolyRadius;
                ptx = clx; pty = cly;
              } else {
                nx = qc * lnx - qs * lny; ny = qs * lnx + qc * lny;
                const float ppx = (qc * lpx - qs * lpy) + px, ppy = (qs * lpx + qc * lpy) + py;
                sep = dot2(qx - ppx, qy - ppy, nx, ny) - kPolyRadius - kPolyRadius;
                ptx = qx; pty = qy;
                nx = -nx; ny = -ny;
              }
              const float rx = ptx - me.cx, ry = pty - me.cy;
              my_min = fminf(my_min, sep);
              const float C = clampf(kBaumgarte * (sep + kLinearSlop), -kMaxLinCorr, 0.0f);
              const float rn = cross2(rx, ry, nx, ny);
              const float K = mB_ + iB_ * rn * rn;
              const float imp = K > 0.0f ? -C / K : 0.0f;
              const float Px = imp * nx, Py = imp * ny;
              me.cx += mB_ * Px; me.cy += mB_ * Py;
              me.a += iB_ * cross2(rx, ry, Px, Py);
            }
          }
        }
      }
      quad_share_position(B, me);
      min_sep = fminf(fminf(quad_bcast<0>(my_min), quad_bcast<1>(my_min)), quad_bcast<2>(my_min));
    }
    const bool contacts_ok = min_sep >= -3.0f * kLinearSlop;
    bool joints_ok = true;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int L = 1 - jj;
      const int bi = L + 1;
      const float mA = kInvM[0], mB = kInvM[bi], iA = kInvI[0], iB = kInvI[bi];
      // limit part of b2RevoluteJoint::SolvePositionConstraints, branch-free: the limit's value, the slop's sign and the
      // clamp's bounds are per-lane constants of the step (plim_*), the very operations of the two branches follow
      const bool lim = at_limit[L];
      const float Cang = (B[bi].a - B[0].a) - plim_value[L];
      const float angErr = lim ? -(lim_sign[L] * Cang) : 0.0f;                 // lower: -C, upper: C
      const float Cc = clampf(Cang + plim_slop[L], plim_lo[L], plim_hi[L]);
      const float limImp = lim ? -mmass[L] * Cc : 0.0f;
      B[0].a -= iA * limImp; B[bi].a += iB * limImp;
      // the two rotations of the joint in ONE evaluation: even lanes of the quad take the hull's angle, odd lanes the leg's,
      // and the quad exchanges the results (every lane holds identical copies of both angles, so each lane gets exactly the
      // bits it would have computed itself — 26 instructions and four quad broadcasts instead of 52 on a path that a
      // grounded lander walks up to 60 times per step)
      float qsA, qcA, qsB, qcB;
      {
        float sn, cs;
        det_sincosf((role & 1) ? B[bi].a : B[0].a, &sn, &cs);
        qsA = quad_bcast<0>(sn); qcA = quad_bcast<0>(cs);
        qsB = quad_bcast<1>(sn); qcB = quad_bcast<1>(cs);
      }
      const float lax = 0.0f, lay = 0.0f - kHullLcY;
      const float lbx = leg_sign(L) * kLegAway, lby = kLegDown;
      const float rAx = qcA * lax - qsA * lay, rAy = qsA * lax + qcA * lay;
      const float rBx = qcB * lbx - qsB * lby, rBy = qsB * lbx + qcB * lby;
      const float Cx = B[bi].cx + rBx - B[0].cx - rAx, Cy = B[bi].cy + rBy - B[0].cy - rAy;
      // posErr = sqrtf(Cx * Cx + Cy * Cy) is only compared with the slop: compare the square with the exact threshold
      // instead (the correctly rounded square root is ~20 instructions on a path a grounded lander walks 60 x per step)
      const float posErr2 = Cx * Cx + Cy * Cy;
      const float k11 = mA + mB + iA * rAy * rAy + iB * rBy * rBy;
      const float k12 = -iA * rAx * rAy - iB * rBx * rBy;
      const float k22 = mA + mB + iA * rAx * rAx + iB * rBx * rBx;
      float det = k11 * k22 - k12 * k12;
      if (det != 0.0f) det = 1.0f / det;
      const float impx = -(det * (k22 * Cx - k12 * Cy)), impy = -(det * (k11 * Cy - k12 * Cx));
      B[0].cx -= mA * impx; B[0].cy -= mA * impy;
      B[0].a -= iA * cross2(rAx, rAy, impx, impy);
      B[bi].cx += mB * impx; B[bi].cy += mB * impy;
      B[bi].a += iB * cross2(rBx, rBy, impx, impy);
      joints_ok = joints_ok && (posErr2 <= kLinearSlopSqMax) && (angErr <= kAngularSlop);
#ifdef GYMRL_LUNAR_PROF
      if (it == 10 && lds.prof && role == 0) {       // what keeps an env iterating: per cause, summed over envs
        if (!(posErr2 <= kLinearSlopSqMax)) atomicAdd(&lds.prof[14], 1ull);
        if (!(angErr <= kAngularSlop)) atomicAdd(&lds.prof[15], 1ull);
      }
#endif
    }
#ifdef GYMRL_LUNAR_PROF
    if (it == 10 && lds.prof && role == 0) {
      if (!contacts_ok) atomicAdd(&lds.prof[13], 1ull);
      atomicAdd(&lds.prof[5], 1ull);                  // env-steps still iterating at iteration 10
    }
#endif
    if (contacts_ok && joints_ok) { position_solved = true; break; }
#ifdef GYMRL_LUNAR_PROF
    ++prof_pos_iters;
#endif
  }
#ifdef GYMRL_LUNAR_PROF
  {
    int wmax = prof_pos_iters + (position_solved ? 1 : 0), mine = wmax;
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off, 64));
    if (lds.prof && (threadIdx.x & 63) == 0) { lds.prof[11] += (unsigned long long)wmax; lds.prof[12] += (unsigned long long)mine; }
    if (lds.envp && role == 0) {
      lds.envp[0] += any_contact ? 1ull : 0ull;
      lds.envp[lds.envn] += (unsigned long long)mine;
      lds.envp[2 * (size_t)lds.envn] += wave_contact ? 1ull : 0ull;
    }
  }
#endif

  LUNAR_PROF_MARK(pt5);
  LUNAR_PROF_ADD(lds, 4, pt4, pt5);          // position iterations (until every env of the wave is solved)
  // sleep management (island = hull + both legs)
  float min_sleep = 3.4e38f;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    if (B[b].w * B[b].w > kAngSleepTol * kAngSleepTol ||
        dot2(B[b].vx, B[b].vy, B[b].vx, B[b].vy) > kLinSleepTol * kLinSleepTol) {
      W.sleep[b] = 0.0f; min_sleep = 0.0f;
    } else {
      W.sleep[b] += h;
      min_sleep = fminf(min_sleep, W.sleep[b]);
    }
  }
  if (min_sleep >= kTimeToSleep && position_solved) W.flags |= 16u;
}

// ------------------------------------------------------------------ gymnasium layer
__device__ __forceinline__ void lander_obs(const World& W, float (&o)[8]) {
  float sn, cs;
  det_sincosf(W.b[0].a, &sn, &cs);
  const float posx = W.b[0].cx - (cs * 0.0f - sn * kHullLcY);
  const float posy = W.b[0].cy - (sn * 0.0f + cs * kHullLcY);
  o[0] = (posx - kW / 2.0f) / (kW / 2.0f);
  o[1] = (posy - (kHelipadY + kLegDown)) / (kH / 2.0f);
  o[2] = W.b[0].vx * (kW / 2.0f) / 50.0f;
  o[3] = W.b[0].vy * (kH / 2.0f) / 50.0f;
  o[4] = W.b[0].a;
  o[5] = 20.0f * W.b[0].w / 50.0f;
  o[6] = (W.flags & 1u) ? 1.0f : 0.0f;
  o[7] = (W.flags & 2u) ? 1.0f : 0.0f;
}

__device__ __forceinline__ float shaping_of(const float (&o)[8]) {
  return -100.0f * sqrtf(o[0] * o[0] + o[1] * o[1]) - 100.0f * sqrtf(o[2] * o[2] + o[3] * o[3]) -
         100.0f * fabsf(o[4]) + 10.0f * o[6] + 10.0f * o[7];
}

// gymnasium step(): returns reward / terminated, fills o (replicated on the quad's lanes).
__device__ __forceinline__ void env_step_once(World& W, const Lds& lds, int role, int action, uint64_t seed,
                                              uint64_t env, uint32_t episode, uint32_t step_idx, float fx,
                                              float fy, float (&o)[8], float& reward, bool& terminated) {
  const u32x4 r = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_STEP | step_idx);
  const float d0 = (-1.0f + 2.0f * u01f(r.x)) / kScale, d1 = (-1.0f + 2.0f * u01f(r.y)) / kScale;
  float mp, sp;
  world_step(W, lds, role, action, d0, d1, fx, fy, mp, sp);
  lander_obs(W, o);
  const float shaping = shaping_of(o);
  reward = 0.0f;
  if (W.flags & 8u) reward = shaping - W.prev_shaping;
  W.prev_shaping = shaping; W.flags |= 8u;
  reward -= mp * 0.30f;
  reward -= sp * 0.03f;
  terminated = false;
  if ((W.flags & 4u) || fabsf(o[0]) >= 1.0f) { terminated = true; reward = -100.0f; }
  if (W.flags & 16u) { terminated = true; reward = 100.0f; }
}

// gymnasium reset() minus its trailing step(0): terrain, bodies, random initial force
// (fx, fy) that the caller feeds to the first world_step.
__device__ __forceinline__ void init_episode(World& W, const Lds& lds, uint64_t seed, uint64_t env,
                                             uint32_t episode, float& fx, float& fy) {
  float hgt[12];
#pragma unroll
  for (int blk = 0; blk < 3; ++blk) {
    const u32x4 r = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | (uint32_t)blk);
    hgt[4 * blk + 0] = (kH / 2.0f) * u01f(r.x); hgt[4 * blk + 1] = (kH / 2.0f) * u01f(r.y);
    hgt[4 * blk + 2] = (kH / 2.0f) * u01f(r.z); hgt[4 * blk + 3] = (kH / 2.0f) * u01f(r.w);
  }
#pragma unroll
  for (int i = 3; i <= 7; ++i) hgt[i] = kHelipadY;
#pragma unroll
  for (int i = 0; i < 11; ++i) {
    const float prev = hgt[i == 0 ? 11 : i - 1];   // python's height[-1] wrap at i = 0
    W.ty[i] = 0.33f * (prev + hgt[i] + hgt[i + 1]);
  }
  const u32x4 rf = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 3u);
  fx = -1000.0f + 2000.0f * u01f(rf.x); fy = -1000.0f + 2000.0f * u01f(rf.y);
  // hull at (W/2, H) angle 0: centre of mass = origin + localCenter
  W.b[0] = Body{kW / 2.0f, kH + kHullLcY, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    const float i = leg_sign(L);
    W.b[L + 1] = Body{kW / 2.0f - i * kLegAway, kH, i * 0.05f, 0.0f, 0.0f, 0.0f};
    W.j[L] = Joint{0.0f, 0.0f, 0.0f, 0.0f, 0};
  }
#pragma unroll
  for (int b = 0; b < 3; ++b) W.sleep[b] = 0.0f;
  W.edge0 = 0;
#pragma nounroll
  for (int k = 0; k < 2 * kMfWords; ++k) lds.w[k * kEnvBlock] = 0u;
  W.touching = 0u; W.flags = 0u; W.prev_shaping = 0.0f;
}

// ------------------------------------------------------------------ SoA state in HBM
// Everything an env needs between steps; [field][N] dword arrays.  Word order:
//   bodies 18 | sleep 3 | joints 10 | manifolds 3 bodies x 2 slots x 16 | edge0 3 | touching |
//   terrain 11 | flags | prev_shaping
constexpr int kWorldWords = 18 + 3 + 10 + 3 * 2 * kMfWords + 3 + 1 + 11 + 1 + 1;   // 144
constexpr int kWMf = 31, kWEdge0 = 31 + 96, kWTail = 31 + 96 + 3;

// A second world per env ("spare") holds the NEXT episode's post-reset state, computed by
// gymrl_env_refill off the critical path (reset() costs a full solver pass because it ends
// with step(0)); the step kernel swaps it in when an episode ends and falls back to the
// inline reset when the spare is not ready.  Bit-identical either way: the spare is a pure
// function of (seed, env id, episode).
constexpr uint32_t kNoSpare = 0xFFFFFFFFu;

struct LunarState {
  uint32_t* words;     // [kWorldWords][N]
  EpisodeFields ep;
  uint32_t* spare_words;    // [kWorldWords][N]
  float* spare_obs;         // [8][N]
  uint32_t* spare_episode;  // episode index the spare was built for, or kNoSpare
  size_t bytes;
  __host__ __device__ LunarState(void* buf, int n) {
    Carver c(buf, n);
    words = c.take<uint32_t>(kWorldWords);
    ep.ep_ret = c.take<double>(); ep.ep_len = c.take<int32_t>(); ep.episode = c.take<uint32_t>();
    spare_words = c.take<uint32_t>(kWorldWords);
    spare_obs = c.take<float>(8);
    spare_episode = c.take<uint32_t>();
    bytes = c.off;
  }
};

struct WordIO {
  uint32_t* base; int n; int i; int k; bool wr;   // wr: this lane performs stores
  __device__ __forceinline__ void f(float& x, bool store) {
    if (store) { if (wr) base[(size_t)k * n + i] = __float_as_uint(x); } else x = __uint_as_float(base[(size_t)k * n + i]);
    ++k;
  }
  __device__ __forceinline__ void u(uint32_t& x, bool store) {
    if (store) { if (wr) base[(size_t)k * n + i] = x; } else x = base[(size_t)k * n + i];
    ++k;
  }
  __device__ __forceinline__ void d(int& x, bool store) {
    if (store) { if (wr) base[(size_t)k * n + i] = (uint32_t)x; } else x = (int)base[(size_t)k * n + i];
    ++k;
  }
};

// Loads: every lane of the quad reads the replicated words and its own body's manifolds.
// Stores: lane 0 writes the replicated words, lanes 0..2 their body's manifolds + edge0.
__device__ __forceinline__ void world_io(World& W, const Lds& lds, int role, uint32_t* base, int n, int i,
                                         bool store) {
  const int mb = role == 3 ? 0 : role;
  WordIO io{base, n, i, 0, role == 0};
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    io.f(W.b[b].cx, store); io.f(W.b[b].cy, store); io.f(W.b[b].a, store);
    io.f(W.b[b].vx, store); io.f(W.b[b].vy, store); io.f(W.b[b].w, store);
  }
#pragma unroll
  for (int b = 0; b < 3; ++b) io.f(W.sleep[b], store);
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    io.f(W.j[L].ix, store); io.f(W.j[L].iy, store); io.f(W.j[L].iz, store); io.f(W.j[L].im, store);
    io.d(W.j[L].state, store);
  }
  // own body's manifold words stream HBM <-> LDS without touching the register file
  {
    WordIO mo{base, n, i, kWMf + mb * 2 * kMfWords, role < 3};
#pragma nounroll
    for (int k = 0; k < 2 * kMfWords; ++k) mo.u(lds.w[k * kEnvBlock], store);
    WordIO eo{base, n, i, kWEdge0 + mb, role < 3};
    eo.d(W.edge0, store);
  }
  io.k = kWTail;
  io.u(W.touching, store);
#pragma unroll
  for (int k = 0; k < 11; ++k) io.f(W.ty[k], store);
  io.u(W.flags, store);
  io.f(W.prev_shaping, store);
}

// The persistent rollout kernels keep a world in LDS between the steps of one launch: the registers' part is parked in
// this lane's column of `lds.park` (the manifold words already live in `lds.w` and simply stay there), so that only the
// first step of a launch reads the [word][N] state from HBM and only the last one writes it back — 6.9 + 3.1 us of a
// ~135-us vector step were that round trip through L2 (profiles/r03_rollout_state_io.txt).  The episode bookkeeping the
// step reads (episode, length, return) is parked with it; its global copy is still written every step.
constexpr int kParkWords = 31 + 1 + 14 + 4;
__device__ __forceinline__ void world_park(World& W, const Lds& lds, bool store, uint32_t& episode, int& len, double& ret) {
  int k = 0;
  auto f = [&](float& x) { if (store) lds.park[k * kEnvBlock] = __float_as_uint(x); else x = __uint_as_float(lds.park[k * kEnvBlock]); ++k; };
  auto u = [&](uint32_t& x) { if (store) lds.park[k * kEnvBlock] = x; else x = lds.park[k * kEnvBlock]; ++k; };
  auto d = [&](int& x) { if (store) lds.park[k * kEnvBlock] = (uint32_t)x; else x = (int)lds.park[k * kEnvBlock]; ++k; };
#pragma unroll
  for (int b = 0; b < 3; ++b) { f(W.b[b].cx); f(W.b[b].cy); f(W.b[b].a); f(W.b[b].vx); f(W.b[b].vy); f(W.b[b].w); }
#pragma unroll
  for (int b = 0; b < 3; ++b) f(W.sleep[b]);
#pragma unroll
  for (int L = 0; L < 2; ++L) { f(W.j[L].ix); f(W.j[L].iy); f(W.j[L].iz); f(W.j[L].im); d(W.j[L].state); }
  d(W.edge0);
  u(W.touching);
#pragma unroll
  for (int t = 0; t < 11; ++t) f(W.ty[t]);
  u(W.flags);
  f(W.prev_shaping);
  u(episode); d(len);
  uint32_t lo = (uint32_t)(unsigned long long)__double_as_longlong(ret), hi = (uint32_t)((unsigned long long)__double_as_longlong(ret) >> 32);
  u(lo); u(hi);
  if (!store) ret = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// obs [N][8]: lanes 0/1 of each quad write the two 16-B halves of their env's row, so a
// wave's 32 storing lanes cover 512 contiguous bytes.
__device__ __forceinline__ void store_obs_quad(float* __restrict__ dst, int env, int role, const float (&o)[8]) {
  if (role == 0) reinterpret_cast<float4*>(dst)[2 * (size_t)env] = make_float4(o[0], o[1], o[2], o[3]);
  if (role == 1) reinterpret_cast<float4*>(dst)[2 * (size_t)env + 1] = make_float4(o[4], o[5], o[6], o[7]);
}

// Builds the spare world of episode `want` of env i (reset() of that episode incl. its trailing step(0)) and
// publishes it: the body of lunar_refill_kernel, shared with the persistent rollout kernel (whose wave 1
// runs it while wave 0 steps).  Call with a quad-uniform `need`.
__device__ __forceinline__ void lunar_refill_quad(const LunarState& st, const Lds& lds, int n, int i, int role, bool need,
                                                  uint32_t want, uint64_t seed, int64_t env_id0) {
  if (need) {
    World W;
    float fx, fy, rew, o[8]; bool term;
    const uint64_t env = (uint64_t)(env_id0 + i);
    init_episode(W, lds, seed, env, want, fx, fy);
    env_step_once(W, lds, role, 0, seed, env, want, 0u, fx, fy, o, rew, term);
    world_io(W, lds, role, st.spare_words, n, i, true);
    if (role < 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) st.spare_obs[(size_t)(4 * role + k) * n + i] = o[4 * role + k];
    }
    __threadfence();                      // world before flag (a concurrent step may poll it)
  }
  // the flag must follow every lane's stores: lanes 1,2 wrote manifold words
  __builtin_amdgcn_wave_barrier();
  if (need && role == 0) __hip_atomic_store(&st.spare_episode[i], want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// One env.step() of env i by its quad (ppo_lunarlander.py:211 + the reset-on-done of :220-223): the body of
// lunar_step_kernel, shared with the persistent rollout kernel.  `act` is the already loaded action;
// o_next receives the next policy input (post-reset where the episode ended).
struct StepOut {
  float* obs_out; float* term_obs_out; float* rew_out; uint8_t* terminated_out; uint8_t* truncated_out;
  uint8_t* done_out; float* ep_ret_out; int32_t* ep_len_out; double* ep_stats;
};

// `refill` (wave-uniform): instead of stepping, the wave builds the spare world of episode episode[i] + 1 of every env that
// lacks one — the persistent rollout kernels run this on wave 1 while wave 0 steps (reset() ends with a full physics step,
// 180 velocity sweeps, which wave 0 would otherwise run inline on the step an episode ends, every other env of the wave
// waiting).  Both modes share ONE inlined copy of the solver (the pass loop below): the kernel's registers and code size
// are those of the stepping path alone.  A spare is a pure function of (seed, env id, episode), so the slab is bit-identical
// with or without it; `episode[i]` may be advanced by wave 0 in this very step — then the spare is for an episode that
// already started, is never matched, and is rebuilt on the next step.
__device__ __forceinline__ void lunar_step_quad(const LunarState& st, const Lds& lds, int n, int i, int role, bool valid,
                                                int action_in, uint64_t seed, int64_t env_id0, const StepOut& out,
                                                float (&o_next)[8], bool refill = false, int io_mode = 0) {
  // io_mode (stepping waves of the persistent kernels): bit 0 = the world comes from the LDS parking area, bit 1 = it goes back there
  float* __restrict__ obs_out = out.obs_out; float* __restrict__ term_obs_out = out.term_obs_out;
  float* __restrict__ rew_out = out.rew_out; uint8_t* __restrict__ terminated_out = out.terminated_out;
  uint8_t* __restrict__ truncated_out = out.truncated_out; uint8_t* __restrict__ done_out = out.done_out;
  float* __restrict__ ep_ret_out = out.ep_ret_out; int32_t* __restrict__ ep_len_out = out.ep_len_out;
  double* __restrict__ ep_stats = out.ep_stats;
  const bool lead = role == 0;
  bool done = false;
  double ret = 0.0; int len = 0;
  bool active = valid;
  uint32_t want = 0u;
  if (refill) {                                       // which envs lack the spare of their next episode (quad-uniform: lane 0's view)
    uint32_t need = 0u;
    if (valid) {
      want = __hip_atomic_load(&st.ep.episode[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      need = __hip_atomic_load(&st.spare_episode[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want ? 1u : 0u;
    }
    want = quad_bcast_u<0>(want);
    active = quad_bcast_u<0>(need) != 0u;
  }
  if (active) {
    World W;
    float o_term[8];
    const uint64_t env = (uint64_t)(env_id0 + i);
    uint32_t episode = 0u;
    double ret0 = 0.0;
    int act = 0;
    if (!refill) {
      if (io_mode & 1) world_park(W, lds, false, episode, len, ret0);
      else {
        world_io(W, lds, role, st.words, n, i, false);
        episode = st.ep.episode[i];
        len = st.ep.ep_len[i];
        ret0 = st.ep.ep_ret[i];
      }
      act = action_in;
      act = act < 0 ? 0 : (act > 3 ? 3 : act);
    }
    uint32_t ep = refill ? want : episode, step_idx = refill ? 0u : (uint32_t)len;
    float fx = 0.0f, fy = 0.0f;
    // pass 0 = the requested step; pass 1 = a new episode's reset(), whose trailing step(0) reuses the one inlined solver:
    // stepping waves enter it only where the episode ended and no spare world is ready, a refill wave starts there.
#pragma nounroll
    for (int pass = refill ? 1 : 0; pass < 2; ++pass) {
      float o[8], reward; bool terminated;
      if (pass == 1) init_episode(W, lds, seed, env, ep, fx, fy);
      env_step_once(W, lds, role, act, seed, env, ep, step_idx, fx, fy, o, reward, terminated);
      if (pass == 0) {
        len += 1;
        const bool truncated = len >= kMaxSteps;
        done = terminated || truncated;
        ret = ret0 + (double)reward;
        if (lead) {
          rew_out[i] = reward;
          if (terminated_out) terminated_out[i] = terminated;
          if (truncated_out) truncated_out[i] = truncated;
          if (done_out) done_out[i] = done;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { o_term[k] = o[k]; o_next[k] = o[k]; }
        if (!done) { if (lead) { st.ep.ep_ret[i] = ret; st.ep.ep_len[i] = len; } break; }
        ep = episode + 1u; step_idx = 0u; act = 0;
        // every lane reads the spare flag BEFORE lane 0 rewrites the episode bookkeeping (agent scope: the flag may have
        // been published by the workgroup's refill wave a moment ago — a per-CU cache line from an earlier poll must not answer)
        const bool have_spare = __hip_atomic_load(&st.spare_episode[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ep;
        if (lead) {
          if (ep_ret_out) ep_ret_out[i] = (float)ret;
          if (ep_len_out) ep_len_out[i] = len;
          st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0;
          // without a spare nothing is read from the spare slot, so the new episode number may be published at once
          if (!have_spare) __hip_atomic_store(&st.ep.episode[i], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (have_spare) {                            // next episode already prepared off the critical path
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          world_io(W, lds, role, st.spare_words, n, i, false);
#pragma unroll
          for (int k = 0; k < 8; ++k) o_next[k] = st.spare_obs[(size_t)k * n + i];
          // consume, THEN publish: the refill wave overwrites the spare slot as soon as it reads episode == ep, so the
          // store is a release that follows every lane's loads of the slot (one wave: the release's vmcnt(0) wait covers
          // the other three lanes of the quad) — the handshake no longer rests on the refill taking longer than these loads
          __builtin_amdgcn_wave_barrier();
          if (lead) __hip_atomic_store(&st.ep.episode[i], ep, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) o_next[k] = o[k];
      }
    }
    if (!refill) {
      if (io_mode & 2) {
        uint32_t p_ep = done ? episode + 1u : episode;
        int p_len = done ? 0 : len;
        double p_ret = done ? 0.0 : ret;
        world_park(W, lds, true, p_ep, p_len, p_ret);
      } else world_io(W, lds, role, st.words, n, i, true);
      store_obs_quad(obs_out, i, role, o_next);
      if (term_obs_out) store_obs_quad(term_obs_out, i, role, o_term);
    } else {                                          // publish the spare: world, observation, then the flag
      world_io(W, lds, role, st.spare_words, n, i, true);
      if (role < 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) st.spare_obs[(size_t)(4 * role + k) * n + i] = o_next[4 * role + k];
      }
      __threadfence();                                // world before flag (the stepping wave may poll it)
    }
  }
  if (refill) {
    __builtin_amdgcn_wave_barrier();                  // the flag must follow every lane's stores: lanes 1, 2 wrote manifold words
    if (active && lead) __hip_atomic_store(&st.spare_episode[i], want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  accumulate_ep_stats(ep_stats, done && lead, ret, len);
}

}  // namespace lunar
}  // namespace gymrl
