/*
 * lunar_oracle.c — CPU restatement of LunarLander-v3 (discrete, no wind).
 * TEST INFRASTRUCTURE ONLY (see gymrl_oracle.c header).
 *
 * What the reference calls: gym.make("LunarLander-v3").reset()/.step(a)
 * (ppo_lunarlander.py:160,200,211,222).  That arithmetic is gymnasium's
 * lunar_lander.py driving Box2D 2.3 — third-party, not under /root/reference,
 * not installable here: PARITY UNPINNED for this file.  It restates the published
 * structure of both (names in comments are Box2D's / gymnasium's):
 *   world:   gravity (0,-10); hull polygon (density 5, friction 0.1) + two leg boxes
 *            (density 1, friction 0.2) on revolute joints (motor 40 N*m at +-0.3
 *            rad/s, limits [0.4,0.9] / [-0.9,-0.4]); terrain = 10 edges, friction 0.1
 *   Step:    b2World::Step(1/50, 180, 60): Collide (b2CollideEdgeAndPolygon, feature-id
 *            warm starting, Begin/EndContact), b2Island::Solve (integrate velocities,
 *            contact + joint warm start, 180 sequential-impulse sweeps with the 2-point
 *            block solver, integrate positions with translation/rotation caps, <= 60
 *            Baumgarte position sweeps with early exit), sleep bookkeeping
 *   env:     engine impulses with 2 dispersion draws per step, 8-dim observation,
 *            shaping reward, -100 crash / out of bounds, +100 asleep, TimeLimit 1000
 * Plain C, one env at a time, structs and loops; every f32 operation is written in
 * the order the HIP kernel (gymrl_amd/csrc/env_lunar.hip) uses so that the two can be
 * compared bit for bit.  Deliberate simplifications shared by both sides (documented
 * in DESIGN.md): the y=0 base edge of the moon body and engine particles are omitted;
 * island order is (legs[1], hull, legs[0]); per body only the (at most two) terrain
 * edges under its x-extent are tested, in ascending order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void orc_philox(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
void orc_sincosf(float x, float* s, float* c);

#define RNG_ENV_RESET 0x10000000u
#define RNG_ENV_STEP  0x20000000u
static float u01f(uint32_t x) { return (float)(x >> 8) * 0x1p-24f; }

/* --------------------------------------------------------------- constants */
static const float SCALE = 30.0f;
#define WV (600.0f / 30.0f)
#define HV (400.0f / 30.0f)
static const float HELIPAD_Y = HV / 4.0f;
static const float LEG_DOWN = 18.0f / 30.0f, LEG_AWAY = 20.0f / 30.0f;
static const float DT = 1.0f / 50.0f;
enum { VEL_ITERS = 180, POS_ITERS = 60, MAX_STEPS = 1000 };
static const float LINEAR_SLOP = 0.005f;
#define PI_F 3.14159265359f
static const float ANGULAR_SLOP = 2.0f / 180.0f * PI_F;
static const float POLY_RADIUS = 2.0f * 0.005f;
static const float MAX_LIN_CORR = 0.2f, MAX_ANG_CORR = 8.0f / 180.0f * PI_F, BAUMGARTE = 0.2f;
static const float MAX_TRANSLATION = 2.0f, MAX_ROTATION = 0.5f * PI_F;
static const float TIME_TO_SLEEP = 0.5f, LIN_SLEEP_TOL = 0.01f, ANG_SLEEP_TOL = 2.0f / 180.0f * PI_F;
static const float MOTOR_TORQUE = 40.0f;

/* body 0 = hull, 1 = legs[0] (i=-1), 2 = legs[1] (i=+1); b2PolygonShape::ComputeMass results */
static const float INV_M[3] = {0.20761245674740486f, 14.0625f, 14.0625f};
static const float INV_I[3] = {1.2757043935679302f, 558.3639705882352f, 558.3639705882352f};
static const float HULL_LCY = 0.10130718954248369f;
static const float HULL_VX[6] = {17.f / 30, 17.f / 30, 14.f / 30, -14.f / 30, -17.f / 30, -17.f / 30};
static const float HULL_VY[6] = {-10.f / 30, 0.f, 17.f / 30, 17.f / 30, 0.f, -10.f / 30};
static const float HULL_NX[6] = {1.0f, 0.9847835588179369f, 0.0f, -0.9847835588179369f, -1.0f, 0.0f};
static const float HULL_NY[6] = {0.0f, 0.17378533390904766f, 1.0f, 0.17378533390904766f, 0.0f, -1.0f};
static const float LEG_VX[4] = {-2.f / 30, 2.f / 30, 2.f / 30, -2.f / 30};
static const float LEG_VY[4] = {-8.f / 30, -8.f / 30, 8.f / 30, 8.f / 30};
static const float LEG_NX[4] = {0.0f, 1.0f, 0.0f, -1.0f};
static const float LEG_NY[4] = {-1.0f, 0.0f, 1.0f, 0.0f};

typedef struct { float cx, cy, a, vx, vy, w; } body_t;
typedef struct { float ix, iy, iz, im; int state; } joint_t;
typedef struct { float lpx, lpy; uint32_t key; float ni, ti; } cpoint_t;
typedef struct { int count, faceB; float lnx, lny, lpx, lpy; cpoint_t p[2]; } manifold_t;
typedef struct {   /* b2ContactVelocityConstraint */
  int count; float nx, ny, rx[2], ry[2], nmass[2], tmass[2], k11, k12, k22, i11, i12, i22;
} vc_t;

typedef struct {
  body_t b[3];
  float sleep[3];
  joint_t j[2];
  manifold_t m[3][2];
  int edge0[3];
  uint32_t touching;
  float ty[11];
  uint32_t flags;        /* bit0 legs[0] contact, bit1 legs[1], bit2 game_over, bit3 prev_shaping set, bit4 asleep */
  float prev_shaping;
  double ep_ret;
  int32_t ep_len;
  uint32_t episode;
} lander_t;

long long orc_lunar_dbg_pos_iters = 0;   /* debug counter: position sweeps executed */
static float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
static float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
static float clampf(float x, float lo, float hi) { return fmaxf(lo, fminf(x, hi)); }
static float leg_sign(int L) { return L == 0 ? -1.0f : 1.0f; }
static float joint_lower(int L) { return L == 0 ? 0.4f : -0.9f; }
static float joint_upper(int L) { return L == 0 ? 0.9f : -0.4f; }

static void body_xf(const body_t* b, float lcy, float* px, float* py, float* qs, float* qc) {
  orc_sincosf(b->a, qs, qc);
  *px = b->cx - (*qc * 0.0f - *qs * lcy);
  *py = b->cy - (*qs * 0.0f + *qc * lcy);
}

/* b2CollideEdgeAndPolygon (b2EPCollider::Collide), edge without ghost vertices, edge
 * frame == world.  Polygon given in body frame + transform. */
static void collide_edge_polygon(manifold_t* mf, int NV, const float* VX, const float* VY,
                                 const float* NX, const float* NY, float px, float py, float qs,
                                 float qc, float ccx, float ccy, float v1x, float v1y, float v2x,
                                 float v2y) {
  float wx[6], wy[6], wnx[6], wny[6];
  mf->count = 0;
  for (int i = 0; i < NV; ++i) {
    wx[i] = (qc * VX[i] - qs * VY[i]) + px;
    wy[i] = (qs * VX[i] + qc * VY[i]) + py;
    wnx[i] = qc * NX[i] - qs * NY[i];
    wny[i] = qs * NX[i] + qc * NY[i];
  }
  float ex = v2x - v1x, ey = v2y - v1y;
  float el = sqrtf(ex * ex + ey * ey);
  float inv = 1.0f / el;
  ex *= inv; ey *= inv;
  float n1x = ey, n1y = -ex;
  float offset1 = dot2(n1x, n1y, ccx - v1x, ccy - v1y);
  int front = offset1 >= 0.0f;
  float Nx = front ? n1x : -n1x, Ny = front ? n1y : -n1y;
  float radius = 2.0f * POLY_RADIUS;

  float esep = 3.4e38f;                                       /* ComputeEdgeSeparation */
  for (int i = 0; i < NV; ++i) {
    float s = dot2(Nx, Ny, wx[i] - v1x, wy[i] - v1y);
    if (s < esep) esep = s;
  }
  if (esep > radius) return;
  float psep = -3.4e38f; int pidx = -1;                       /* ComputePolygonSeparation */
  for (int i = 0; i < NV; ++i) {
    float nx = -wnx[i], ny = -wny[i];
    float s1 = dot2(nx, ny, wx[i] - v1x, wy[i] - v1y);
    float s2 = dot2(nx, ny, wx[i] - v2x, wy[i] - v2y);
    float s = fminf(s1, s2);
    if (s > radius) return;
    if (s > psep) { psep = s; pidx = i; }
  }
  int use_edge = (pidx < 0) || !(psep > 0.98f * esep + 0.001f);

  float i0x, i0y, i1x, i1y, rv1x, rv1y, rv2x, rv2y, rnx, rny;
  uint32_t id0, id1; int ri1, ri2;
  if (use_edge) {
    int best = 0; float bestv = dot2(Nx, Ny, wnx[0], wny[0]);
    for (int i = 1; i < NV; ++i) {
      float v = dot2(Nx, Ny, wnx[i], wny[i]);
      if (v < bestv) { bestv = v; best = i; }
    }
    int b2 = best + 1 < NV ? best + 1 : 0;
    i0x = wx[best]; i0y = wy[best]; i1x = wx[b2]; i1y = wy[b2];
    id0 = 0u | ((uint32_t)best << 8) | (1u << 16) | (0u << 24);
    id1 = 0u | ((uint32_t)b2 << 8) | (1u << 16) | (0u << 24);
    if (front) { ri1 = 0; ri2 = 1; rv1x = v1x; rv1y = v1y; rv2x = v2x; rv2y = v2y; rnx = n1x; rny = n1y; }
    else { ri1 = 1; ri2 = 0; rv1x = v2x; rv1y = v2y; rv2x = v1x; rv2y = v1y; rnx = -n1x; rny = -n1y; }
  } else {
    i0x = v1x; i0y = v1y; i1x = v2x; i1y = v2y;
    id0 = 0u | ((uint32_t)pidx << 8) | (0u << 16) | (1u << 24);
    id1 = id0;
    ri1 = pidx; ri2 = pidx + 1 < NV ? pidx + 1 : 0;
    rv1x = wx[ri1]; rv1y = wy[ri1]; rv2x = wx[ri2]; rv2y = wy[ri2]; rnx = wnx[ri1]; rny = wny[ri1];
  }
  float sn1x = rny, sn1y = -rnx;
  float so1 = dot2(sn1x, sn1y, rv1x, rv1y);
  float so2 = dot2(-sn1x, -sn1y, rv2x, rv2y);

  /* b2ClipSegmentToLine, twice */
  float cx[2] = {0, 0}, cy[2] = {0, 0}; uint32_t cid[2] = {0, 0}; int np = 0;
  {
    float d0 = dot2(sn1x, sn1y, i0x, i0y) - so1, d1 = dot2(sn1x, sn1y, i1x, i1y) - so1;
    if (d0 <= 0.0f) { cx[np] = i0x; cy[np] = i0y; cid[np] = id0; ++np; }
    if (d1 <= 0.0f) { cx[np] = i1x; cy[np] = i1y; cid[np] = id1; ++np; }
    if (d0 * d1 < 0.0f) {
      float t = d0 / (d0 - d1);
      cx[np] = i0x + t * (i1x - i0x); cy[np] = i0y + t * (i1y - i0y);
      cid[np] = (uint32_t)ri1 | (((id0 >> 8) & 0xFFu) << 8) | (0u << 16) | (1u << 24);
      ++np;
    }
  }
  if (np < 2) return;
  float fx[2] = {0, 0}, fy[2] = {0, 0}; uint32_t fid[2] = {0, 0}; np = 0;
  {
    float d0 = dot2(-sn1x, -sn1y, cx[0], cy[0]) - so2, d1 = dot2(-sn1x, -sn1y, cx[1], cy[1]) - so2;
    if (d0 <= 0.0f) { fx[np] = cx[0]; fy[np] = cy[0]; fid[np] = cid[0]; ++np; }
    if (d1 <= 0.0f) { fx[np] = cx[1]; fy[np] = cy[1]; fid[np] = cid[1]; ++np; }
    if (d0 * d1 < 0.0f) {
      float t = d0 / (d0 - d1);
      fx[np] = cx[0] + t * (cx[1] - cx[0]); fy[np] = cy[0] + t * (cy[1] - cy[0]);
      fid[np] = (uint32_t)ri2 | (((cid[0] >> 8) & 0xFFu) << 8) | (0u << 16) | (1u << 24);
      ++np;
    }
  }
  if (np < 2) return;

  mf->faceB = use_edge ? 0 : 1;
  if (use_edge) { mf->lnx = rnx; mf->lny = rny; mf->lpx = rv1x; mf->lpy = rv1y; }
  else { mf->lnx = NX[ri1]; mf->lny = NY[ri1]; mf->lpx = VX[ri1]; mf->lpy = VY[ri1]; }
  int cnt = 0;
  for (int k = 0; k < 2; ++k) {
    float sep = dot2(rnx, rny, fx[k] - rv1x, fy[k] - rv1y);
    if (sep <= radius) {
      cpoint_t* cp = &mf->p[cnt];
      if (use_edge) {
        float dx = fx[k] - px, dy = fy[k] - py;
        cp->lpx = qc * dx + qs * dy; cp->lpy = -qs * dx + qc * dy;
        cp->key = fid[k];
      } else {
        uint32_t q = fid[k];
        cp->lpx = fx[k]; cp->lpy = fy[k];
        cp->key = ((q >> 8) & 0xFFu) | ((q & 0xFFu) << 8) | (((q >> 24) & 0xFFu) << 16) | (((q >> 16) & 0xFFu) << 24);
      }
      ++cnt;
    }
  }
  mf->count = cnt;
}

static void poly_of(int b, int* nv, const float** vx, const float** vy, const float** nx, const float** ny) {
  if (b == 0) { *nv = 6; *vx = HULL_VX; *vy = HULL_VY; *nx = HULL_NX; *ny = HULL_NY; }
  else { *nv = 4; *vx = LEG_VX; *vy = LEG_VY; *nx = LEG_NX; *ny = LEG_NY; }
}

/* One world.Step(1/50, 180, 60) preceded by the engine impulses of gymnasium's step(). */
static void world_step(lander_t* W, int action, float disp0, float disp1, float fx, float fy,
                       float* m_power, float* s_power) {
  body_t* B = W->b;
  *m_power = 0.0f; *s_power = 0.0f;
  {
    float sn, cs;
    orc_sincosf(B[0].a, &sn, &cs);
    float tipx = sn, tipy = cs, sidex = -cs, sidey = sn;
    float posx = B[0].cx - (cs * 0.0f - sn * HULL_LCY);
    float posy = B[0].cy - (sn * 0.0f + cs * HULL_LCY);
    if (action == 2) {                                    /* main engine */
      *m_power = 1.0f;
      float ox = tipx * (4.0f / SCALE + 2.0f * disp0) + sidex * disp1;
      float oy = -tipy * (4.0f / SCALE + 2.0f * disp0) - sidey * disp1;
      float ipx = posx + ox, ipy = posy + oy;
      float Ix = -ox * 13.0f * *m_power, Iy = -oy * 13.0f * *m_power;
      B[0].vx += INV_M[0] * Ix; B[0].vy += INV_M[0] * Iy;
      B[0].w += INV_I[0] * cross2(ipx - B[0].cx, ipy - B[0].cy, Ix, Iy);
    }
    if (action == 1 || action == 3) {                     /* orientation engines */
      float dir = (float)(action - 2);
      *s_power = 1.0f;
      float ox = tipx * disp0 + sidex * (3.0f * disp1 + dir * 12.0f / SCALE);
      float oy = -tipy * disp0 - sidey * (3.0f * disp1 + dir * 12.0f / SCALE);
      float ipx = posx + ox - tipx * 17.0f / SCALE;
      float ipy = posy + oy + tipy * 14.0f / SCALE;
      float Ix = -ox * 0.6f * *s_power, Iy = -oy * 0.6f * *s_power;
      B[0].vx += INV_M[0] * Ix; B[0].vy += INV_M[0] * Iy;
      B[0].w += INV_I[0] * cross2(ipx - B[0].cx, ipy - B[0].cy, Ix, Iy);
    }
  }

  /* ---- b2ContactManager::Collide ---- */
  uint32_t touching_now = 0u;
  for (int b = 0; b < 3; ++b) {
    int nv; const float *VX, *VY, *NX, *NY;
    poly_of(b, &nv, &VX, &VY, &NX, &NY);
    float px, py, qs, qc;
    body_xf(&B[b], b == 0 ? HULL_LCY : 0.0f, &px, &py, &qs, &qc);
    float minx = 3.4e38f, maxx = -3.4e38f, miny = 3.4e38f;
    for (int i = 0; i < nv; ++i) {
      float x = (qc * VX[i] - qs * VY[i]) + px, y = (qs * VX[i] + qc * VY[i]) + py;
      minx = fminf(minx, x); maxx = fmaxf(maxx, x); miny = fminf(miny, y);
    }
    int e_lo = (int)floorf((minx - 2.0f * POLY_RADIUS) * 0.5f);
    int e_hi = (int)floorf((maxx + 2.0f * POLY_RADIUS) * 0.5f);
    if (e_lo < 0) e_lo = 0;
    if (e_hi > 9) e_hi = 9;
    manifold_t old[2] = {W->m[b][0], W->m[b][1]};
    int old_e0 = W->edge0[b];
    W->edge0[b] = e_lo;
    for (int s = 0; s < 2; ++s) {
      manifold_t* mf = &W->m[b][s];
      memset(mf, 0, sizeof(*mf));
      int e = e_lo + s;
      if (e <= e_hi && e >= 0 && e <= 9) {
        float y1 = W->ty[e], y2 = W->ty[e + 1];
        float x1 = 2.0f * (float)e, x2 = 2.0f * (float)(e + 1);
        if (miny - 2.0f * POLY_RADIUS <= fmaxf(y1, y2)) {
          float ccx = b == 0 ? B[0].cx : px, ccy = b == 0 ? B[0].cy : py;
          collide_edge_polygon(mf, nv, VX, VY, NX, NY, px, py, qs, qc, ccx, ccy, x1, y1, x2, y2);
        }
        for (int k = 0; k < 2; ++k) {                     /* b2Contact::Update: match feature ids */
          mf->p[k].ni = 0.0f; mf->p[k].ti = 0.0f;
          if (k < mf->count) {
            int found = 0;
            for (int os = 0; os < 2 && !found; ++os) {
              if (old_e0 + os != e) continue;
              for (int q = 0; q < old[os].count && !found; ++q)
                if (old[os].p[q].key == mf->p[k].key) { mf->p[k].ni = old[os].p[q].ni; mf->p[k].ti = old[os].p[q].ti; found = 1; }
            }
          }
        }
        if (mf->count > 0) touching_now |= 1u << (10 * b + e);
      }
    }
  }
  {                                                        /* ContactDetector */
    uint32_t began = touching_now & ~W->touching, ended = W->touching & ~touching_now;
    if (began & 0x3FFu) W->flags |= 4u;
    for (int L = 0; L < 2; ++L)
      for (int e = 0; e < 10; ++e) {
        uint32_t bit = 1u << (10 * (L + 1) + e);
        if (began & bit) W->flags |= (1u << L);
        if (ended & bit) W->flags &= ~(1u << L);
      }
    W->touching = touching_now;
  }

  /* ---- b2Island::Solve ---- */
  const float h = DT;
  for (int b = 0; b < 3; ++b) {
    float Fx = b == 0 ? fx : 0.0f, Fy = b == 0 ? fy : 0.0f;
    B[b].vx += h * (0.0f + INV_M[b] * Fx);
    B[b].vy += h * (-10.0f + INV_M[b] * Fy);
  }
  vc_t vc[3][2];
  memset(vc, 0, sizeof(vc));
  for (int b = 0; b < 3; ++b) {                           /* InitializeVelocityConstraints + WarmStart */
    float px, py, qs, qc;
    body_xf(&B[b], b == 0 ? HULL_LCY : 0.0f, &px, &py, &qs, &qc);
    float mB = INV_M[b], iB = INV_I[b];
    for (int s = 0; s < 2; ++s) {
      manifold_t* mf = &W->m[b][s];
      vc_t* c = &vc[b][s];
      c->count = mf->count;
      if (mf->count == 0) continue;
      float nx, ny, wpx[2], wpy[2];
      if (!mf->faceB) {                                   /* b2WorldManifold e_faceA */
        nx = mf->lnx; ny = mf->lny;
        for (int k = 0; k < 2; ++k) {
          float clx = (qc * mf->p[k].lpx - qs * mf->p[k].lpy) + px;
          float cly = (qs * mf->p[k].lpx + qc * mf->p[k].lpy) + py;
          float d = POLY_RADIUS - dot2(clx - mf->lpx, cly - mf->lpy, nx, ny);
          float cAx = clx + d * nx, cAy = cly + d * ny;
          float cBx = clx - POLY_RADIUS * nx, cBy = cly - POLY_RADIUS * ny;
          wpx[k] = 0.5f * (cAx + cBx); wpy[k] = 0.5f * (cAy + cBy);
        }
      } else {                                            /* e_faceB */
        nx = qc * mf->lnx - qs * mf->lny; ny = qs * mf->lnx + qc * mf->lny;
        float ppx = (qc * mf->lpx - qs * mf->lpy) + px, ppy = (qs * mf->lpx + qc * mf->lpy) + py;
        for (int k = 0; k < 2; ++k) {
          float clx = mf->p[k].lpx, cly = mf->p[k].lpy;
          float d = POLY_RADIUS - dot2(clx - ppx, cly - ppy, nx, ny);
          float cBx = clx + d * nx, cBy = cly + d * ny;
          float cAx = clx - POLY_RADIUS * nx, cAy = cly - POLY_RADIUS * ny;
          wpx[k] = 0.5f * (cAx + cBx); wpy[k] = 0.5f * (cAy + cBy);
        }
        nx = -nx; ny = -ny;
      }
      c->nx = nx; c->ny = ny;
      float tx = ny, ty = -nx;
      for (int k = 0; k < 2; ++k) {
        c->rx[k] = wpx[k] - B[b].cx; c->ry[k] = wpy[k] - B[b].cy;
        float rn = cross2(c->rx[k], c->ry[k], nx, ny);
        float kn = mB + iB * rn * rn;
        c->nmass[k] = kn > 0.0f ? 1.0f / kn : 0.0f;
        float rt = cross2(c->rx[k], c->ry[k], tx, ty);
        float kt = mB + iB * rt * rt;
        c->tmass[k] = kt > 0.0f ? 1.0f / kt : 0.0f;
      }
      if (mf->count == 2) {                               /* block solver preparation */
        float rn1 = cross2(c->rx[0], c->ry[0], nx, ny), rn2 = cross2(c->rx[1], c->ry[1], nx, ny);
        float k11 = mB + iB * rn1 * rn1, k22 = mB + iB * rn2 * rn2, k12 = mB + iB * rn1 * rn2;
        if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
          float det = k11 * k22 - k12 * k12;
          if (det != 0.0f) det = 1.0f / det;
          c->k11 = k11; c->k12 = k12; c->k22 = k22;
          c->i11 = det * k22; c->i12 = -det * k12; c->i22 = det * k11;
        } else {
          c->count = 1;
        }
      }
      for (int k = 0; k < c->count; ++k) {
        float Px = mf->p[k].ni * nx + mf->p[k].ti * tx, Py = mf->p[k].ni * ny + mf->p[k].ti * ty;
        B[b].w += iB * cross2(c->rx[k], c->ry[k], Px, Py);
        B[b].vx += mB * Px; B[b].vy += mB * Py;
      }
    }
  }
  /* b2RevoluteJoint::InitVelocityConstraints, island joint order: legs[1] then legs[0] */
  float jrAx[2], jrAy[2], jrBx[2], jrBy[2], K11[2], K12[2], K13[2], K22[2], K23[2], K33[2], mmass[2];
  {
    float qsA, qcA;
    orc_sincosf(B[0].a, &qsA, &qcA);
    for (int jj = 0; jj < 2; ++jj) {
      int L = 1 - jj, bi = L + 1;
      float qsB, qcB;
      orc_sincosf(B[bi].a, &qsB, &qcB);
      float lax = 0.0f - 0.0f, lay = 0.0f - HULL_LCY;
      float lbx = leg_sign(L) * LEG_AWAY, lby = LEG_DOWN;
      jrAx[L] = qcA * lax - qsA * lay; jrAy[L] = qsA * lax + qcA * lay;
      jrBx[L] = qcB * lbx - qsB * lby; jrBy[L] = qsB * lbx + qcB * lby;
      float mA = INV_M[0], mB = INV_M[bi], iA = INV_I[0], iB = INV_I[bi];
      K11[L] = mA + mB + jrAy[L] * jrAy[L] * iA + jrBy[L] * jrBy[L] * iB;
      K12[L] = -jrAy[L] * jrAx[L] * iA - jrBy[L] * jrBx[L] * iB;
      K13[L] = -jrAy[L] * iA - jrBy[L] * iB;
      K22[L] = mA + mB + jrAx[L] * jrAx[L] * iA + jrBx[L] * jrBx[L] * iB;
      K23[L] = jrAx[L] * iA + jrBx[L] * iB;
      K33[L] = iA + iB;
      mmass[L] = 1.0f / (iA + iB);
      joint_t* J = &W->j[L];
      float ang = B[bi].a - B[0].a;
      if (ang <= joint_lower(L)) { if (J->state != 1) J->iz = 0.0f; J->state = 1; }
      else if (ang >= joint_upper(L)) { if (J->state != 2) J->iz = 0.0f; J->state = 2; }
      else { J->state = 0; J->iz = 0.0f; }
      float Px = J->ix, Py = J->iy;
      B[0].vx -= mA * Px; B[0].vy -= mA * Py;
      B[0].w -= iA * (cross2(jrAx[L], jrAy[L], Px, Py) + J->im + J->iz);
      B[bi].vx += mB * Px; B[bi].vy += mB * Py;
      B[bi].w += iB * (cross2(jrBx[L], jrBy[L], Px, Py) + J->im + J->iz);
    }
  }

  static const int ORDER[3] = {2, 0, 1};
  for (int it = 0; it < VEL_ITERS; ++it) {
    for (int jj = 0; jj < 2; ++jj) {                      /* b2RevoluteJoint::SolveVelocityConstraints */
      int L = 1 - jj, bi = L + 1;
      joint_t* J = &W->j[L];
      float mA = INV_M[0], mB = INV_M[bi], iA = INV_I[0], iB = INV_I[bi];
      {
        float Cdot = B[bi].w - B[0].w - 0.3f * leg_sign(L);
        float imp = -mmass[L] * Cdot;
        float old = J->im, maxImp = h * MOTOR_TORQUE;
        J->im = clampf(old + imp, -maxImp, maxImp);
        imp = J->im - old;
        B[0].w -= iA * imp; B[bi].w += iB * imp;
      }
      float Cx = B[bi].vx + (-B[bi].w * jrBy[L]) - B[0].vx - (-B[0].w * jrAy[L]);
      float Cy = B[bi].vy + (B[bi].w * jrBx[L]) - B[0].vy - (B[0].w * jrAx[L]);
      float ipx, ipy, ipz;
      if (J->state != 0) {
        float Cz = B[bi].w - B[0].w;
        float a11 = K11[L], a12 = K12[L], a13 = K13[L], a22 = K22[L], a23 = K23[L], a33 = K33[L];
        float c1x = a22 * a33 - a23 * a23, c1y = a23 * a13 - a12 * a33, c1z = a12 * a23 - a22 * a13;
        float det = a11 * c1x + a12 * c1y + a13 * c1z;
        if (det != 0.0f) det = 1.0f / det;
        float bx = Cx, by = Cy, bz = Cz;
        float sx = det * (bx * c1x + by * c1y + bz * c1z);
        float c2x = by * a33 - bz * a23, c2y = bz * a13 - bx * a33, c2z = bx * a23 - by * a13;
        float sy = det * (a11 * c2x + a12 * c2y + a13 * c2z);
        float c3x = a22 * bz - a23 * by, c3y = a23 * bx - a12 * bz, c3z = a12 * by - a22 * bx;
        float sz = det * (a11 * c3x + a12 * c3y + a13 * c3z);
        ipx = -sx; ipy = -sy; ipz = -sz;
        float newImp = J->iz + ipz;
        int release = (J->state == 1) ? (newImp < 0.0f) : (newImp > 0.0f);
        if (release) {
          float rx = -Cx + J->iz * a13, ry = -Cy + J->iz * a23;
          float d2 = a11 * a22 - a12 * a12;
          if (d2 != 0.0f) d2 = 1.0f / d2;
          float ux = d2 * (a22 * rx - a12 * ry), uy = d2 * (a11 * ry - a12 * rx);
          ipx = ux; ipy = uy; ipz = -J->iz;
          J->ix += ux; J->iy += uy; J->iz = 0.0f;
        } else {
          J->ix += ipx; J->iy += ipy; J->iz += ipz;
        }
      } else {
        float a11 = K11[L], a12 = K12[L], a22 = K22[L];
        float d2 = a11 * a22 - a12 * a12;
        if (d2 != 0.0f) d2 = 1.0f / d2;
        float rx = -Cx, ry = -Cy;
        ipx = d2 * (a22 * rx - a12 * ry); ipy = d2 * (a11 * ry - a12 * rx); ipz = 0.0f;
        J->ix += ipx; J->iy += ipy;
      }
      B[0].vx -= mA * ipx; B[0].vy -= mA * ipy;
      B[0].w -= iA * (cross2(jrAx[L], jrAy[L], ipx, ipy) + ipz);
      B[bi].vx += mB * ipx; B[bi].vy += mB * ipy;
      B[bi].w += iB * (cross2(jrBx[L], jrBy[L], ipx, ipy) + ipz);
    }
    for (int ob = 0; ob < 3; ++ob) {                      /* b2ContactSolver::SolveVelocityConstraints */
      int b = ORDER[ob];
      float mB = INV_M[b], iB = INV_I[b];
      float fr = b == 0 ? sqrtf(0.1f * 0.1f) : sqrtf(0.2f * 0.1f);   /* b2MixFriction */
      for (int s = 0; s < 2; ++s) {
        vc_t* c = &vc[b][s];
        if (c->count == 0) continue;
        manifold_t* mf = &W->m[b][s];
        float nx = c->nx, ny = c->ny, tx = ny, ty = -nx;
        for (int k = 0; k < c->count; ++k) {
          float dvx = B[b].vx + (-B[b].w * c->ry[k]), dvy = B[b].vy + (B[b].w * c->rx[k]);
          float vt = dot2(dvx, dvy, tx, ty);
          float lam = c->tmass[k] * (-vt);
          float maxF = fr * mf->p[k].ni;
          float nw = clampf(mf->p[k].ti + lam, -maxF, maxF);
          lam = nw - mf->p[k].ti;
          mf->p[k].ti = nw;
          float Px = lam * tx, Py = lam * ty;
          B[b].vx += mB * Px; B[b].vy += mB * Py;
          B[b].w += iB * cross2(c->rx[k], c->ry[k], Px, Py);
        }
        if (c->count == 1) {
          float dvx = B[b].vx + (-B[b].w * c->ry[0]), dvy = B[b].vy + (B[b].w * c->rx[0]);
          float vn = dot2(dvx, dvy, nx, ny);
          float lam = -c->nmass[0] * vn;
          float nw = fmaxf(mf->p[0].ni + lam, 0.0f);
          lam = nw - mf->p[0].ni;
          mf->p[0].ni = nw;
          float Px = lam * nx, Py = lam * ny;
          B[b].vx += mB * Px; B[b].vy += mB * Py;
          B[b].w += iB * cross2(c->rx[0], c->ry[0], Px, Py);
        } else {
          float a1 = mf->p[0].ni, a2 = mf->p[1].ni;
          float dv1x = B[b].vx + (-B[b].w * c->ry[0]), dv1y = B[b].vy + (B[b].w * c->rx[0]);
          float dv2x = B[b].vx + (-B[b].w * c->ry[1]), dv2y = B[b].vy + (B[b].w * c->rx[1]);
          float vn1 = dot2(dv1x, dv1y, nx, ny), vn2 = dot2(dv2x, dv2y, nx, ny);
          float b1 = vn1 - (c->k11 * a1 + c->k12 * a2);
          float b2 = vn2 - (c->k12 * a1 + c->k22 * a2);
          float x1 = -(c->i11 * b1 + c->i12 * b2), x2 = -(c->i12 * b1 + c->i22 * b2);
          int ok = (x1 >= 0.0f && x2 >= 0.0f);
          if (!ok) { x1 = -c->nmass[0] * b1; x2 = 0.0f; vn2 = c->k12 * x1 + b2; ok = (x1 >= 0.0f && vn2 >= 0.0f); }
          if (!ok) { x1 = 0.0f; x2 = -c->nmass[1] * b2; vn1 = c->k12 * x2 + b1; ok = (x2 >= 0.0f && vn1 >= 0.0f); }
          if (!ok) { x1 = 0.0f; x2 = 0.0f; ok = (b1 >= 0.0f && b2 >= 0.0f); }
          if (ok) {
            float d1 = x1 - a1, d2 = x2 - a2;
            float P1x = d1 * nx, P1y = d1 * ny, P2x = d2 * nx, P2y = d2 * ny;
            B[b].vx += mB * (P1x + P2x); B[b].vy += mB * (P1y + P2y);
            B[b].w += iB * (cross2(c->rx[0], c->ry[0], P1x, P1y) + cross2(c->rx[1], c->ry[1], P2x, P2y));
            mf->p[0].ni = x1; mf->p[1].ni = x2;
          }
        }
      }
    }
  }

  for (int b = 0; b < 3; ++b) {                           /* integrate positions */
    float tx = h * B[b].vx, ty = h * B[b].vy;
    if (dot2(tx, ty, tx, ty) > MAX_TRANSLATION * MAX_TRANSLATION) {
      float ratio = MAX_TRANSLATION / sqrtf(dot2(tx, ty, tx, ty));
      B[b].vx *= ratio; B[b].vy *= ratio;
    }
    float rot = h * B[b].w;
    if (rot * rot > MAX_ROTATION * MAX_ROTATION) {
      float ratio = MAX_ROTATION / fabsf(rot);
      B[b].w *= ratio;
    }
    B[b].cx += h * B[b].vx; B[b].cy += h * B[b].vy; B[b].a += h * B[b].w;
  }

  int position_solved = 0;
  for (int it = 0; it < POS_ITERS; ++it) {
    orc_lunar_dbg_pos_iters++;
    float min_sep = 0.0f;
    for (int ob = 0; ob < 3; ++ob) {                      /* b2ContactSolver::SolvePositionConstraints */
      int b = ORDER[ob];
      float mB = INV_M[b], iB = INV_I[b], lcy = b == 0 ? HULL_LCY : 0.0f;
      for (int s = 0; s < 2; ++s) {
        manifold_t* mf = &W->m[b][s];
        for (int k = 0; k < mf->count; ++k) {
          float px, py, qs, qc;
          body_xf(&B[b], lcy, &px, &py, &qs, &qc);
          float nx, ny, ptx, pty, sep;
          float qx = mf->p[k].lpx, qy = mf->p[k].lpy;
          if (!mf->faceB) {
            nx = mf->lnx; ny = mf->lny;
            float clx = (qc * qx - qs * qy) + px, cly = (qs * qx + qc * qy) + py;
            sep = dot2(clx - mf->lpx, cly - mf->lpy, nx, ny) - POLY_RADIUS - POLY_RADIUS;
            ptx = clx; pty = cly;
          } else {
            nx = qc * mf->lnx - qs * mf->lny; ny = qs * mf->lnx + qc * mf->lny;
            float ppx = (qc * mf->lpx - qs * mf->lpy) + px, ppy = (qs * mf->lpx + qc * mf->lpy) + py;
            sep = dot2(qx - ppx, qy - ppy, nx, ny) - POLY_RADIUS - POLY_RADIUS;
            ptx = qx; pty = qy;
            nx = -nx; ny = -ny;
          }
          float rx = ptx - B[b].cx, ry = pty - B[b].cy;
          min_sep = fminf(min_sep, sep);
          float C = clampf(BAUMGARTE * (sep + LINEAR_SLOP), -MAX_LIN_CORR, 0.0f);
          float rn = cross2(rx, ry, nx, ny);
          float K = mB + iB * rn * rn;
          float imp = K > 0.0f ? -C / K : 0.0f;
          float Px = imp * nx, Py = imp * ny;
          B[b].cx += mB * Px; B[b].cy += mB * Py;
          B[b].a += iB * cross2(rx, ry, Px, Py);
        }
      }
    }
    int contacts_ok = min_sep >= -3.0f * LINEAR_SLOP;
    int joints_ok = 1;
    for (int jj = 0; jj < 2; ++jj) {                      /* b2RevoluteJoint::SolvePositionConstraints */
      int L = 1 - jj, bi = L + 1;
      joint_t* J = &W->j[L];
      float mA = INV_M[0], mB = INV_M[bi], iA = INV_I[0], iB = INV_I[bi];
      float angErr = 0.0f;
      if (J->state != 0) {
        float ang = B[bi].a - B[0].a;
        float limImp = 0.0f;
        if (J->state == 1) {
          float C = ang - joint_lower(L);
          angErr = -C;
          C = clampf(C + ANGULAR_SLOP, -MAX_ANG_CORR, 0.0f);
          limImp = -mmass[L] * C;
        } else {
          float C = ang - joint_upper(L);
          angErr = C;
          C = clampf(C - ANGULAR_SLOP, 0.0f, MAX_ANG_CORR);
          limImp = -mmass[L] * C;
        }
        B[0].a -= iA * limImp; B[bi].a += iB * limImp;
      }
      float qsA, qcA, qsB, qcB;
      orc_sincosf(B[0].a, &qsA, &qcA);
      orc_sincosf(B[bi].a, &qsB, &qcB);
      float lax = 0.0f, lay = 0.0f - HULL_LCY;
      float lbx = leg_sign(L) * LEG_AWAY, lby = LEG_DOWN;
      float rAx = qcA * lax - qsA * lay, rAy = qsA * lax + qcA * lay;
      float rBx = qcB * lbx - qsB * lby, rBy = qsB * lbx + qcB * lby;
      float Cx = B[bi].cx + rBx - B[0].cx - rAx, Cy = B[bi].cy + rBy - B[0].cy - rAy;
      float posErr = sqrtf(Cx * Cx + Cy * Cy);
      float k11 = mA + mB + iA * rAy * rAy + iB * rBy * rBy;
      float k12 = -iA * rAx * rAy - iB * rBx * rBy;
      float k22 = mA + mB + iA * rAx * rAx + iB * rBx * rBx;
      float det = k11 * k22 - k12 * k12;
      if (det != 0.0f) det = 1.0f / det;
      float impx = -(det * (k22 * Cx - k12 * Cy)), impy = -(det * (k11 * Cy - k12 * Cx));
      B[0].cx -= mA * impx; B[0].cy -= mA * impy;
      B[0].a -= iA * cross2(rAx, rAy, impx, impy);
      B[bi].cx += mB * impx; B[bi].cy += mB * impy;
      B[bi].a += iB * cross2(rBx, rBy, impx, impy);
      joints_ok = joints_ok && (posErr <= LINEAR_SLOP) && (angErr <= ANGULAR_SLOP);
    }
    if (contacts_ok && joints_ok) { position_solved = 1; break; }
  }

  float min_sleep = 3.4e38f;                              /* island sleep bookkeeping */
  for (int b = 0; b < 3; ++b) {
    if (B[b].w * B[b].w > ANG_SLEEP_TOL * ANG_SLEEP_TOL ||
        dot2(B[b].vx, B[b].vy, B[b].vx, B[b].vy) > LIN_SLEEP_TOL * LIN_SLEEP_TOL) {
      W->sleep[b] = 0.0f; min_sleep = 0.0f;
    } else {
      W->sleep[b] += h;
      min_sleep = fminf(min_sleep, W->sleep[b]);
    }
  }
  if (min_sleep >= TIME_TO_SLEEP && position_solved) W->flags |= 16u;
}

/* ------------------------------------------------------------ gymnasium layer */
static void lander_obs(const lander_t* W, float* o) {
  float sn, cs;
  orc_sincosf(W->b[0].a, &sn, &cs);
  float posx = W->b[0].cx - (cs * 0.0f - sn * HULL_LCY);
  float posy = W->b[0].cy - (sn * 0.0f + cs * HULL_LCY);
  o[0] = (posx - WV / 2.0f) / (WV / 2.0f);
  o[1] = (posy - (HELIPAD_Y + LEG_DOWN)) / (HV / 2.0f);
  o[2] = W->b[0].vx * (WV / 2.0f) / 50.0f;
  o[3] = W->b[0].vy * (HV / 2.0f) / 50.0f;
  o[4] = W->b[0].a;
  o[5] = 20.0f * W->b[0].w / 50.0f;
  o[6] = (W->flags & 1u) ? 1.0f : 0.0f;
  o[7] = (W->flags & 2u) ? 1.0f : 0.0f;
}

static void env_step_once(lander_t* W, int action, uint64_t seed, uint64_t env, uint32_t episode,
                          uint32_t step_idx, float fx, float fy, float* o, float* reward,
                          int* terminated) {
  uint32_t r[4];
  orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_STEP | step_idx, r);
  float d0 = (-1.0f + 2.0f * u01f(r[0])) / SCALE, d1 = (-1.0f + 2.0f * u01f(r[1])) / SCALE;
  float mp, sp;
  world_step(W, action, d0, d1, fx, fy, &mp, &sp);
  lander_obs(W, o);
  float shaping = -100.0f * sqrtf(o[0] * o[0] + o[1] * o[1]) - 100.0f * sqrtf(o[2] * o[2] + o[3] * o[3]) -
                  100.0f * fabsf(o[4]) + 10.0f * o[6] + 10.0f * o[7];
  *reward = 0.0f;
  if (W->flags & 8u) *reward = shaping - W->prev_shaping;
  W->prev_shaping = shaping; W->flags |= 8u;
  *reward -= mp * 0.30f;
  *reward -= sp * 0.03f;
  *terminated = 0;
  if ((W->flags & 4u) || fabsf(o[0]) >= 1.0f) { *terminated = 1; *reward = -100.0f; }
  if (W->flags & 16u) { *terminated = 1; *reward = 100.0f; }
}

static void init_episode(lander_t* W, uint64_t seed, uint64_t env, uint32_t episode, float* fx, float* fy) {
  float hgt[12];
  for (int blk = 0; blk < 3; ++blk) {
    uint32_t r[4];
    orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | (uint32_t)blk, r);
    for (int k = 0; k < 4; ++k) hgt[4 * blk + k] = (HV / 2.0f) * u01f(r[k]);
  }
  for (int i = 3; i <= 7; ++i) hgt[i] = HELIPAD_Y;
  for (int i = 0; i < 11; ++i) {
    float prev = hgt[i == 0 ? 11 : i - 1];
    W->ty[i] = 0.33f * (prev + hgt[i] + hgt[i + 1]);
  }
  uint32_t rf[4];
  orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 3u, rf);
  *fx = -1000.0f + 2000.0f * u01f(rf[0]); *fy = -1000.0f + 2000.0f * u01f(rf[1]);
  memset(W->b, 0, sizeof(W->b)); memset(W->j, 0, sizeof(W->j)); memset(W->m, 0, sizeof(W->m));
  W->b[0].cx = WV / 2.0f; W->b[0].cy = HV + HULL_LCY;
  for (int L = 0; L < 2; ++L) {
    float i = leg_sign(L);
    W->b[L + 1].cx = WV / 2.0f - i * LEG_AWAY; W->b[L + 1].cy = HV; W->b[L + 1].a = i * 0.05f;
  }
  for (int b = 0; b < 3; ++b) { W->sleep[b] = 0.0f; W->edge0[b] = 0; }
  W->touching = 0u; W->flags = 0u; W->prev_shaping = 0.0f;
}

void* orc_lunar_alloc(int n) { return calloc((size_t)n, sizeof(lander_t)); }

void orc_lunar_reset_one(void* st, int i, uint64_t seed, uint64_t env, uint32_t episode, float* obs) {
  lander_t* W = &((lander_t*)st)[i];
  float fx, fy, rew; int term;
  init_episode(W, seed, env, episode, &fx, &fy);
  env_step_once(W, 0, seed, env, episode, 0u, fx, fy, obs, &rew, &term);   /* reset() ends with step(0) */
  W->ep_ret = 0.0; W->ep_len = 0; W->episode = episode;
}

void orc_lunar_step_one(void* st, int i, uint64_t seed, uint64_t env, int action, float* obs_next,
                        float* obs_term, float* rew, uint8_t* terminated, uint8_t* truncated,
                        int* done, double* ep_ret, int* ep_len) {
  lander_t* W = &((lander_t*)st)[i];
  float reward; int term;
  if (action < 0) action = 0;
  if (action > 3) action = 3;
  env_step_once(W, action, seed, env, W->episode, (uint32_t)W->ep_len, 0.0f, 0.0f, obs_term, &reward, &term);
  int len = W->ep_len + 1;
  int trunc = len >= MAX_STEPS;
  *done = term || trunc;
  *ep_ret = W->ep_ret + (double)reward;
  *ep_len = len;
  *rew = reward; *terminated = (uint8_t)term; *truncated = (uint8_t)trunc;
  if (*done) {
    orc_lunar_reset_one(st, i, seed, env, W->episode + 1u, obs_next);
  } else {
    memcpy(obs_next, obs_term, sizeof(float) * 8);
    W->ep_ret = *ep_ret; W->ep_len = len;
  }
}

/* debug/inspection: raw body state of env i (18 floats) */
void orc_lunar_get_bodies(void* st, int i, float* out18) { memcpy(out18, ((lander_t*)st)[i].b, sizeof(float) * 18); }

/* debug/inspection: the whole world of env i in the 144-word order of the HIP state (env_lunar_device.hpp world_io):
 * bodies 18 | sleep 3 | joints 10 | manifolds 3 x 2 x 16 | edge0 3 | touching | terrain 11 | flags | prev_shaping */
void orc_lunar_get_words(void* st, int i, uint32_t* out144) {
  const lander_t* W = &((lander_t*)st)[i];
  int k = 0;
#define PUTF(x) do { float f__ = (x); memcpy(&out144[k++], &f__, 4); } while (0)
#define PUTU(x) do { out144[k++] = (uint32_t)(x); } while (0)
  for (int b = 0; b < 3; ++b) { PUTF(W->b[b].cx); PUTF(W->b[b].cy); PUTF(W->b[b].a); PUTF(W->b[b].vx); PUTF(W->b[b].vy); PUTF(W->b[b].w); }
  for (int b = 0; b < 3; ++b) PUTF(W->sleep[b]);
  for (int L = 0; L < 2; ++L) { PUTF(W->j[L].ix); PUTF(W->j[L].iy); PUTF(W->j[L].iz); PUTF(W->j[L].im); PUTU(W->j[L].state); }
  for (int b = 0; b < 3; ++b)
    for (int s2 = 0; s2 < 2; ++s2) {
      const manifold_t* m = &W->m[b][s2];
      PUTU(m->count); PUTU(m->faceB); PUTF(m->lnx); PUTF(m->lny); PUTF(m->lpx); PUTF(m->lpy);
      for (int q = 0; q < 2; ++q) { PUTF(m->p[q].lpx); PUTF(m->p[q].lpy); PUTU(m->p[q].key); PUTF(m->p[q].ni); PUTF(m->p[q].ti); }
    }
  for (int b = 0; b < 3; ++b) PUTU(W->edge0[b]);
  PUTU(W->touching);
  for (int t = 0; t < 11; ++t) PUTF(W->ty[t]);
  PUTU(W->flags);
  PUTF(W->prev_shaping);
#undef PUTF
#undef PUTU
}

/* the mass constants the solver uses: 1/m and 1/I of hull, leg, leg (tests derive them from the shapes) */
void orc_lunar_constants(float* out6) {
  for (int b = 0; b < 3; ++b) { out6[b] = INV_M[b]; out6[3 + b] = INV_I[b]; }
}
