/*
 * tfluids_oracle.c -- CPU restatement of the FluidNet `tfluids` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity ORACLE for the MI355X HIP path in fluidnet_amd/csrc. It is a plain-C99,
 * from-scratch restatement of the reference's float CPU algorithm (reference paths are relative
 * to /root/reference/torch/tfluids). It is never linked into, imported by, or called from the
 * product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PINNING: every function here is checked bit-for-bit (tests/test_oracle.py) against the
 * reference's own sources compiled in this container (oracle/_ref, recipe in oracle/Makefile),
 * against golden vectors generated from that build (tests/golden/, generator
 * tests/golden/make_golden.py) and against the portable known-answer cases of
 * generic/CalcLineTraceTest.m and test_tfluids.lua:675-753. The two solvers exist only as CUDA in
 * the reference (generic/tfluids.cc:836-839 raises): ora_solveLinearSystemJacobi and
 * ora_solveLinearSystemPCG restate generic/tfluids.cu:1765-1921 / :864-1759 and are pinned bit for
 * bit to those very lines compiled for the host (oracle/ref_jacobi.cc, oracle/ref_pcg.cc). For PCG
 * the five cuSPARSE / cuBLAS primitives the reference calls are not in its tree: their published
 * algorithms are restated (oracle/ref_shim/cusparse_host.h); cuSPARSE's own summation order is
 * the one thing left unpinned there.
 *
 * Layout: every field is a contiguous fp32 tensor [B][C][Z][Y][X], x fastest
 * (third_party/grid.h:68-78). MAC component c of cell (i,j,k) lives on the cell's negative
 * c-face. Cell centres are at (i+.5, j+.5, k+.5). Flags are Manta bit flags stored as floats
 * (third_party/cell_type.h:22-33) and tested via (int)f & bit (grid.h:140-174).
 *
 * Arithmetic notes (needed for bit parity with the reference built -ffp-contract=off):
 *   - all arithmetic is float unless the reference mixes in an unsuffixed double literal; the
 *     MacCormack correction `strength * 0.5 * (old - bwd)` (third_party/tfluids.cc:231,693) is
 *     evaluated in double and rounded once when added to fwd;
 *   - float->int conversions truncate toward zero, exactly like static_cast<int32_t>.
 * Build with: gcc -std=c99 -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { F_FLUID = 1, F_OBSTACLE = 2, F_EMPTY = 4, F_INFLOW = 8, F_OUTFLOW = 16, F_OPEN = 32,
       F_STICK = 128 };

/* advection methods, generic/advect_type.h:20-27 + advect_type.cc:18-37 */
enum { M_EULER = 0, M_MACCORMACK = 1, M_EULER_OURS = 2, M_RK2_OURS = 3, M_RK3_OURS = 4,
       M_MACCORMACK_OURS = 5 };

typedef struct { float x, y, z; } v3;

/* one batch item's view: dims + plane strides */
typedef struct {
  int X, Y, Z, is3d;
  long sy, sz, sc; /* element strides for y, z, channel */
} dom_t;

static dom_t mkdom(int Z, int Y, int X, int is3d) {
  dom_t d;
  d.X = X; d.Y = Y; d.Z = Z; d.is3d = is3d;
  d.sy = X; d.sz = (long)X * Y; d.sc = (long)X * Y * Z;
  return d;
}
#define AT(d, i, j, k) ((long)(i) + (long)(j) * (d)->sy + (long)(k) * (d)->sz)
#define ATC(d, i, j, k, c) (AT(d, i, j, k) + (long)(c) * (d)->sc)

static int flag_at(const dom_t* d, const float* f, int i, int j, int k) {
  return (int)f[AT(d, i, j, k)];
}
static int is_fluid(const dom_t* d, const float* f, int i, int j, int k) {
  return (flag_at(d, f, i, j, k) & F_FLUID) != 0;
}
static int is_obst(const dom_t* d, const float* f, int i, int j, int k) {
  return (flag_at(d, f, i, j, k) & F_OBSTACLE) != 0;
}
static int is_empty(const dom_t* d, const float* f, int i, int j, int k) {
  return (flag_at(d, f, i, j, k) & F_EMPTY) != 0;
}
static int is_outflow(const dom_t* d, const float* f, int i, int j, int k) {
  return (flag_at(d, f, i, j, k) & F_OUTFLOW) != 0;
}
static int is_stick(const dom_t* d, const float* f, int i, int j, int k) {
  return (flag_at(d, f, i, j, k) & F_STICK) != 0;
}
/* "border" = the one-cell shell; every op hard-codes bnd = 1 (third_party/tfluids.cc:467). */
static int on_border(const dom_t* d, int i, int j, int k) {
  return i < 1 || i > d->X - 2 || j < 1 || j > d->Y - 2 || (d->is3d && (k < 1 || k > d->Z - 2));
}

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static int iclamp(int v, int lo, int hi) { return imax(imin(v, hi), lo); } /* init.cu:33-35 */
/* std::min<real>(hi, std::max<real>(lo, v)), third_party/tfluids.cc:246-248 */
static float fclamp(float v, float lo, float hi) {
  float m = (lo < v) ? v : lo;
  return (m < hi) ? m : hi;
}

/* vec3::norm with its threshold, generic/vec3.h:119-127 (float kEpsilon = 1e-6f) */
static float v3norm(v3 a) {
  float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
  return (l2 > 1e-6f) ? sqrtf(l2) : 0.0f;
}
static v3 v3normalize(v3 a) { /* generic/vec3.h:129-141 */
  float n = v3norm(a);
  v3 r = {0.0f, 0.0f, 0.0f};
  if (n > 1e-6f) { r.x = a.x / n; r.y = a.y / n; r.z = a.z / n; }
  return r;
}
static v3 v3scale(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }

/* ------------------------------------------------------------------------------------------
 * MAC-grid samplers: third_party/grid.cc:346-417
 * ---------------------------------------------------------------------------------------- */
static v3 get_centered(const dom_t* d, const float* U, int i, int j, int k) {
  v3 r;
  r.x = 0.5f * (U[ATC(d, i, j, k, 0)] + U[ATC(d, i + 1, j, k, 0)]);
  r.y = 0.5f * (U[ATC(d, i, j, k, 1)] + U[ATC(d, i, j + 1, k, 1)]);
  r.z = d->is3d ? 0.5f * (U[ATC(d, i, j, k, 2)] + U[ATC(d, i, j, k + 1, 2)]) : 0.0f;
  return r;
}
static v3 get_at_mac(const dom_t* d, const float* U, int i, int j, int k, int axis) {
  v3 r;
  if (axis == 0) {
    r.x = U[ATC(d, i, j, k, 0)];
    r.y = 0.25f * (U[ATC(d, i, j, k, 1)] + U[ATC(d, i - 1, j, k, 1)] +
                   U[ATC(d, i, j + 1, k, 1)] + U[ATC(d, i - 1, j + 1, k, 1)]);
    r.z = d->is3d ? 0.25f * (U[ATC(d, i, j, k, 2)] + U[ATC(d, i - 1, j, k, 2)] +
                             U[ATC(d, i, j, k + 1, 2)] + U[ATC(d, i - 1, j, k + 1, 2)]) : 0.0f;
  } else if (axis == 1) {
    r.x = 0.25f * (U[ATC(d, i, j, k, 0)] + U[ATC(d, i, j - 1, k, 0)] +
                   U[ATC(d, i + 1, j, k, 0)] + U[ATC(d, i + 1, j - 1, k, 0)]);
    r.y = U[ATC(d, i, j, k, 1)];
    r.z = d->is3d ? 0.25f * (U[ATC(d, i, j, k, 2)] + U[ATC(d, i, j - 1, k, 2)] +
                             U[ATC(d, i, j, k + 1, 2)] + U[ATC(d, i, j - 1, k + 1, 2)]) : 0.0f;
  } else {
    r.x = 0.25f * (U[ATC(d, i, j, k, 0)] + U[ATC(d, i, j, k - 1, 0)] +
                   U[ATC(d, i + 1, j, k, 0)] + U[ATC(d, i + 1, j, k - 1, 0)]);
    r.y = 0.25f * (U[ATC(d, i, j, k, 1)] + U[ATC(d, i, j, k - 1, 1)] +
                   U[ATC(d, i, j + 1, k, 1)] + U[ATC(d, i, j + 1, k - 1, 1)]);
    r.z = d->is3d ? U[ATC(d, i, j, k, 2)] : 0.0f;
  }
  return r;
}

/* ------------------------------------------------------------------------------------------
 * Interpolation: third_party/grid.cc:82-130 (buildIndex), :182-202 (interpol),
 * :204-332 (interpolWithFluid), :437-457 (interpolComponent)
 * ---------------------------------------------------------------------------------------- */
typedef struct { int xi, yi, zi; float s0, s1, t0, t1, f0, f1; } lerp_t;

static lerp_t build_index(const dom_t* d, v3 pos) {
  lerp_t L;
  float px = pos.x - 0.5f, py = pos.y - 0.5f, pz = pos.z - 0.5f;
  L.xi = (int)px; L.yi = (int)py; L.zi = (int)pz;
  L.s1 = px - (float)L.xi; L.s0 = 1.0f - L.s1;
  L.t1 = py - (float)L.yi; L.t0 = 1.0f - L.t1;
  L.f1 = pz - (float)L.zi; L.f0 = 1.0f - L.f1;
  if (px < 0.0f) { L.xi = 0; L.s0 = 1.0f; L.s1 = 0.0f; }
  if (py < 0.0f) { L.yi = 0; L.t0 = 1.0f; L.t1 = 0.0f; }
  if (pz < 0.0f) { L.zi = 0; L.f0 = 1.0f; L.f1 = 0.0f; }
  if (L.xi >= d->X - 1) { L.xi = d->X - 2; L.s0 = 0.0f; L.s1 = 1.0f; }
  if (L.yi >= d->Y - 1) { L.yi = d->Y - 2; L.t0 = 0.0f; L.t1 = 1.0f; }
  if (d->Z > 1 && L.zi >= d->Z - 1) { L.zi = d->Z - 2; L.f0 = 0.0f; L.f1 = 1.0f; }
  return L;
}

/* plain bi/tri-linear sample of channel plane `g` (already offset to the channel) */
static float interpol(const dom_t* d, const float* g, v3 pos) {
  lerp_t L = build_index(d, pos);
  int xi = L.xi, yi = L.yi, zi = L.zi;
  if (d->is3d) {
    return ((g[AT(d, xi, yi, zi)] * L.t0 + g[AT(d, xi, yi + 1, zi)] * L.t1) * L.s0 +
            (g[AT(d, xi + 1, yi, zi)] * L.t0 + g[AT(d, xi + 1, yi + 1, zi)] * L.t1) * L.s1) * L.f0 +
           ((g[AT(d, xi, yi, zi + 1)] * L.t0 + g[AT(d, xi, yi + 1, zi + 1)] * L.t1) * L.s0 +
            (g[AT(d, xi + 1, yi, zi + 1)] * L.t0 + g[AT(d, xi + 1, yi + 1, zi + 1)] * L.t1) * L.s1) * L.f1;
  }
  return (g[AT(d, xi, yi, 0)] * L.t0 + g[AT(d, xi, yi + 1, 0)] * L.t1) * L.s0 +
         (g[AT(d, xi + 1, yi, 0)] * L.t0 + g[AT(d, xi + 1, yi + 1, 0)] * L.t1) * L.s1;
}

/* 1-D lerp that drops non-fluid taps, grid.cc:204-222 */
static void lerp_fluid(float va, int fa, float vb, int fb, float ta, float tb, int* fo, float* vo) {
  if (!fa && !fb) { *vo = 0.0f; *fo = 0; }
  else if (!fa) { *vo = vb; *fo = 1; }
  else if (!fb) { *vo = va; *fo = 1; }
  else { *vo = va * ta + vb * tb; *fo = 1; }
}

static float interpol_with_fluid(const dom_t* d, const float* g, const float* flags, v3 pos) {
  lerp_t L = build_index(d, pos);
  int xi = L.xi, yi = L.yi, zi = d->is3d ? L.zi : 0;
  int f_ab, f_cd, f_abcd, fl;
  float v_ab, v_cd, v_abcd, val;
  lerp_fluid(g[AT(d, xi, yi, zi)], is_fluid(d, flags, xi, yi, zi),
             g[AT(d, xi, yi + 1, zi)], is_fluid(d, flags, xi, yi + 1, zi), L.t0, L.t1, &f_ab, &v_ab);
  lerp_fluid(g[AT(d, xi + 1, yi, zi)], is_fluid(d, flags, xi + 1, yi, zi),
             g[AT(d, xi + 1, yi + 1, zi)], is_fluid(d, flags, xi + 1, yi + 1, zi), L.t0, L.t1,
             &f_cd, &v_cd);
  lerp_fluid(v_ab, f_ab, v_cd, f_cd, L.s0, L.s1, &f_abcd, &v_abcd);
  if (d->is3d) {
    int f_ef, f_gh, f_efgh;
    float v_ef, v_gh, v_efgh;
    lerp_fluid(g[AT(d, xi, yi, zi + 1)], is_fluid(d, flags, xi, yi, zi + 1),
               g[AT(d, xi, yi + 1, zi + 1)], is_fluid(d, flags, xi, yi + 1, zi + 1), L.t0, L.t1,
               &f_ef, &v_ef);
    lerp_fluid(g[AT(d, xi + 1, yi, zi + 1)], is_fluid(d, flags, xi + 1, yi, zi + 1),
               g[AT(d, xi + 1, yi + 1, zi + 1)], is_fluid(d, flags, xi + 1, yi + 1, zi + 1),
               L.t0, L.t1, &f_gh, &v_gh);
    lerp_fluid(v_ef, f_ef, v_gh, f_gh, L.s0, L.s1, &f_efgh, &v_efgh);
    lerp_fluid(v_abcd, f_abcd, v_efgh, f_efgh, L.f0, L.f1, &fl, &val);
  } else {
    fl = f_abcd; val = v_abcd;
  }
  return fl ? val : interpol(d, g, pos);
}

/* ------------------------------------------------------------------------------------------
 * Line trace: generic/calc_line_trace.cc
 * ---------------------------------------------------------------------------------------- */
#define HIT_MARGIN 1e-5f   /* calc_line_trace.cc:22 */
#define TRACE_EPS 1e-12f   /* calc_line_trace.cc:23 */

static int out_of_domain(const dom_t* d, v3 p) { /* :43-51; touching a wall counts as outside */
  return p.x <= 0.0f || p.x >= (float)d->X || p.y <= 0.0f || p.y >= (float)d->Y ||
         p.z <= 0.0f || p.z >= (float)d->Z;
}
/* :86-91 + :53-62; -1 = the reference would THError (cell index outside the grid) */
static int blocked_at(const dom_t* d, const float* flags, v3 p) {
  int i = (int)p.x, j = (int)p.y, k = (int)p.z;
  if (i < 0 || i >= d->X || j < 0 || j >= d->Y || k < 0 || k >= d->Z) return -1;
  return !is_fluid(d, flags, i, j, k);
}

/* Ray/box test, calc_line_trace.cc:101-171 (Graphics Gems RayBox with the reference's fixes). */
static int ray_box(const float* lo, const float* hi, const float* org, const float* dir, float* out) {
  int inside = 1, quad[3], a, which;
  float plane[3], tmax[3];
  const float err_tol = 1e-6f;
  for (a = 0; a < 3; a++) {
    if (org[a] < lo[a]) { quad[a] = 1; plane[a] = lo[a]; inside = 0; }
    else if (org[a] > hi[a]) { quad[a] = 0; plane[a] = hi[a]; inside = 0; }
    else { quad[a] = 2; plane[a] = 0.0f; }
  }
  if (inside) { out[0] = org[0]; out[1] = org[1]; out[2] = org[2]; return 1; }
  for (a = 0; a < 3; a++)
    tmax[a] = (quad[a] != 2 && dir[a] != 0.0f) ? (plane[a] - org[a]) / dir[a] : -1.0f;
  which = 0;
  for (a = 1; a < 3; a++) if (tmax[which] < tmax[a]) which = a;
  if (tmax[which] < 0.0f) return 0;
  for (a = 0; a < 3; a++) {
    if (which != a) {
      out[a] = org[a] + tmax[which] * dir[a];
      if (out[a] < (lo[a] - err_tol) || out[a] > (hi[a] + err_tol)) return 0;
    } else {
      out[a] = plane[a];
    }
  }
  return 1;
}

/* calc_line_trace.cc:205-286: pull `next` back onto the domain wall inset by HIT_MARGIN */
static int ray_border(const dom_t* d, v3 pos, v3 next, v3* ipos) {
  float min_step = FLT_MAX;
  float p[3] = {pos.x, pos.y, pos.z}, n[3] = {next.x, next.y, next.z};
  float sz[3] = {(float)d->X, (float)d->Y, (float)d->Z};
  int a;
  for (a = 0; a < 3; a++) {
    if (n[a] <= HIT_MARGIN) {
      float dd = n[a] - p[a];
      if (fabsf(dd) >= TRACE_EPS) { float st = (HIT_MARGIN - p[a]) / dd; if (st < min_step) min_step = st; }
    }
  }
  for (a = 0; a < 3; a++) {
    if (n[a] >= (sz[a] - HIT_MARGIN)) {
      float dd = n[a] - p[a];
      if (fabsf(dd) >= TRACE_EPS) { float st = (sz[a] - HIT_MARGIN - p[a]) / dd; if (st < min_step) min_step = st; }
    }
  }
  if (min_step < 0.0f || min_step >= FLT_MAX) return 0;
  ipos->x = min_step * (next.x - pos.x) + pos.x;
  ipos->y = min_step * (next.y - pos.y) + pos.y;
  ipos->z = min_step * (next.z - pos.z) + pos.z;
  return 1;
}

static float fminf_std(float a, float b) { return (b < a) ? b : a; } /* std::min(a,b) */
static float fmaxf_std(float a, float b) { return (a < b) ? b : a; } /* std::max(a,b) */

/* calcLineTrace, calc_line_trace.cc:313-503. Returns 1 = hit, 0 = no hit, <0 = the reference
 * would have raised (THError) -- callers count these. */
static int line_trace(const dom_t* d, const float* flags, v3 pos, v3 delta, v3* out) {
  float length, cur = 0.0f;
  v3 dt, next;
  int blk;
  if (out_of_domain(d, pos)) return -1;
  if (blocked_at(d, flags, pos) != 0) return -2;
  *out = pos;
  length = v3norm(delta);
  if (length <= TRACE_EPS) return 0;
  dt.x = delta.x / length; dt.y = delta.y / length; dt.z = delta.z / length;
  while (cur < (length - HIT_MARGIN)) {
    float step = fminf_std(length - cur, 1.0f);
    next.x = out->x + dt.x * step; next.y = out->y + dt.y * step; next.z = out->z + dt.z * step;
    if (out_of_domain(d, next)) {
      v3 ip;
      if (!ray_border(d, *out, next, &ip)) {
        ip.x = fminf_std(fmaxf_std(next.x, HIT_MARGIN), (float)d->X - HIT_MARGIN);
        ip.y = fminf_std(fmaxf_std(next.y, HIT_MARGIN), (float)d->Y - HIT_MARGIN);
        ip.z = fminf_std(fmaxf_std(next.z, HIT_MARGIN), (float)d->Z - HIT_MARGIN);
      }
      if (out_of_domain(d, ip)) return -3;
      blk = blocked_at(d, flags, ip);
      if (blk < 0) return -4;
      if (!blk) { *out = ip; return 1; }
      next = ip;
    }
    blk = blocked_at(d, flags, next);
    if (blk < 0) return -4;
    if (blk) {
      int count;
      for (count = 0; count <= 4; count++) {
        float lo[3], hi[3], org[3], dir[3], hitp[3];
        v3 ctr;
        blk = blocked_at(d, flags, next);
        if (blk < 0) return -4;
        if (!blk) break;
        if (count == 4) return -5;
        ctr.x = (float)((int)next.x) + 0.5f;
        ctr.y = (float)((int)next.y) + 0.5f;
        ctr.z = (float)((int)next.z) + 0.5f;
        lo[0] = ctr.x - 0.5f - HIT_MARGIN; lo[1] = ctr.y - 0.5f - HIT_MARGIN; lo[2] = ctr.z - 0.5f - HIT_MARGIN;
        hi[0] = ctr.x + 0.5f + HIT_MARGIN; hi[1] = ctr.y + 0.5f + HIT_MARGIN; hi[2] = ctr.z + 0.5f + HIT_MARGIN;
        org[0] = out->x; org[1] = out->y; org[2] = out->z;
        dir[0] = dt.x; dir[1] = dt.y; dir[2] = dt.z;
        if (!ray_box(lo, hi, org, dir, hitp)) return 1; /* keep *out (loop invariant: valid) */
        next.x = hitp[0]; next.y = hitp[1]; next.z = hitp[2];
      }
      *out = next;
      if (out_of_domain(d, *out)) return -6;
      if (blocked_at(d, flags, *out) != 0) return -7;
      return 1;
    }
    *out = next;
    cur += step;
  }
  return 0;
}

int ora_calcLineTrace(const float* pos, const float* delta, const float* flags, int Z, int Y,
                      int X, int is3d, float* new_pos) {
  dom_t d = mkdom(Z, Y, X, is3d);
  v3 p = {pos[0], pos[1], pos[2]}, dl = {delta[0], delta[1], delta[2]}, o = {0, 0, 0};
  int r = line_trace(&d, flags, p, dl, &o);
  new_pos[0] = o.x; new_pos[1] = o.y; new_pos[2] = o.z;
  return r;
}

/* ------------------------------------------------------------------------------------------
 * advectScalar: third_party/tfluids.cc:23-588
 * ---------------------------------------------------------------------------------------- */
static v3 cell_centre(int i, int j, int k) {
  v3 p = {(float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f};
  return p;
}
static float sample_s(const dom_t* d, const float* g, const float* flags, v3 p, int outside) {
  return outside ? interpol(d, g, p) : interpol_with_fluid(d, g, flags, p);
}
static v3 sample_vel(const dom_t* d, const float* U, v3 p) { /* 3x interpolComponent */
  v3 r;
  r.x = interpol(d, U, p);
  r.y = interpol(d, U + d->sc, p);
  r.z = d->is3d ? interpol(d, U + 2 * d->sc, p) : 0.0f;
  return r;
}

/* SemiLagrangeEulerOurs[SavePos], tfluids.cc:152-207. pos_out may be NULL. */
static float sl_euler_ours(const dom_t* d, const float* flags, const float* U, const float* src,
                           float dt, int i, int j, int k, int outside, float* pos_out, int* nerr) {
  v3 c = cell_centre(i, j, k), back = c;
  if (is_fluid(d, flags, i, j, k)) {
    v3 disp = v3scale(get_centered(d, U, i, j, k), -dt);
    if (line_trace(d, flags, c, disp, &back) < 0) (*nerr)++;
  }
  if (pos_out) {
    if (!is_fluid(d, flags, i, j, k)) { /* tfluids.cc:160: vec3(i,j,k) + 0.5 */
      back.x = (float)i + 0.5f; back.y = (float)j + 0.5f; back.z = (float)k + 0.5f;
    }
    pos_out[ATC(d, i, j, k, 0)] = back.x;
    pos_out[ATC(d, i, j, k, 1)] = back.y;
    if (d->is3d) pos_out[ATC(d, i, j, k, 2)] = back.z;
  }
  if (!is_fluid(d, flags, i, j, k)) return src[AT(d, i, j, k)];
  return sample_s(d, src, flags, back, outside);
}

/* SemiLagrangeRK2Ours, tfluids.cc:23-77 */
static float sl_rk2_ours(const dom_t* d, const float* flags, const float* U, const float* src,
                         float dt, int i, int j, int k, int outside, int* nerr) {
  v3 c = cell_centre(i, j, k), half, back, disp;
  int hit;
  if (!is_fluid(d, flags, i, j, k)) return src[AT(d, i, j, k)];
  disp = v3scale(get_centered(d, U, i, j, k), -dt * 0.5f);
  hit = line_trace(d, flags, c, disp, &half);
  if (hit < 0) { (*nerr)++; hit = 0; }
  if (hit) return sample_s(d, src, flags, half, outside);
  disp = v3scale(sample_vel(d, U, half), -dt);
  if (line_trace(d, flags, c, disp, &back) < 0) (*nerr)++;
  return sample_s(d, src, flags, back, outside);
}

/* SemiLagrangeRK3Ours, tfluids.cc:79-147 (CPU variant: a k3 hit samples at k3_pos) */
static float sl_rk3_ours(const dom_t* d, const float* flags, const float* U, const float* src,
                         float dt, int i, int j, int k, int outside, int* nerr) {
  v3 c = cell_centre(i, j, k), k1, k2, k3, p2, p3, back, disp;
  int hit;
  if (!is_fluid(d, flags, i, j, k)) return src[AT(d, i, j, k)];
  k1 = get_centered(d, U, i, j, k);
  hit = line_trace(d, flags, c, v3scale(k1, -dt * 0.5f), &p2);
  if (hit < 0) { (*nerr)++; hit = 0; }
  if (hit) return sample_s(d, src, flags, p2, outside);
  k2 = sample_vel(d, U, p2);
  hit = line_trace(d, flags, c, v3scale(k2, -dt * 0.75f), &p3);
  if (hit < 0) { (*nerr)++; hit = 0; }
  if (hit) return sample_s(d, src, flags, p3, outside);
  k3 = sample_vel(d, U, p3);
  {
    /* (real)(2.0/9.0) etc. are rounded to float before the multiply, tfluids.cc:135-137 */
    v3 a = v3scale(k1, -dt * (float)(2.0 / 9.0));
    v3 b = v3scale(k2, -dt * (float)(3.0 / 9.0));
    v3 e = v3scale(k3, -dt * (float)(4.0 / 9.0));
    disp.x = (a.x + b.x) + e.x; disp.y = (a.y + b.y) + e.y; disp.z = (a.z + b.z) + e.z;
  }
  if (line_trace(d, flags, c, disp, &back) < 0) (*nerr)++;
  return sample_s(d, src, flags, back, outside);
}

/* Manta SemiLagrange, tfluids.cc:209-218 */
static float sl_manta(const dom_t* d, const float* U, const float* src, float dt, int i, int j, int k) {
  v3 c = cell_centre(i, j, k), u = get_centered(d, U, i, j, k), p;
  p.x = c.x - u.x * dt; p.y = c.y - u.y * dt; p.z = c.z - u.z * dt;
  return interpol(d, src, p);
}

static void minmax(float* lo, float* hi, float v) {
  if (v < *lo) *lo = v;
  if (v > *hi) *hi = v;
}

/* doClampComponent[MAC], tfluids.cc:250-295 and :701-746. g = channel plane of `orig`. */
static float manta_clamp_component(const dom_t* d, float dst, const float* g, float fwd, v3 pos, v3 vel) {
  float lo = FLT_MAX, hi = -FLT_MAX;
  int l;
  for (l = 0; l < 2; l++) {
    int px, py, pz, i0, j0, k0, i1, j1, k1;
    if (l == 0) { px = (int)(pos.x - vel.x); py = (int)(pos.y - vel.y); pz = (int)(pos.z - vel.z); }
    else { px = (int)(pos.x + vel.x); py = (int)(pos.y + vel.y); pz = (int)(pos.z + vel.z); }
    /* gridSize passed in is size-1, so the upper clamp is size-2 */
    i0 = iclamp(px, 0, d->X - 2);
    j0 = iclamp(py, 0, d->Y - 2);
    k0 = iclamp(pz, 0, d->is3d ? (d->Z - 2) : 1);
    i1 = i0 + 1; j1 = j0 + 1; k1 = d->is3d ? k0 + 1 : k0;
    /* isInBounds(p, 0), grid.cc:42-52: in 2-D z must be exactly 0 */
    if (d->is3d) {
      if (k0 < 0 || k1 >= d->Z) return fwd;
    } else if (k0 != 0 || k1 != 0) {
      return fwd;
    }
    if (i0 < 0 || j0 < 0 || i1 >= d->X || j1 >= d->Y) return fwd;
    minmax(&lo, &hi, g[AT(d, i0, j0, k0)]);
    minmax(&lo, &hi, g[AT(d, i1, j0, k0)]);
    minmax(&lo, &hi, g[AT(d, i0, j1, k0)]);
    minmax(&lo, &hi, g[AT(d, i1, j1, k0)]);
    if (d->is3d) {
      minmax(&lo, &hi, g[AT(d, i0, j0, k1)]);
      minmax(&lo, &hi, g[AT(d, i1, j0, k1)]);
      minmax(&lo, &hi, g[AT(d, i0, j1, k1)]);
      minmax(&lo, &hi, g[AT(d, i1, j1, k1)]);
    }
  }
  return fclamp(dst, lo, hi);
}

/* Manta MacCormackClamp (scalar), tfluids.cc:297-327 */
static float manta_clamp_scalar(const dom_t* d, const float* flags, const float* U, float dval,
                                const float* orig, const float* fwd, float dt, int i, int j, int k) {
  v3 ijk = {(float)i, (float)j, (float)k};
  v3 u = get_centered(d, U, i, j, k), ud = v3scale(u, dt);
  int fx, fy, fz, bx, by, bz, ux = d->X - 1, uy = d->Y - 1, uz = d->Z - 1;
  dval = manta_clamp_component(d, dval, orig, fwd[AT(d, i, j, k)], ijk, ud);
  fx = (int)((ijk.x + 0.5f) - ud.x); fy = (int)((ijk.y + 0.5f) - ud.y); fz = (int)((ijk.z + 0.5f) - ud.z);
  bx = (int)((ijk.x + 0.5f) + ud.x); by = (int)((ijk.y + 0.5f) + ud.y); bz = (int)((ijk.z + 0.5f) + ud.z);
  if (fx < 0 || fy < 0 || fz < 0 || bx < 0 || by < 0 || bz < 0 || fx > ux || fy > uy ||
      (fz > uz && d->is3d) || bx > ux || by > uy || (bz > uz && d->is3d) ||
      is_obst(d, flags, fx, fy, fz) || is_obst(d, flags, bx, by, bz)) {
    dval = fwd[AT(d, i, j, k)];
  }
  return dval;
}

/* getClampBounds + MacCormackClampOurs, tfluids.cc:331-413 (forward position only) */
static float ours_clamp_scalar(const dom_t* d, const float* flags, const float* dst,
                               const float* src, const float* fwd, const float* fwd_pos,
                               int outside, int i, int j, int k) {
  float lo = INFINITY, hi = -INFINITY;
  float px = fwd_pos[ATC(d, i, j, k, 0)], py = fwd_pos[ATC(d, i, j, k, 1)];
  float pz = d->is3d ? fwd_pos[ATC(d, i, j, k, 2)] : 0.0f;
  int i0 = iclamp((int)px, 0, d->X - 1), j0 = iclamp((int)py, 0, d->Y - 1);
  int k0 = d->is3d ? iclamp((int)pz, 0, d->Z - 1) : 0;
  int n = 0, a, b, c;
  for (c = k0 - 1; c <= k0 + 1; c++)
    for (b = j0 - 1; b <= j0 + 1; b++)
      for (a = i0 - 1; a <= i0 + 1; a++) {
        if (c < 0 || c >= d->Z || b < 0 || b >= d->Y || a < 0 || a >= d->X) continue;
        if (outside || is_fluid(d, flags, a, b, c)) { minmax(&lo, &hi, src[AT(d, a, b, c)]); n++; }
      }
  if (n < 1) return fwd[AT(d, i, j, k)];
  return fclamp(dst[AT(d, i, j, k)], lo, hi);
}

/* tfluids_(Main_advectScalar), tfluids.cc:415-588. Buffers: s, flags, fwd, bwd, s_dst are
 * [B][1][Z][Y][X]; U, fwd_pos, bwd_pos are [B][C][Z][Y][X]. Returns -(#cells whose trace
 * would have made the reference raise), 0 when clean. */
int ora_advectScalar(float dt, const float* s, const float* U, const float* flags, float* fwd,
                     float* bwd, int is3d, int method, float* fwd_pos, float* bwd_pos,
                     int sample_outside_fluid, float strength, float* s_dst, int B, int Z,
                     int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b, nerr = 0;
  int mac = (method == M_MACCORMACK || method == M_MACCORMACK_OURS);
  long N = (long)X * Y * Z;
  for (b = 0; b < B; b++) {
    const float* sb = s + b * N;
    const float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    float* fwdb = fwd + b * N;
    float* bwdb = bwd + b * N;
    float* dstb = s_dst + b * N;
    float* fpb = fwd_pos + b * N * C;
    float* bpb = bwd_pos + b * N * C;
    float* cur = mac ? fwdb : dstb;
    int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k) reduction(+ : nerr)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          float v;
          if (on_border(d, i, j, k)) {
            cur[AT(d, i, j, k)] = 0.0f;
            fpb[ATC(d, i, j, k, 0)] = (float)i + 0.5f;
            fpb[ATC(d, i, j, k, 1)] = (float)j + 0.5f;
            if (is3d) fpb[ATC(d, i, j, k, 2)] = (float)k + 0.5f;
            continue;
          }
          switch (method) {
            case M_EULER: case M_MACCORMACK: v = sl_manta(d, Ub, sb, dt, i, j, k); break;
            case M_RK2_OURS: v = sl_rk2_ours(d, fb, Ub, sb, dt, i, j, k, sample_outside_fluid, &nerr); break;
            case M_RK3_OURS: v = sl_rk3_ours(d, fb, Ub, sb, dt, i, j, k, sample_outside_fluid, &nerr); break;
            case M_EULER_OURS: v = sl_euler_ours(d, fb, Ub, sb, dt, i, j, k, sample_outside_fluid, NULL, &nerr); break;
            default: v = sl_euler_ours(d, fb, Ub, sb, dt, i, j, k, sample_outside_fluid, fpb, &nerr); break;
          }
          cur[AT(d, i, j, k)] = v;
        }
    if (!mac) continue;
#pragma omp parallel for collapse(2) private(i, j, k) reduction(+ : nerr)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          if (on_border(d, i, j, k)) {
            bwdb[AT(d, i, j, k)] = 0.0f;
            bpb[ATC(d, i, j, k, 0)] = (float)i + 0.5f;
            bpb[ATC(d, i, j, k, 1)] = (float)j + 0.5f;
            if (is3d) bpb[ATC(d, i, j, k, 2)] = (float)k + 0.5f;
            continue;
          }
          if (method == M_MACCORMACK) bwdb[AT(d, i, j, k)] = sl_manta(d, Ub, fwdb, -dt, i, j, k);
          else bwdb[AT(d, i, j, k)] = sl_euler_ours(d, fb, Ub, fwdb, -dt, i, j, k, sample_outside_fluid, bpb, &nerr);
        }
    /* MacCormackCorrect (no border test), tfluids.cc:220-234; double arithmetic on purpose */
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          long a = AT(d, i, j, k);
          float v = fwdb[a];
          if (is_fluid(d, fb, i, j, k)) v = (float)((double)v + (double)strength * 0.5 * (double)(sb[a] - bwdb[a]));
          dstb[a] = v;
        }
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          if (on_border(d, i, j, k)) continue;
          if (method == M_MACCORMACK)
            dstb[AT(d, i, j, k)] = manta_clamp_scalar(d, fb, Ub, dstb[AT(d, i, j, k)], sb, fwdb, dt, i, j, k);
          else
            dstb[AT(d, i, j, k)] = ours_clamp_scalar(d, fb, dstb, sb, fwdb, fpb, sample_outside_fluid, i, j, k);
        }
  }
  return -nerr;
}

/* ------------------------------------------------------------------------------------------
 * advectVel: third_party/tfluids.cc:594-920
 * ---------------------------------------------------------------------------------------- */
static void sl_mac(const dom_t* d, const float* flags, const float* U, const float* src, float dt,
                   int ours, int i, int j, int k, float* out3, int* nerr) {
  int C = d->is3d ? 3 : 2, c;
  v3 ctr = cell_centre(i, j, k);
  out3[2] = 0.0f;
  if (ours && !is_fluid(d, flags, i, j, k)) { /* tfluids.cc:598-601 */
    for (c = 0; c < C; c++) out3[c] = src[ATC(d, i, j, k, c)];
    return;
  }
  for (c = 0; c < C; c++) {
    v3 u = get_at_mac(d, U, i, j, k, c), p;
    if (ours) {
      if (line_trace(d, flags, ctr, v3scale(u, -dt), &p) < 0) (*nerr)++;
    } else { /* SemiLagrangeMAC, tfluids.cc:634-658 */
      p.x = ctr.x - u.x * dt; p.y = ctr.y - u.y * dt; p.z = ctr.z - u.z * dt;
    }
    out3[c] = interpol(d, src + c * d->sc, p);
  }
}

int ora_advectVel(float dt, const float* U, const float* flags, float* fwd, float* bwd, int is3d,
                  int method, float strength, float* U_dst, int B, int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b, nerr = 0, ours, mac;
  long N = (long)X * Y * Z;
  if (method == M_RK2_OURS || method == M_RK3_OURS) method = M_MACCORMACK_OURS; /* :799-802 */
  ours = (method == M_EULER_OURS || method == M_MACCORMACK_OURS);
  mac = (method == M_MACCORMACK || method == M_MACCORMACK_OURS);
  for (b = 0; b < B; b++) {
    const float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    float* fwdb = fwd + b * N * C;
    float* bwdb = bwd + b * N * C;
    float* dstb = U_dst + b * N * C;
    float* cur = mac ? fwdb : dstb;
    int i, j, k, c;
#pragma omp parallel for collapse(2) private(i, j, k, c) reduction(+ : nerr)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          float v[3] = {0.0f, 0.0f, 0.0f};
          if (!on_border(d, i, j, k)) sl_mac(d, fb, Ub, Ub, dt, ours, i, j, k, v, &nerr);
          for (c = 0; c < C; c++) cur[ATC(d, i, j, k, c)] = v[c];
        }
    if (!mac) continue;
#pragma omp parallel for collapse(2) private(i, j, k, c) reduction(+ : nerr)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          float v[3] = {0.0f, 0.0f, 0.0f};
          if (!on_border(d, i, j, k)) sl_mac(d, fb, Ub, fwdb, -dt, ours, i, j, k, v, &nerr);
          for (c = 0; c < C; c++) bwdb[ATC(d, i, j, k, c)] = v[c];
        }
    /* MacCormackCorrectMAC, tfluids.cc:660-699 */
#pragma omp parallel for collapse(2) private(i, j, k, c)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          int skip[3] = {0, 0, 0};
          if (!is_fluid(d, fb, i, j, k)) skip[0] = skip[1] = skip[2] = 1;
          if (i > 0 && !is_fluid(d, fb, i - 1, j, k)) skip[0] = 1;
          if (j > 0 && !is_fluid(d, fb, i, j - 1, k)) skip[1] = 1;
          if (is3d && k > 0 && !is_fluid(d, fb, i, j, k - 1)) skip[2] = 1;
          for (c = 0; c < C; c++) {
            long a = ATC(d, i, j, k, c);
            float v = fwdb[a];
            if (!skip[c]) v = (float)((double)v + (double)strength * 0.5 * (double)(Ub[a] - bwdb[a]));
            dstb[a] = v;
          }
        }
    /* MacCormackClampMAC, tfluids.cc:748-774 */
#pragma omp parallel for collapse(2) private(i, j, k, c)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          v3 ijk = {(float)i, (float)j, (float)k};
          if (on_border(d, i, j, k)) continue;
          for (c = 0; c < C; c++) {
            long a = ATC(d, i, j, k, c);
            v3 ud = v3scale(get_at_mac(d, Ub, i, j, k, c), dt);
            dstb[a] = manta_clamp_component(d, dstb[a], Ub + c * d->sc, fwdb[a], ijk, ud);
          }
        }
  }
  return -nerr;
}

/* ------------------------------------------------------------------------------------------
 * Stencil ops
 * ---------------------------------------------------------------------------------------- */
/* tfluids.cc:926-1002 */
void ora_setWallBcsForward(float* U, const float* flags, int is3d, int B, int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b;
  long N = (long)X * Y * Z;
  for (b = 0; b < B; b++) {
    float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          int cf = is_fluid(d, fb, i, j, k), co = is_obst(d, fb, i, j, k);
          if (!cf && !co) continue;
          if (i > 0 && is_obst(d, fb, i - 1, j, k)) Ub[ATC(d, i, j, k, 0)] = 0.0f;
          if (i > 0 && co && is_fluid(d, fb, i - 1, j, k)) Ub[ATC(d, i, j, k, 0)] = 0.0f;
          if (j > 0 && is_obst(d, fb, i, j - 1, k)) Ub[ATC(d, i, j, k, 1)] = 0.0f;
          if (j > 0 && co && is_fluid(d, fb, i, j - 1, k)) Ub[ATC(d, i, j, k, 1)] = 0.0f;
          if (k > 0 && is_obst(d, fb, i, j, k - 1)) Ub[ATC(d, i, j, k, 2)] = 0.0f;
          if (k > 0 && co && is_fluid(d, fb, i, j, k - 1)) Ub[ATC(d, i, j, k, 2)] = 0.0f;
          if (cf) {
            if ((i > 0 && is_stick(d, fb, i - 1, j, k)) || (i < X - 1 && is_stick(d, fb, i + 1, j, k))) {
              Ub[ATC(d, i, j, k, 1)] = 0.0f;
              if (is3d) Ub[ATC(d, i, j, k, 2)] = 0.0f;
            }
            if ((j > 0 && is_stick(d, fb, i, j - 1, k)) || (j < Y - 1 && is_stick(d, fb, i, j + 1, k))) {
              Ub[ATC(d, i, j, k, 0)] = 0.0f;
              if (is3d) Ub[ATC(d, i, j, k, 2)] = 0.0f;
            }
            if (is3d && ((k > 0 && is_stick(d, fb, i, j, k - 1)) || (k < Z - 1 && is_stick(d, fb, i, j, k + 1)))) {
              Ub[ATC(d, i, j, k, 0)] = 0.0f;
              Ub[ATC(d, i, j, k, 1)] = 0.0f;
            }
          }
        }
  }
}

/* tfluids.cc:1008-1066 (negative divergence, Manta makeRhs convention) */
void ora_velocityDivergenceForward(const float* U, const float* flags, float* div, int is3d,
                                   int B, int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b;
  long N = (long)X * Y * Z;
  for (b = 0; b < B; b++) {
    const float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    float* db = div + b * N;
    int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          float v;
          if (on_border(d, i, j, k) || !is_fluid(d, fb, i, j, k)) { db[AT(d, i, j, k)] = 0.0f; continue; }
          v = Ub[ATC(d, i, j, k, 0)] - Ub[ATC(d, i + 1, j, k, 0)] + Ub[ATC(d, i, j, k, 1)] -
              Ub[ATC(d, i, j + 1, k, 1)];
          if (is3d) v += (Ub[ATC(d, i, j, k, 2)] - Ub[ATC(d, i, j, k + 1, 2)]);
          db[AT(d, i, j, k)] = v;
        }
  }
}

/* tfluids.cc:1072-1156 */
void ora_velocityUpdateForward(float* U, const float* flags, const float* p, int is3d, int B,
                               int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b;
  long N = (long)X * Y * Z;
  for (b = 0; b < B; b++) {
    float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    const float* pb = p + b * N;
    int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          if (on_border(d, i, j, k)) continue;
          if (is_fluid(d, fb, i, j, k)) {
            float pc = pb[AT(d, i, j, k)];
            if (is_fluid(d, fb, i - 1, j, k)) Ub[ATC(d, i, j, k, 0)] -= (pc - pb[AT(d, i - 1, j, k)]);
            if (is_fluid(d, fb, i, j - 1, k)) Ub[ATC(d, i, j, k, 1)] -= (pc - pb[AT(d, i, j - 1, k)]);
            if (is3d && is_fluid(d, fb, i, j, k - 1)) Ub[ATC(d, i, j, k, 2)] -= (pc - pb[AT(d, i, j, k - 1)]);
            if (is_empty(d, fb, i - 1, j, k)) Ub[ATC(d, i, j, k, 0)] -= pc;
            if (is_empty(d, fb, i, j - 1, k)) Ub[ATC(d, i, j, k, 1)] -= pc;
            if (is3d && is_empty(d, fb, i, j, k - 1)) Ub[ATC(d, i, j, k, 2)] -= pc;
          } else if (is_empty(d, fb, i, j, k) && !is_outflow(d, fb, i, j, k)) {
            if (is_fluid(d, fb, i - 1, j, k)) Ub[ATC(d, i, j, k, 0)] += pb[AT(d, i - 1, j, k)];
            else Ub[ATC(d, i, j, k, 0)] = 0.0f;
            if (is_fluid(d, fb, i, j - 1, k)) Ub[ATC(d, i, j, k, 1)] += pb[AT(d, i, j - 1, k)];
            else Ub[ATC(d, i, j, k, 1)] = 0.0f;
            if (is3d) {
              if (is_fluid(d, fb, i, j, k - 1)) Ub[ATC(d, i, j, k, 2)] += pb[AT(d, i, j, k - 1)];
              else Ub[ATC(d, i, j, k, 2)] = 0.0f;
            }
          }
        }
  }
}

static float get_dx(int Z, int Y, int X) { return 1.0f / (float)imax(X, imax(Y, Z)); } /* grid.cc:37-40 */

/* tfluids.cc:1162-1233 */
void ora_addBuoyancy(float* U, const float* flags, const float* density, const float* gravity,
                     float dt, int is3d, int B, int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b;
  long N = (long)X * Y * Z;
  float sc = dt / get_dx(Z, Y, X);
  float sx = -gravity[0] * sc, sy = -gravity[1] * sc, sz = -gravity[2] * sc;
  for (b = 0; b < B; b++) {
    float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    const float* r = density + b * N;
    int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          if (on_border(d, i, j, k) || !is_fluid(d, fb, i, j, k)) continue;
          if (is_fluid(d, fb, i - 1, j, k))
            Ub[ATC(d, i, j, k, 0)] += (0.5f * sx * (r[AT(d, i, j, k)] + r[AT(d, i - 1, j, k)]));
          if (is_fluid(d, fb, i, j - 1, k))
            Ub[ATC(d, i, j, k, 1)] += (0.5f * sy * (r[AT(d, i, j, k)] + r[AT(d, i, j - 1, k)]));
          if (is3d && is_fluid(d, fb, i, j, k - 1))
            Ub[ATC(d, i, j, k, 2)] += (0.5f * sz * (r[AT(d, i, j, k)] + r[AT(d, i, j, k - 1)]));
        }
  }
}

/* tfluids.cc:1239-1306 */
void ora_addGravity(float* U, const float* flags, const float* gravity, float dt, int is3d, int B,
                    int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b;
  long N = (long)X * Y * Z;
  float sc = dt / get_dx(Z, Y, X);
  float fx = gravity[0] * sc, fy = gravity[1] * sc, fz = gravity[2] * sc;
  for (b = 0; b < B; b++) {
    float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          int cf, ce;
          if (on_border(d, i, j, k)) continue;
          cf = is_fluid(d, fb, i, j, k); ce = is_empty(d, fb, i, j, k);
          if (!cf && !ce) continue;
          if (is_fluid(d, fb, i - 1, j, k) || (cf && is_empty(d, fb, i - 1, j, k))) Ub[ATC(d, i, j, k, 0)] += fx;
          if (is_fluid(d, fb, i, j - 1, k) || (cf && is_empty(d, fb, i, j - 1, k))) Ub[ATC(d, i, j, k, 1)] += fy;
          if (is3d && (is_fluid(d, fb, i, j, k - 1) || (cf && is_empty(d, fb, i, j, k - 1)))) Ub[ATC(d, i, j, k, 2)] += fz;
        }
  }
}

/* tfluids.cc:1312-1458. Temps: centered/force [B][C][..], curl [B][3][..], curl_norm [B][1][..]. */
void ora_vorticityConfinement(float* U, const float* flags, float strength, float* centered,
                              float* curl, float* curl_norm, float* force, int is3d, int B, int Z,
                              int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b;
  long N = (long)X * Y * Z;
  for (b = 0; b < B; b++) {
    float* Ub = U + b * N * C;
    const float* fb = flags + b * N;
    float* ce = centered + b * N * C;
    float* cu = curl + b * N * 3;
    float* cn = curl_norm + b * N;
    float* fo = force + b * N * C;
    int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          v3 v = {0.0f, 0.0f, 0.0f};
          if (!on_border(d, i, j, k)) v = get_centered(d, Ub, i, j, k);
          ce[ATC(d, i, j, k, 0)] = v.x; ce[ATC(d, i, j, k, 1)] = v.y;
          if (is3d) ce[ATC(d, i, j, k, 2)] = v.z;
        }
    /* VecGrid::curl, grid.cc:497-515 */
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          v3 w = {0.0f, 0.0f, 0.0f};
          float nrm = 0.0f;
          if (!on_border(d, i, j, k)) {
            w.z = 0.5f * ((ce[ATC(d, i + 1, j, k, 1)] - ce[ATC(d, i - 1, j, k, 1)]) -
                          (ce[ATC(d, i, j + 1, k, 0)] - ce[ATC(d, i, j - 1, k, 0)]));
            if (is3d) {
              w.x = 0.5f * ((ce[ATC(d, i, j + 1, k, 2)] - ce[ATC(d, i, j - 1, k, 2)]) -
                            (ce[ATC(d, i, j, k + 1, 1)] - ce[ATC(d, i, j, k - 1, 1)]));
              w.y = 0.5f * ((ce[ATC(d, i, j, k + 1, 0)] - ce[ATC(d, i, j, k - 1, 0)]) -
                            (ce[ATC(d, i + 1, j, k, 2)] - ce[ATC(d, i - 1, j, k, 2)]));
            }
            nrm = v3norm(w);
          }
          cu[ATC(d, i, j, k, 0)] = w.x; cu[ATC(d, i, j, k, 1)] = w.y; cu[ATC(d, i, j, k, 2)] = w.z;
          cn[AT(d, i, j, k)] = nrm;
        }
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          v3 f = {0.0f, 0.0f, 0.0f};
          if (!on_border(d, i, j, k)) {
            v3 g = {0.0f, 0.0f, 0.0f}, w;
            g.x = 0.5f * (cn[AT(d, i + 1, j, k)] - cn[AT(d, i - 1, j, k)]);
            g.y = 0.5f * (cn[AT(d, i, j + 1, k)] - cn[AT(d, i, j - 1, k)]);
            if (is3d) g.z = 0.5f * (cn[AT(d, i, j, k + 1)] - cn[AT(d, i, j, k - 1)]);
            g = v3normalize(g);
            w.x = cu[ATC(d, i, j, k, 0)]; w.y = cu[ATC(d, i, j, k, 1)]; w.z = cu[ATC(d, i, j, k, 2)];
            f.x = ((g.y * w.z) - (g.z * w.y)) * strength;
            f.y = ((g.z * w.x) - (g.x * w.z)) * strength;
            f.z = ((g.x * w.y) - (g.y * w.x)) * strength;
          }
          fo[ATC(d, i, j, k, 0)] = f.x; fo[ATC(d, i, j, k, 1)] = f.y;
          if (is3d) fo[ATC(d, i, j, k, 2)] = f.z;
        }
    /* AddForceField, tfluids.cc:1312-1339 */
#pragma omp parallel for collapse(2) private(i, j, k)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          int cf, cem;
          if (on_border(d, i, j, k)) continue;
          cf = is_fluid(d, fb, i, j, k); cem = is_empty(d, fb, i, j, k);
          if (!cf && !cem) continue;
          if (is_fluid(d, fb, i - 1, j, k) || (cf && is_empty(d, fb, i - 1, j, k)))
            Ub[ATC(d, i, j, k, 0)] += (0.5f * (fo[ATC(d, i - 1, j, k, 0)] + fo[ATC(d, i, j, k, 0)]));
          if (is_fluid(d, fb, i, j - 1, k) || (cf && is_empty(d, fb, i, j - 1, k)))
            Ub[ATC(d, i, j, k, 1)] += (0.5f * (fo[ATC(d, i, j - 1, k, 1)] + fo[ATC(d, i, j, k, 1)]));
          if (is3d && (is_fluid(d, fb, i, j, k - 1) || (cf && is_empty(d, fb, i, j, k - 1))))
            Ub[ATC(d, i, j, k, 2)] += (0.5f * (fo[ATC(d, i, j, k - 1, 2)] + fo[ATC(d, i, j, k, 2)]));
        }
  }
}

/* ------------------------------------------------------------------------------------------
 * Training-side operators and the nearest-neighbour resampler (8f-1).
 * The reference scatters with `#pragma omp atomic`; with one thread its loop order (k, j, i ascending) fixes the
 * order in which contributions reach a word. These restatements are that serial loop, no OpenMP, so they are
 * deterministic and bit-equal to the reference run with one thread.
 * ---------------------------------------------------------------------------------------- */
/* generic/tfluids.cc:49-130 */
void ora_velocityDivergenceBackward(const float* flags, const float* grad_out, float* grad_u, int is3d, int B,
                                    int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b, i, j, k;
  long N = (long)X * Y * Z;
  for (b = 0; b < B; b++) {
    const float* fb = flags + b * N;
    const float* go = grad_out + b * N;
    float* gu = grad_u + b * N * C;
    memset(gu, 0, sizeof(float) * N * C);
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          float g;
          if (on_border(d, i, j, k) || !is_fluid(d, fb, i, j, k)) continue;
          g = go[AT(d, i, j, k)];
          gu[ATC(d, i, j, k, 0)] += g; gu[ATC(d, i + 1, j, k, 0)] -= g;
          gu[ATC(d, i, j, k, 1)] += g; gu[ATC(d, i, j + 1, k, 1)] -= g;
          if (is3d) { gu[ATC(d, i, j, k, 2)] += g; gu[ATC(d, i, j, k + 1, 2)] -= g; }
        }
  }
}

/* generic/tfluids.cc:216-344 */
void ora_velocityUpdateBackward(const float* flags, const float* grad_out, float* grad_p, int is3d, int B, int Z,
                                int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  int C = is3d ? 3 : 2, b, i, j, k;
  long N = (long)X * Y * Z;
  for (b = 0; b < B; b++) {
    const float* fb = flags + b * N;
    const float* go = grad_out + b * N * C;
    float* gp = grad_p + b * N;
    memset(gp, 0, sizeof(float) * N);
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          float gx, gy, gz;
          if (on_border(d, i, j, k)) continue;
          gx = go[ATC(d, i, j, k, 0)]; gy = go[ATC(d, i, j, k, 1)]; gz = is3d ? go[ATC(d, i, j, k, 2)] : 0.0f;
          if (is_fluid(d, fb, i, j, k)) {
            if (is_fluid(d, fb, i - 1, j, k)) { gp[AT(d, i, j, k)] -= gx; gp[AT(d, i - 1, j, k)] += gx; }
            if (is_fluid(d, fb, i, j - 1, k)) { gp[AT(d, i, j, k)] -= gy; gp[AT(d, i, j - 1, k)] += gy; }
            if (is3d && is_fluid(d, fb, i, j, k - 1)) { gp[AT(d, i, j, k)] -= gz; gp[AT(d, i, j, k - 1)] += gz; }
            if (is_empty(d, fb, i - 1, j, k)) gp[AT(d, i, j, k)] -= gx;
            if (is_empty(d, fb, i, j - 1, k)) gp[AT(d, i, j, k)] -= gy;
            if (is3d && is_empty(d, fb, i, j, k - 1)) gp[AT(d, i, j, k)] -= gz;
          } else if (is_empty(d, fb, i, j, k) && !is_outflow(d, fb, i, j, k)) {
            if (is_fluid(d, fb, i - 1, j, k)) gp[AT(d, i - 1, j, k)] += gx;
            if (is_fluid(d, fb, i, j - 1, k)) gp[AT(d, i, j - 1, k)] += gy;
            if (is3d && is_fluid(d, fb, i, j, k - 1)) gp[AT(d, i, j, k - 1)] += gz;
          }
        }
  }
}

/* generic/tfluids.cc:509-556; rows = B * nfeat */
void ora_volumetricUpSamplingNearestForward(int ratio, const float* in, float* out, long rows, int Zi, int Yi,
                                            int Xi) {
  long r; int z, y, x;
  int Zo = Zi * ratio, Yo = Yi * ratio, Xo = Xi * ratio;
  for (r = 0; r < rows; r++)
    for (z = 0; z < Zo; z++)
      for (y = 0; y < Yo; y++)
        for (x = 0; x < Xo; x++)
          out[((r * Zo + z) * Yo + y) * (long)Xo + x] = in[((r * Zi + z / ratio) * Yi + y / ratio) * (long)Xi + x / ratio];
}

/* generic/tfluids.cc:562-633 (float accumulator, window order z, y, x) */
void ora_volumetricUpSamplingNearestBackward(int ratio, const float* grad_out, float* grad_in, long rows, int Zi,
                                             int Yi, int Xi) {
  long r; int z, y, x, zu, yu, xu;
  int Zo = Zi * ratio, Yo = Yi * ratio, Xo = Xi * ratio;
  for (r = 0; r < rows; r++)
    for (z = 0; z < Zi; z++)
      for (y = 0; y < Yi; y++)
        for (x = 0; x < Xi; x++) {
          float sum = 0.0f;
          for (zu = 0; zu < ratio; zu++)
            for (yu = 0; yu < ratio; yu++)
              for (xu = 0; xu < ratio; xu++)
                sum += grad_out[((r * Zo + z * ratio + zu) * Yo + y * ratio + yu) * (long)Xo + x * ratio + xu];
          grad_in[((r * Zi + z) * Yi + y) * (long)Xi + x] = sum;
        }
}

/* generic/tfluids.cc:136-167 */
void ora_emptyDomain(float* flags, int is3d, int bnd, int B, int Z, int Y, int X) {
  long N = (long)X * Y * Z;
  int b, i, j, k;
  for (b = 0; b < B; b++)
    for (k = 0; k < Z; k++)
      for (j = 0; j < Y; j++)
        for (i = 0; i < X; i++) {
          int border = i < bnd || i > X - 1 - bnd || j < bnd || j > Y - 1 - bnd ||
                       (is3d && (k < bnd || k > Z - 1 - bnd));
          flags[b * N + (long)k * Y * X + (long)j * X + i] = border ? (float)F_OBSTACLE : (float)F_FLUID;
        }
}

/* generic/tfluids.cc:173-210; returns -1 when a cell is neither exactly Fluid nor Obstacle */
int ora_flagsToOccupancy(const float* flags, float* occ, long numel) {
  long n;
  int bad = 0;
  for (n = 0; n < numel; n++) {
    int f = (int)flags[n];
    if (f == F_FLUID) occ[n] = 0.0f;
    else if (f == F_OBSTACLE) occ[n] = 1.0f;
    else bad = 1;
  }
  return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * rectangularBlur, generic/tfluids.cc:642-760: separable box blur of radius `rad` with clamped edges,
 * evaluated per line as a running sum (the arithmetic ORDER is the reference's: that is what makes
 * the restatement bit-exact). 3-D: z pass src -> dst, y pass dst -> tmp, x pass tmp -> dst;
 * 2-D: y pass src -> tmp, x pass tmp -> dst. Fields are [B][C][Z][Y][X].
 * ---------------------------------------------------------------------------------------- */
static void blur_axis(const float* src, int size, long stride, int rad, float* dst) {
  float val = src[0] * (float)(rad + 1);
  int i;
  for (i = 0; i < size && i < rad; i++) val += src[i * stride];
  const float mul_const = 1.0f / (float)(rad * 2 + 1);
  for (i = 0; i < size; i++) {
    const int iminus = i - rad - 1 > 0 ? i - rad - 1 : 0;
    const int iplus = i + rad < size - 1 ? i + rad : size - 1;
    val -= src[iminus * stride];
    val += src[iplus * stride];
    dst[i * stride] = val * mul_const;
  }
}

void ora_rectangularBlur(const float* src, int rad, int is3d, float* dst, float* tmp, int B, int C, int Z, int Y, int X) {
  const long sy = X, sz = (long)Y * X, sf = (long)Z * Y * X;
  const float* cur_src = src;
  float* cur_dst = is3d ? dst : tmp;
  long f; int z, y, x;
  if (is3d) {
    for (f = 0; f < (long)B * C; f++)
      for (y = 0; y < Y; y++)
        for (x = 0; x < X; x++) blur_axis(cur_src + f * sf + y * sy + x, Z, sz, rad, cur_dst + f * sf + y * sy + x);
    cur_src = dst; cur_dst = tmp;
  }
  for (f = 0; f < (long)B * C; f++)
    for (z = 0; z < Z; z++)
      for (x = 0; x < X; x++) blur_axis(cur_src + f * sf + z * sz + x, Y, sy, rad, cur_dst + f * sf + z * sz + x);
  cur_src = tmp; cur_dst = dst;
  for (f = 0; f < (long)B * C; f++)
    for (z = 0; z < Z; z++)
      for (y = 0; y < Y; y++) blur_axis(cur_src + f * sf + z * sz + y * sy, X, 1, rad, cur_dst + f * sf + z * sz + y * sy);
}

/* signedDistanceField, generic/tfluids.cc:766-821: distance to the nearest obstacle cell inside a
 * (2 rad + 1)^dim window, clamped to rad; 0 in obstacle cells. flags, dst: [B][1][Z][Y][X]. */
void ora_signedDistanceField(const float* flags, int rad, float* dst, int B, int Z, int Y, int X) {
  const long N = (long)Z * Y * X;
  int b, z, y, x;
  for (b = 0; b < B; b++)
    for (z = 0; z < Z; z++)
      for (y = 0; y < Y; y++)
        for (x = 0; x < X; x++) {
          const float* f = flags + b * N;
          const long o = b * N + (long)z * Y * X + (long)y * X + x;
          if ((int)f[(long)z * Y * X + (long)y * X + x] & F_OBSTACLE) { dst[o] = 0.0f; continue; }
          float dist_sq = (float)(rad * rad);
          const int z0 = z - rad > 0 ? z - rad : 0, z1 = z + rad < Z - 1 ? z + rad : Z - 1;
          const int y0 = y - rad > 0 ? y - rad : 0, y1 = y + rad < Y - 1 ? y + rad : Y - 1;
          const int x0 = x - rad > 0 ? x - rad : 0, x1 = x + rad < X - 1 ? x + rad : X - 1;
          int zs, ys, xs;
          for (zs = z0; zs <= z1; zs++)
            for (ys = y0; ys <= y1; ys++)
              for (xs = x0; xs <= x1; xs++)
                if ((int)f[(long)zs * Y * X + (long)ys * X + xs] & F_OBSTACLE) {
                  const float cur = (float)((z - zs) * (z - zs) + (y - ys) * (y - ys) + (x - xs) * (x - xs));
                  if (dist_sq > cur) dist_sq = cur;
                }
          dst[o] = sqrtf(dist_sq);
        }
}

/* ------------------------------------------------------------------------------------------
 * solveLinearSystemJacobi -- restates the CUDA path generic/tfluids.cu:1765-1921 (no CPU version
 * exists in the reference: generic/tfluids.cc:836-839). Pinned bit for bit to the reference's kernel + host loop compiled
 * for the host (oracle/ref_jacobi.cc; tests/test_oracle.py).
 * p_prev is scratch [B][1][Z][Y][X]. Returns the last residual max_b ||p - p_prev||_2.
 * ---------------------------------------------------------------------------------------- */
float ora_solveLinearSystemJacobi(float* p, const float* flags, const float* div, float* p_prev,
                                  int is3d, float p_tol, int max_iter, int B, int Z, int Y, int X) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  long N = (long)X * Y * Z, n;
  float* cur = p;
  float* prev = p_prev;
  float residual = 0.0f;
  int iter = 0, b;
  memset(p, 0, sizeof(float) * N * B);
  memset(p_prev, 0, sizeof(float) * N * B);
  for (;;) {
    for (b = 0; b < B; b++) {
      const float* fb = flags + b * N;
      const float* db = div + b * N;
      float* pc = cur + b * N;
      const float* pp = prev + b * N;
      int i, j, k;
#pragma omp parallel for collapse(2) private(i, j, k)
      for (k = 0; k < Z; k++)
        for (j = 0; j < Y; j++)
          for (i = 0; i < X; i++) {
            float c, p1, p2, p3, p4, p5, p6, den;
            if (on_border(d, i, j, k) || is_obst(d, fb, i, j, k)) { pc[AT(d, i, j, k)] = 0.0f; continue; }
            c = pp[AT(d, i, j, k)];
            p1 = pp[AT(d, i - 1, j, k)]; p2 = pp[AT(d, i + 1, j, k)];
            p3 = pp[AT(d, i, j - 1, k)]; p4 = pp[AT(d, i, j + 1, k)];
            p5 = is3d ? pp[AT(d, i, j, k - 1)] : 0.0f;
            p6 = is3d ? pp[AT(d, i, j, k + 1)] : 0.0f;
            if (is_obst(d, fb, i - 1, j, k)) p1 = c;
            if (is_obst(d, fb, i + 1, j, k)) p2 = c;
            if (is_obst(d, fb, i, j - 1, k)) p3 = c;
            if (is_obst(d, fb, i, j + 1, k)) p4 = c;
            if (is3d && is_obst(d, fb, i, j, k - 1)) p5 = c;
            if (is3d && is_obst(d, fb, i, j, k + 1)) p6 = c;
            den = is3d ? 6.0f : 4.0f;
            pc[AT(d, i, j, k)] = (p1 + p2 + p3 + p4 + p5 + p6 + db[AT(d, i, j, k)]) / den;
          }
    }
    /* residual = max over batch of the L2 norm of (p - p_prev), double accumulate */
    residual = 0.0f;
    for (b = 0; b < B; b++) {
      double acc = 0.0;
      for (n = 0; n < N; n++) { double e = (double)p[b * N + n] - (double)p_prev[b * N + n]; acc += e * e; }
      if ((float)sqrt(acc) > residual) residual = (float)sqrt(acc);
    }
    if (residual < p_tol) break;
    iter++;
    if (iter >= max_iter) break;
    { float* t = cur; cur = prev; prev = t; }
  }
  if (cur == p_prev) memcpy(p, p_prev, sizeof(float) * N * B);
  return residual;
}

/* ------------------------------------------------------------------------------------------
 * solveLinearSystemPCG -- restates the CUDA-only reference path generic/tfluids.cu:864-1759 the way the
 * reference structures it: flood-fill components (generic/find_connected_fluid_components.cc), reduced
 * system indices (:864-906), CSR Laplacian (setupLaplacian :908-1093), then PCG (Golub & Van Loan 10.3.1,
 * :1527-1714) with GENERIC CSR ILU(0) / IC(0) factorisations and triangular solves standing in for
 * cusparseScsrilu0 / cusparseScsric0 / cusparseScsrsv (cuSPARSE and cuBLAS are not in the tree: their
 * published algorithms are restated; summation orders inside them are unspecified). PINNED since round 4: bit-equal to the
 * reference's own host function (:864-1759) compiled for the host by oracle/ref_pcg.cc over oracle/ref_shim/cusparse_host.h
 * (tests/test_oracle.py::test_pcg_restatement_equals_compiled_reference_host_function); what stays outside any pin is
 * cuSPARSE's internal summation order, so the device solver is compared on converged solutions (5e-5) + the properties
 * test_tfluids.lua:836-906 checks.
 * precond: 0 none, 1 ilu0, 2 ic0. Returns 0, -1 (fluid cell on the border / non-fluid in a component), -2 NaN.
 * ---------------------------------------------------------------------------------------- */
static float clamp_to_eps(float v) {   /* generic/tfluids.cu:1203-1214 */
  const float eps = FLT_MIN;
  if (fabsf(v) < eps) return v < 0 ? fminf(v, -eps) : fmaxf(v, eps);
  return v;
}
static float sdot(long n, const float* a, const float* b) {
  double acc = 0.0; long t;
  for (t = 0; t < n; t++) acc += (double)a[t] * b[t];
  return (float)acc;
}
/* y = A x, CSR; sym_upper: only the upper triangle is stored (CUSPARSE_MATRIX_TYPE_SYMMETRIC) */
static void csrmv(long n, const int* row, const int* col, const float* val, int sym_upper, const float* x, float* y) {
  long r; int q;
  for (r = 0; r < n; r++) y[r] = 0.0f;
  for (r = 0; r < n; r++)
    for (q = row[r]; q < row[r + 1]; q++) {
      y[r] += val[q] * x[col[q]];
      if (sym_upper && col[q] != r) y[col[q]] += val[q] * x[r];
    }
}
static int csr_find(const int* row, const int* col, long r, int c) {
  int q;
  for (q = row[r]; q < row[r + 1]; q++) if (col[q] == c) return q;
  return -1;
}
/* in-place ILU(0) of a full CSR matrix with sorted columns (IKJ variant) */
static void csr_ilu0(long n, const int* row, const int* col, float* val) {
  long i; int q, q2;
  for (i = 1; i < n; i++)
    for (q = row[i]; q < row[i + 1] && col[q] < i; q++) {
      const int k = col[q];
      val[q] = val[q] / val[csr_find(row, col, k, k)];
      for (q2 = q + 1; q2 < row[i + 1]; q2++) {
        const int kj = csr_find(row, col, k, col[q2]);
        if (kj >= 0) val[q2] -= val[q] * val[kj];
      }
    }
}
/* in-place IC(0) of an upper-triangular CSR matrix: A ~ R^T R */
static void csr_ic0_upper(long n, const int* row, const int* col, float* val) {
  long k; int q, q2;
  for (k = 0; k < n; k++) {
    const int dq = row[k];             /* the diagonal is the first entry of an upper-triangular row */
    val[dq] = sqrtf(val[dq]);
    for (q = dq + 1; q < row[k + 1]; q++) val[q] = val[q] / val[dq];
    for (q = dq + 1; q < row[k + 1]; q++) {
      const int j = col[q];
      for (q2 = q; q2 < row[k + 1]; q2++) {      /* a(j,l) -= r(k,j) r(k,l) for l >= j in the pattern of row j */
        const int jl = csr_find(row, col, j, col[q2]);
        if (jl >= 0) val[jl] -= val[q] * val[q2];
      }
    }
  }
}
/* lower solve with the strictly-lower part of a full CSR (unit diagonal): L y = b */
static void csr_lsolve_unit(long n, const int* row, const int* col, const float* val, const float* b, float* y) {
  long i; int q;
  for (i = 0; i < n; i++) {
    float v = b[i];
    for (q = row[i]; q < row[i + 1] && col[q] < i; q++) v -= val[q] * y[col[q]];
    y[i] = v;
  }
}
/* upper solve with the upper part (incl. diagonal) of a CSR: U z = y; works for full and upper-only storage */
static void csr_usolve(long n, const int* row, const int* col, const float* val, const float* y, float* z) {
  long i; int q;
  for (i = n - 1; i >= 0; i--) {
    float v = y[i], dg = 1.0f;
    for (q = row[i]; q < row[i + 1]; q++) {
      if (col[q] == i) dg = val[q];
      else if (col[q] > i) v -= val[q] * z[col[q]];
    }
    z[i] = v / dg;
  }
}
/* R^T y = b with R upper-triangular CSR (column-oriented forward substitution) */
static void csr_utsolve(long n, const int* row, const int* col, const float* val, const float* b, float* y) {
  long i; int q;
  for (i = 0; i < n; i++) y[i] = b[i];
  for (i = 0; i < n; i++) {
    y[i] = y[i] / val[row[i]];
    for (q = row[i] + 1; q < row[i + 1]; q++) y[col[q]] -= val[q] * y[i];
  }
}

int ora_solveLinearSystemPCG(float* p, const float* flags, const float* div, int is3d, int precond, float tol,
                             int max_iter, int B, int Z, int Y, int X, float* out_residual) {
  dom_t dm = mkdom(Z, Y, X, is3d);
  const dom_t* d = &dm;
  const long N = (long)X * Y * Z;
  int* comp = (int*)malloc(sizeof(int) * N);
  int* sysidx = (int*)malloc(sizeof(int) * N);
  int* stack = (int*)malloc(sizeof(int) * N);
  int* row = (int*)malloc(sizeof(int) * (N + 1));
  int* col = (int*)malloc(sizeof(int) * N * 7);
  float* val = (float*)malloc(sizeof(float) * N * 7);
  float* valp = (float*)malloc(sizeof(float) * N * 7);
  float* vec = (float*)malloc(sizeof(float) * N * 8);
  float max_res = -INFINITY;
  int b, status = 0;
  memset(p, 0, sizeof(float) * N * B);
  for (b = 0; b < B && status == 0; b++) {
    const float* fb = flags + b * N;
    const float* db = div + b * N;
    float* pb = p + b * N;
    int ncomp = 0, c, i, j, k;
    long n;
    int* sizes;
    /* findConnectedFluidComponents: scan order, depth-first stack */
    for (n = 0; n < N; n++) comp[n] = -1;
    sizes = (int*)calloc((size_t)N + 1, sizeof(int));
    for (n = 0; n < N; n++) {
      int sp = 0;
      if (comp[n] != -1 || !(((int)fb[n]) & F_FLUID)) continue;
      stack[sp++] = (int)n;
      while (sp > 0) {
        const int cur = stack[--sp];
        const int ci = cur % X, cj = (cur / X) % Y, ck = cur / (X * Y);
        const int nb[6][3] = {{ci - 1, cj, ck}, {ci + 1, cj, ck}, {ci, cj - 1, ck}, {ci, cj + 1, ck}, {ci, cj, ck - 1}, {ci, cj, ck + 1}};
        int q;
        comp[cur] = ncomp; sizes[ncomp]++;
        for (q = 0; q < (is3d ? 6 : 4); q++) {
          const int xi = nb[q][0], yj = nb[q][1], zk = nb[q][2];
          long m;
          if (xi < 0 || xi >= X || yj < 0 || yj >= Y || zk < 0 || zk >= Z) continue;
          m = AT(d, xi, yj, zk);
          if ((((int)fb[m]) & F_FLUID) && comp[m] == -1) { comp[m] = -2; stack[sp++] = (int)m; }
        }
      }
      ncomp++;
    }
    for (c = 0; c < ncomp && status == 0; c++) {
      long numel = 0, nz = 0, r;
      int pc = precond, iter = 0, upper;
      float *rhs, *x, *rv, *z, *s, *w, *y, *tmp;
      float rr1, rr0 = 0.0f, num, den, alpha, beta, prev_num = 0.0f, mean;
      if (sizes[c] == 1) continue;
      if (sizes[c] < 5) pc = 0;
      upper = (pc == 2);
      for (n = 0; n < N; n++) sysidx[n] = (comp[n] == c) ? (int)numel++ : -1;
      /* setupLaplacian */
      row[0] = 0; r = 0;
      for (k = 0; k < Z && status == 0; k++)
        for (j = 0; j < Y && status == 0; j++)
          for (i = 0; i < X; i++) {
            const long o = AT(d, i, j, k);
            float diag = 0.0f;
            if (comp[o] != c) continue;
            if (on_border(d, i, j, k) || !is_fluid(d, fb, i, j, k)) { status = -1; break; }
            if (!is_obst(d, fb, i - 1, j, k)) diag += 1;
            if (!is_obst(d, fb, i + 1, j, k)) diag += 1;
            if (!is_obst(d, fb, i, j - 1, k)) diag += 1;
            if (!is_obst(d, fb, i, j + 1, k)) diag += 1;
            if (is3d && !is_obst(d, fb, i, j, k - 1)) diag += 1;
            if (is3d && !is_obst(d, fb, i, j, k + 1)) diag += 1;
            if (is3d && !upper && is_fluid(d, fb, i, j, k - 1)) { val[nz] = -1.0f; col[nz++] = sysidx[AT(d, i, j, k - 1)]; }
            if (!upper && is_fluid(d, fb, i, j - 1, k)) { val[nz] = -1.0f; col[nz++] = sysidx[AT(d, i, j - 1, k)]; }
            if (!upper && is_fluid(d, fb, i - 1, j, k)) { val[nz] = -1.0f; col[nz++] = sysidx[AT(d, i - 1, j, k)]; }
            val[nz] = diag; col[nz++] = sysidx[o];
            if (is_fluid(d, fb, i + 1, j, k)) { val[nz] = -1.0f; col[nz++] = sysidx[AT(d, i + 1, j, k)]; }
            if (is_fluid(d, fb, i, j + 1, k)) { val[nz] = -1.0f; col[nz++] = sysidx[AT(d, i, j + 1, k)]; }
            if (is3d && is_fluid(d, fb, i, j, k + 1)) { val[nz] = -1.0f; col[nz++] = sysidx[AT(d, i, j, k + 1)]; }
            row[++r] = (int)nz;
          }
      if (status) break;
      rhs = vec; x = vec + numel; rv = vec + 2 * numel; z = vec + 3 * numel; s = vec + 4 * numel; w = vec + 5 * numel;
      y = vec + 6 * numel; tmp = vec + 7 * numel; (void)tmp;
      for (n = 0; n < N; n++) if (sysidx[n] >= 0) rhs[sysidx[n]] = db[n];
      memcpy(valp, val, sizeof(float) * nz);
      if (pc == 1) csr_ilu0(numel, row, col, valp);
      if (pc == 2) csr_ic0_upper(numel, row, col, valp);
      for (r = 0; r < numel; r++) { x[r] = 0.0f; rv[r] = rhs[r]; }
      rr1 = sdot(numel, rv, rv);
      if (rr1 != rr1) { status = -2; break; }
      while (rr1 > tol * tol && iter <= max_iter) {
        if (pc == 1) { csr_lsolve_unit(numel, row, col, valp, rv, y); csr_usolve(numel, row, col, valp, y, z); }
        else if (pc == 2) { csr_utsolve(numel, row, col, valp, rv, y); csr_usolve(numel, row, col, valp, y, z); }
        iter++;
        if (iter == 1) {
          memcpy(s, pc ? z : rv, sizeof(float) * numel);
        } else if (pc) {
          num = sdot(numel, rv, z);
          beta = num / clamp_to_eps(prev_num);
          for (r = 0; r < numel; r++) s[r] = beta * s[r] + z[r];
        } else {
          beta = rr1 / clamp_to_eps(rr0);
          for (r = 0; r < numel; r++) s[r] = beta * s[r] + rv[r];
        }
        csrmv(numel, row, col, val, upper, s, w);
        num = pc ? sdot(numel, rv, z) : rr1;
        den = sdot(numel, s, w);
        alpha = num / clamp_to_eps(den);
        for (r = 0; r < numel; r++) x[r] = alpha * s[r] + x[r];
        prev_num = num;                       /* rm2.zm2 of the next iteration */
        for (r = 0; r < numel; r++) rv[r] = -alpha * w[r] + rv[r];
        rr0 = rr1;
        rr1 = sdot(numel, rv, rv);
        if (rr1 != rr1) { status = -2; break; }
      }
      if (status) break;
      if (sqrtf(rr1) > max_res) max_res = sqrtf(rr1);
      { double acc = 0.0; for (r = 0; r < numel; r++) acc += x[r]; mean = (float)(acc / (double)numel); }
      for (n = 0; n < N; n++) if (sysidx[n] >= 0) pb[n] = x[sysidx[n]] - mean;
    }
    free(sizes);
  }
  free(comp); free(sysidx); free(stack); free(row); free(col); free(val); free(valp); free(vec);
  if (out_residual) *out_residual = max_res;
  return status;
}

/* ------------------------------------------------------------------------------------------
 * normalizePressureMean, generic/tfluids.cc:845-925: per batch element, flood-fill the fluid components
 * (generic/find_connected_fluid_components.cc) and subtract each component's mean pressure. The reference
 * accumulates the mean with `#pragma omp atomic` float adds (order not fixed); here it is a double sum
 * rounded once, so agreement with the compiled reference is to rounding, not bitwise.
 * ---------------------------------------------------------------------------------------- */
void ora_normalizePressureMean(float* p, const float* flags, int is3d, int B, int Z, int Y, int X) {
  const long N = (long)X * Y * Z;
  int* comp = (int*)malloc(sizeof(int) * N);
  int* stack = (int*)malloc(sizeof(int) * N);
  double* sum = (double*)malloc(sizeof(double) * (N + 1));
  int* cnt = (int*)malloc(sizeof(int) * (N + 1));
  int b;
  for (b = 0; b < B; b++) {
    const float* fb = flags + b * N;
    float* pb = p + b * N;
    long n;
    int ncomp = 0;
    for (n = 0; n < N; n++) comp[n] = -1;
    for (n = 0; n < N; n++) {
      int sp = 0;
      if (comp[n] != -1 || !(((int)fb[n]) & F_FLUID)) continue;
      stack[sp++] = (int)n; sum[ncomp] = 0.0; cnt[ncomp] = 0;
      while (sp > 0) {
        const int cur = stack[--sp];
        const int ci = cur % X, cj = (cur / X) % Y, ck = cur / (X * Y);
        const int nb[6][3] = {{ci - 1, cj, ck}, {ci + 1, cj, ck}, {ci, cj - 1, ck}, {ci, cj + 1, ck}, {ci, cj, ck - 1}, {ci, cj, ck + 1}};
        int q;
        comp[cur] = ncomp; cnt[ncomp]++; sum[ncomp] += (double)pb[cur];
        for (q = 0; q < (is3d ? 6 : 4); q++) {
          const int xi = nb[q][0], yj = nb[q][1], zk = nb[q][2];
          long m;
          if (xi < 0 || xi >= X || yj < 0 || yj >= Y || zk < 0 || zk >= Z) continue;
          m = (long)xi + (long)yj * X + (long)zk * X * Y;
          if ((((int)fb[m]) & F_FLUID) && comp[m] == -1) { comp[m] = -2; stack[sp++] = (int)m; }
        }
      }
      ncomp++;
    }
    for (n = 0; n < N; n++)
      if (comp[n] >= 0) pb[n] = pb[n] - (float)(sum[comp[n]] / (double)cnt[comp[n]]);
  }
  free(comp); free(stack); free(sum); free(cnt);
}
