/*
 * splat_oracle.c — CPU restatement of the 2D-Gaussian-splat hot path of GS-SDF.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product path (gs-sdf_amd/) never links it.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in the reference's un-vendored
 * submodule jianhengLiu/gsplat_cpp (a fork of nerfstudio gsplat's 2DGS kernels, pinned SHA
 * unknown: /root/reference/.gitmodules:1-18, directory empty) and the reference ships no
 * tests or golden vectors.  This file therefore restates the published algorithm (2DGS,
 * Huang et al. SIGGRAPH'24; gsplat >= 1.4 `fully_fused_projection_2dgs`,
 * `isect_tiles`/`isect_offset_encode`, `rasterize_to_pixels_2dgs`) anchored on the
 * reference's own call sites:
 *     include/neural_gaussian/neural_gaussian.cpp:188-192  (projection, 9 outputs)
 *     include/neural_gaussian/neural_gaussian.cpp:199-200  (view colours / SH)
 *     include/neural_gaussian/neural_gaussian.cpp:207-209  (tile_encode)
 *     include/neural_gaussian/neural_gaussian.cpp:215-223  (rasterize, 7 outputs)
 *     include/neural_gaussian/neural_gaussian.cpp:626-633  (consumers of densify grads)
 * Fork-specific outputs whose definition cannot be recovered (`samples`, `samples_weights`,
 * `render_depths`, `visibilities`) follow the decisions frozen in DESIGN.md section "SPEC".
 * The oracle is validated by itself (tests/test_oracle_selfcheck.py): f64-vs-f32 agreement,
 * finite differences and an independent torch-autograd restatement of every VJP.
 *
 * The file is compiled twice: -DREAL=float  -> liborc_splat_f32.so (bit-exact integer parity,
 * CPU baseline) and -DREAL=double -> liborc_splat_f64.so (gradient checking).
 * Build flags must keep IEEE semantics: -O2 -ffp-contract=off, no -ffast-math.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#ifdef REAL_IS_FLOAT
#define R_SQRT(x) sqrtf(x)
#define R_CEIL(x) ceilf(x)
#define R_FLOOR(x) floorf(x)
#else
#define R_SQRT(x) sqrt(x)
#define R_CEIL(x) ceil(x)
#define R_FLOOR(x) floor(x)
#endif

#define TILE_ALPHA_MIN ((REAL)(1.0 / 255.0))
#define ALPHA_MAX ((REAL)0.999)
#define T_EPS ((REAL)1e-4)
#define FILTER_INV_SQUARE ((REAL)2.0)

static inline REAL rmax(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL rmin(REAL a, REAL b) { return a < b ? a : b; }

/* ---------------------------------------------------------------------------------------
 * counter-based normal pair used for the fork's stochastic `samples` (SPEC S-3)
 * ------------------------------------------------------------------------------------- */
static inline uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
static void sample_eps(uint64_t seed, uint32_t gid, REAL *eu, REAL *ev) {
  if (seed == 0) { *eu = 0; *ev = 0; return; }
  uint32_t h1 = mix32(gid * 2u + 0x9E3779B9u * (uint32_t)seed + (uint32_t)(seed >> 32));
  uint32_t h2 = mix32(h1 ^ 0x68E31DA4u);
  double u1 = ((double)(h1 >> 8) + 1.0) / 16777216.0; /* (0,1] */
  double u2 = (double)(h2 >> 8) / 16777216.0;          /* [0,1) */
  double r = sqrt(-2.0 * log(u1));
  *eu = (REAL)(r * cos(6.283185307179586 * u2));
  *ev = (REAL)(r * sin(6.283185307179586 * u2));
}

/* quaternion (w,x,y,z) -> rotation, reference formula include/utils/utils.cpp:538-558,
 * after normalisation (gsplat normalises inside the kernel). */
static void quat_to_rot(const REAL *q, REAL *qn, REAL *inv_norm, REAL R[9]) {
  REAL w = q[0], x = q[1], y = q[2], z = q[3];
  REAL n2 = ((w * w + x * x) + y * y) + z * z;
  REAL inv = (REAL)1 / R_SQRT(n2);
  w *= inv; x *= inv; y *= inv; z *= inv;
  qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z; *inv_norm = inv;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

typedef struct {
  int culled;
  REAL mc[3];        /* camera-space centre */
  REAL Rq[9];        /* splat rotation (world) */
  REAL Rc[9];        /* R_view * Rq */
  REAL qn[4], inv_norm;
  REAL Mu[3], Mv[3], Mw[3];
  REAL f[3];         /* (1,1,-1)/distance */
  REAL mean2d[2];
  REAL radius;
  REAL mult;         /* normal flip */
} proj_t;

/* SPEC A.1 (gsplat fully_fused_projection_fwd_2dgs; reference call neural_gaussian.cpp:190) */
static void project_one(const REAL *mean, const REAL *quat, const REAL *scale, const REAL *vm,
                        const REAL *K, int W, int H, REAL near_p, REAL far_p, REAL radius_clip,
                        proj_t *o) {
  o->culled = 1;
  const REAL R00 = vm[0], R01 = vm[1], R02 = vm[2], t0 = vm[3];
  const REAL R10 = vm[4], R11 = vm[5], R12 = vm[6], t1 = vm[7];
  const REAL R20 = vm[8], R21 = vm[9], R22 = vm[10], t2 = vm[11];
  o->mc[0] = ((R00 * mean[0] + R01 * mean[1]) + R02 * mean[2]) + t0;
  o->mc[1] = ((R10 * mean[0] + R11 * mean[1]) + R12 * mean[2]) + t1;
  o->mc[2] = ((R20 * mean[0] + R21 * mean[1]) + R22 * mean[2]) + t2;
  if (o->mc[2] < near_p || o->mc[2] > far_p) return;
  quat_to_rot(quat, o->qn, &o->inv_norm, o->Rq);
  const REAL *q = o->Rq;
  REAL *c = o->Rc;
  for (int j = 0; j < 3; ++j) {
    c[0 + j] = (R00 * q[0 + j] + R01 * q[3 + j]) + R02 * q[6 + j];
    c[3 + j] = (R10 * q[0 + j] + R11 * q[3 + j]) + R12 * q[6 + j];
    c[6 + j] = (R20 * q[0 + j] + R21 * q[3 + j]) + R22 * q[6 + j];
  }
  /* the normal flip is decided here, BEFORE the radius / screen culls: the backward recomputes this function for rows that were visible in the
   * forward, and in another precision such a row may fail a cull by a rounding — its mult must still be defined (round 5: it was read
   * uninitialised by the f64 backward for splats whose box touches the screen edge) */
  {
    REAL dotv = (-c[2]) * o->mc[0] + (-c[5]) * o->mc[1] + (-c[8]) * o->mc[2];
    o->mult = dotv > 0 ? (REAL)1 : (REAL)-1;
  }
  const REAL su = scale[0], sv = scale[1];
  /* H rows: H[i] = (su*Rc[i][0], sv*Rc[i][1], mc[i]) */
  REAL H0[3] = {su * c[0], sv * c[1], o->mc[0]};
  REAL H1[3] = {su * c[3], sv * c[4], o->mc[1]};
  REAL H2[3] = {su * c[6], sv * c[7], o->mc[2]};
  const REAL fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  for (int j = 0; j < 3; ++j) {
    o->Mu[j] = fx * H0[j] + cx * H2[j];
    o->Mv[j] = fy * H1[j] + cy * H2[j];
    o->Mw[j] = H2[j];
  }
  const REAL *Mu = o->Mu, *Mv = o->Mv, *Mw = o->Mw;
  REAL dist = (Mw[0] * Mw[0] + Mw[1] * Mw[1]) - Mw[2] * Mw[2];
  if (dist == 0) return;
  REAL inv = (REAL)1 / dist;
  o->f[0] = inv; o->f[1] = inv; o->f[2] = -inv;
  const REAL *f = o->f;
  o->mean2d[0] = (f[0] * Mu[0] * Mw[0] + f[1] * Mu[1] * Mw[1]) + f[2] * Mu[2] * Mw[2];
  o->mean2d[1] = (f[0] * Mv[0] * Mw[0] + f[1] * Mv[1] * Mw[1]) + f[2] * Mv[2] * Mw[2];
  REAL tx = (f[0] * Mu[0] * Mu[0] + f[1] * Mu[1] * Mu[1]) + f[2] * Mu[2] * Mu[2];
  REAL ty = (f[0] * Mv[0] * Mv[0] + f[1] * Mv[1] * Mv[1]) + f[2] * Mv[2] * Mv[2];
  REAL hx = o->mean2d[0] * o->mean2d[0] - tx;
  REAL hy = o->mean2d[1] * o->mean2d[1] - ty;
  REAL ext = rmax((REAL)1e-4, rmax(hx, hy));
  REAL radius = R_CEIL((REAL)3 * R_SQRT(ext));
  if (!(radius > radius_clip)) return; /* also drops NaN */
  if (o->mean2d[0] + radius <= 0 || o->mean2d[0] - radius >= (REAL)W ||
      o->mean2d[1] + radius <= 0 || o->mean2d[1] - radius >= (REAL)H)
    return;
  if (!(radius < (REAL)2147483000.0)) return; /* int32 overflow guard (inf) */
  o->radius = radius;
  o->culled = 0;
}

/* Packed projection forward.  Outputs have capacity C*N rows; returns M (nnz).
 * Row order: increasing (camera, gaussian).  */
int64_t orc_projection_2dgs_fwd(int64_t N, int64_t C, const REAL *means, const REAL *quats,
                                const REAL *scales, const REAL *viewmats, const REAL *Ks, int W,
                                int H, REAL near_p, REAL far_p, REAL radius_clip, uint64_t seed,
                                int64_t *camera_ids, int64_t *gaussian_ids, int32_t *radii,
                                REAL *means2d, REAL *depths, REAL *ray_transforms, REAL *normals,
                                REAL *samples, REAL *samples_weights) {
  /* Two passes so that the (camera, gaussian) pairs can be projected on all cores (the cpu_baseline leg times this on the host's cores):
   * pass 1 marks the survivors, a serial running sum gives every survivor its row (the serial loop's order: camera-major, gaussian
   * ascending), pass 2 projects the survivors again and writes their rows.  Per pair the arithmetic is the serial loop's. */
  const int64_t P = C * N;
  unsigned char *keep = (unsigned char *)malloc((size_t)(P > 0 ? P : 1));
  int64_t *row = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P + 1));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < P; ++i) {
    const int64_t c = i / N, n = i - c * N;
    proj_t p;
    project_one(means + 3 * n, quats + 4 * n, scales + 3 * n, viewmats + 16 * c, Ks + 9 * c, W, H, near_p, far_p, radius_clip, &p);
    keep[i] = p.culled ? 0 : 1;
  }
  row[0] = 0;
  for (int64_t i = 0; i < P; ++i) row[i + 1] = row[i] + keep[i];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < P; ++i) {
    if (!keep[i]) continue;
    const int64_t c = i / N, n = i - c * N, m = row[i];
    proj_t p;
    project_one(means + 3 * n, quats + 4 * n, scales + 3 * n, viewmats + 16 * c, Ks + 9 * c, W, H, near_p, far_p, radius_clip, &p);
    camera_ids[m] = c; gaussian_ids[m] = n; radii[m] = (int32_t)p.radius;
    means2d[2 * m] = p.mean2d[0]; means2d[2 * m + 1] = p.mean2d[1];
    depths[m] = p.mc[2];
    for (int j = 0; j < 3; ++j) {
      ray_transforms[9 * m + j] = p.Mu[j];
      ray_transforms[9 * m + 3 + j] = p.Mv[j];
      ray_transforms[9 * m + 6 + j] = p.Mw[j];
      normals[3 * m + j] = p.mult * p.Rc[3 * j + 2];
    }
    REAL eu, ev;
    sample_eps(seed, (uint32_t)n, &eu, &ev);
    for (int j = 0; j < 3; ++j)
      samples[3 * m + j] = means[3 * n + j] + (scales[3 * n] * eu) * p.Rq[3 * j] +
                           (scales[3 * n + 1] * ev) * p.Rq[3 * j + 1];
    samples_weights[m] = (REAL)exp(-0.5 * (double)(eu * eu + ev * ev));
  }
  const int64_t total = row[P];
  free(keep); free(row);
  return total;
}

/* SPEC A.6: VJP of the projection.  Dense [N,.] outputs (sparse_grad=false,
 * neural_gaussian.cpp:529), accumulated over cameras. Outputs must be zeroed by the caller. */
static void projection_bwd_row(int64_t m, const REAL *means, const REAL *quats,
                             const REAL *scales, const REAL *viewmats, const REAL *Ks, int W, int H,
                             uint64_t seed, const int64_t *camera_ids, const int64_t *gaussian_ids,
                             const REAL *v_means2d, const REAL *v_depths,
                             const REAL *v_ray_transforms, const REAL *v_normals,
                             const REAL *v_samples, REAL *v_means, REAL *v_quats, REAL *v_scales) {
    int64_t c = camera_ids[m], n = gaussian_ids[m];
    const REAL *vm = viewmats + 16 * c, *K = Ks + 9 * c;
    proj_t p;
    project_one(means + 3 * n, quats + 4 * n, scales + 3 * n, vm, K, W, H, (REAL)-1e30, (REAL)1e30,
                (REAL)-1, &p);
    /* recompute without culling; dist==0 rows never reach here */
    const REAL *Mu = p.Mu, *Mv = p.Mv, *Mw = p.Mw, *f = p.f;
    REAL vMu[3], vMv[3], vMw[3];
    for (int j = 0; j < 3; ++j) {
      vMu[j] = v_ray_transforms[9 * m + j];
      vMv[j] = v_ray_transforms[9 * m + 3 + j];
      vMw[j] = v_ray_transforms[9 * m + 6 + j];
    }
    REAL gx = v_means2d[2 * m], gy = v_means2d[2 * m + 1];
    for (int j = 0; j < 3; ++j) {
      vMu[j] += gx * f[j] * Mw[j];
      vMv[j] += gy * f[j] * Mw[j];
      vMw[j] += gx * (f[j] * Mu[j] - 2 * f[j] * Mw[j] * p.mean2d[0]) +
                gy * (f[j] * Mv[j] - 2 * f[j] * Mw[j] * p.mean2d[1]);
    }
    const REAL fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    REAL vH0[3], vH1[3], vH2[3];
    for (int j = 0; j < 3; ++j) {
      vH0[j] = fx * vMu[j];
      vH1[j] = fy * vMv[j];
      vH2[j] = cx * vMu[j] + cy * vMv[j] + vMw[j];
    }
    const REAL su = scales[3 * n], sv = scales[3 * n + 1];
    const REAL *c9 = p.Rc;
    REAL v_mc[3] = {vH0[2], vH1[2], vH2[2] + v_depths[m]};
    REAL v_su = vH0[0] * c9[0] + vH1[0] * c9[3] + vH2[0] * c9[6];
    REAL v_sv = vH0[1] * c9[1] + vH1[1] * c9[4] + vH2[1] * c9[7];
    REAL vRc[9];
    vRc[0] = su * vH0[0]; vRc[3] = su * vH1[0]; vRc[6] = su * vH2[0];
    vRc[1] = sv * vH0[1]; vRc[4] = sv * vH1[1]; vRc[7] = sv * vH2[1];
    vRc[2] = p.mult * v_normals[3 * m]; vRc[5] = p.mult * v_normals[3 * m + 1];
    vRc[8] = p.mult * v_normals[3 * m + 2];
    /* Rc = Rv * Rq  ->  vRq = Rv^T vRc ; mc = Rv mu + t -> v_mu = Rv^T v_mc */
    REAL Rv[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
    REAL vRq[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        vRq[3 * i + j] = Rv[0 + i] * vRc[0 + j] + Rv[3 + i] * vRc[3 + j] + Rv[6 + i] * vRc[6 + j];
    REAL v_mu[3];
    for (int i = 0; i < 3; ++i) v_mu[i] = Rv[0 + i] * v_mc[0] + Rv[3 + i] * v_mc[1] + Rv[6 + i] * v_mc[2];
    /* samples = mu + su*eu*Rq[:,0] + sv*ev*Rq[:,1] */
    REAL eu, ev;
    sample_eps(seed, (uint32_t)n, &eu, &ev);
    if (v_samples) {
      const REAL *vs = v_samples + 3 * m;
      for (int i = 0; i < 3; ++i) {
        v_mu[i] += vs[i];
        vRq[3 * i + 0] += su * eu * vs[i];
        vRq[3 * i + 1] += sv * ev * vs[i];
        v_su += eu * p.Rq[3 * i + 0] * vs[i];
        v_sv += ev * p.Rq[3 * i + 1] * vs[i];
      }
    }
    /* rotation -> normalised quaternion */
    REAL w = p.qn[0], x = p.qn[1], y = p.qn[2], z = p.qn[3];
    const REAL *g = vRq;
    REAL vq[4];
    vq[0] = 2 * (x * (g[7] - g[5]) + y * (g[2] - g[6]) + z * (g[3] - g[1]));
    vq[1] = 2 * (-2 * x * (g[4] + g[8]) + y * (g[1] + g[3]) + z * (g[2] + g[6]) + w * (g[7] - g[5]));
    vq[2] = 2 * (x * (g[1] + g[3]) - 2 * y * (g[0] + g[8]) + z * (g[5] + g[7]) + w * (g[2] - g[6]));
    vq[3] = 2 * (x * (g[2] + g[6]) + y * (g[5] + g[7]) - 2 * z * (g[0] + g[4]) + w * (g[3] - g[1]));
    REAL dotq = vq[0] * w + vq[1] * x + vq[2] * y + vq[3] * z;
    for (int i = 0; i < 4; ++i) v_quats[4 * n + i] += (vq[i] - dotq * p.qn[i]) * p.inv_norm;
    for (int i = 0; i < 3; ++i) v_means[3 * n + i] += v_mu[i];
    v_scales[3 * n] += v_su;
    v_scales[3 * n + 1] += v_sv;
}

void orc_projection_2dgs_bwd(int64_t N, int64_t C, int64_t M, const REAL *means, const REAL *quats,
                             const REAL *scales, const REAL *viewmats, const REAL *Ks, int W, int H,
                             uint64_t seed, const int64_t *camera_ids, const int64_t *gaussian_ids,
                             const REAL *v_means2d, const REAL *v_depths,
                             const REAL *v_ray_transforms, const REAL *v_normals,
                             const REAL *v_samples, REAL *v_means, REAL *v_quats, REAL *v_scales) {
  (void)N; (void)C;
  /* rows of one camera name distinct gaussians (the forward's output order: camera-major, gaussian ascending): they are independent and run on
   * all cores, camera after camera, so that a gaussian's sum over the cameras is accumulated in the serial loop's order.  Any other row
   * order takes the serial loop. */
  int ordered = 1;
  for (int64_t m = 1; m < M && ordered; ++m)
    ordered = camera_ids[m] > camera_ids[m - 1] || (camera_ids[m] == camera_ids[m - 1] && gaussian_ids[m] > gaussian_ids[m - 1]);
  if (!ordered) {
    for (int64_t m = 0; m < M; ++m)
      projection_bwd_row(m, means, quats, scales, viewmats, Ks, W, H, seed, camera_ids, gaussian_ids, v_means2d, v_depths, v_ray_transforms,
                         v_normals, v_samples, v_means, v_quats, v_scales);
    return;
  }
  for (int64_t a = 0; a < M;) {
    int64_t b = a;
    while (b < M && camera_ids[b] == camera_ids[a]) ++b;
#pragma omp parallel for schedule(static)
    for (int64_t m = a; m < b; ++m)
      projection_bwd_row(m, means, quats, scales, viewmats, Ks, W, H, seed, camera_ids, gaussian_ids, v_means2d, v_depths, v_ray_transforms,
                         v_normals, v_samples, v_means, v_quats, v_scales);
    a = b;
  }
}

/* ---------------------------------------------------------------------------------------
 * P2: view-dependent colours (gsplat_cpp::get_view_colors, neural_gaussian.cpp:199-200)
 * SPEC A.2: rgb = max(SH(dir) . coeffs + 0.5, 0), dir = normalise(mean - campos)
 * ------------------------------------------------------------------------------------- */
static void cam_pos(const REAL *vm, REAL cp[3]) {
  /* campos = inverse(viewmat)[:3,3] = -A^{-1} t  (A = upper 3x3, adjugate inverse: the view
   * matrix is fp32 and only approximately orthonormal, so A^T is not used) */
  REAL a = vm[0], b = vm[1], c = vm[2], d = vm[4], e = vm[5], f = vm[6], g = vm[8], h = vm[9], i = vm[10];
  REAL A00 = e * i - f * h, A01 = c * h - b * i, A02 = b * f - c * e;
  REAL A10 = f * g - d * i, A11 = a * i - c * g, A12 = c * d - a * f;
  REAL A20 = d * h - e * g, A21 = b * g - a * h, A22 = a * e - b * d;
  REAL det = a * A00 + b * A10 + c * A20;
  REAL id = (REAL)1 / det;
  REAL t0 = vm[3], t1 = vm[7], t2 = vm[11];
  cp[0] = -((A00 * t0 + A01 * t1) + A02 * t2) * id;
  cp[1] = -((A10 * t0 + A11 * t1) + A12 * t2) * id;
  cp[2] = -((A20 * t0 + A21 * t1) + A22 * t2) * id;
}

#define SH_C0 0.28209479177387814
#define SH_C1 0.4886025119029199
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554,  -0.4570457994644658,
                                0.3731763325901154,  -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};

static void sh_basis(int deg, REAL x, REAL y, REAL z, REAL *b) {
  b[0] = (REAL)SH_C0;
  if (deg < 1) return;
  b[1] = (REAL)(-SH_C1) * y; b[2] = (REAL)SH_C1 * z; b[3] = (REAL)(-SH_C1) * x;
  if (deg < 2) return;
  REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  b[4] = (REAL)SH_C2[0] * xy; b[5] = (REAL)SH_C2[1] * yz; b[6] = (REAL)SH_C2[2] * (2 * zz - xx - yy);
  b[7] = (REAL)SH_C2[3] * xz; b[8] = (REAL)SH_C2[4] * (xx - yy);
  if (deg < 3) return;
  b[9] = (REAL)SH_C3[0] * y * (3 * xx - yy); b[10] = (REAL)SH_C3[1] * xy * z;
  b[11] = (REAL)SH_C3[2] * y * (4 * zz - xx - yy);
  b[12] = (REAL)SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
  b[13] = (REAL)SH_C3[4] * x * (4 * zz - xx - yy); b[14] = (REAL)SH_C3[5] * z * (xx - yy);
  b[15] = (REAL)SH_C3[6] * x * (xx - 3 * yy);
}
/* d basis / d (x,y,z), each 16x3 */
static void sh_basis_grad(int deg, REAL x, REAL y, REAL z, REAL db[16][3]) {
  memset(db, 0, sizeof(REAL) * 48);
  if (deg < 1) return;
  db[1][1] = (REAL)(-SH_C1); db[2][2] = (REAL)SH_C1; db[3][0] = (REAL)(-SH_C1);
  if (deg < 2) return;
  REAL xx = x * x, yy = y * y, zz = z * z;
  db[4][0] = (REAL)SH_C2[0] * y; db[4][1] = (REAL)SH_C2[0] * x;
  db[5][1] = (REAL)SH_C2[1] * z; db[5][2] = (REAL)SH_C2[1] * y;
  db[6][0] = (REAL)SH_C2[2] * (-2 * x); db[6][1] = (REAL)SH_C2[2] * (-2 * y); db[6][2] = (REAL)SH_C2[2] * (4 * z);
  db[7][0] = (REAL)SH_C2[3] * z; db[7][2] = (REAL)SH_C2[3] * x;
  db[8][0] = (REAL)SH_C2[4] * (2 * x); db[8][1] = (REAL)SH_C2[4] * (-2 * y);
  if (deg < 3) return;
  db[9][0] = (REAL)SH_C3[0] * 6 * x * y; db[9][1] = (REAL)SH_C3[0] * (3 * xx - 3 * yy);
  db[10][0] = (REAL)SH_C3[1] * y * z; db[10][1] = (REAL)SH_C3[1] * x * z; db[10][2] = (REAL)SH_C3[1] * x * y;
  db[11][0] = (REAL)SH_C3[2] * (-2 * x * y); db[11][1] = (REAL)SH_C3[2] * (4 * zz - xx - 3 * yy);
  db[11][2] = (REAL)SH_C3[2] * 8 * y * z;
  db[12][0] = (REAL)SH_C3[3] * (-6 * x * z); db[12][1] = (REAL)SH_C3[3] * (-6 * y * z);
  db[12][2] = (REAL)SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
  db[13][0] = (REAL)SH_C3[4] * (4 * zz - 3 * xx - yy); db[13][1] = (REAL)SH_C3[4] * (-2 * x * y);
  db[13][2] = (REAL)SH_C3[4] * 8 * x * z;
  db[14][0] = (REAL)SH_C3[5] * 2 * x * z; db[14][1] = (REAL)SH_C3[5] * (-2 * y * z);
  db[14][2] = (REAL)SH_C3[5] * (xx - yy);
  db[15][0] = (REAL)SH_C3[6] * (3 * xx - 3 * yy); db[15][1] = (REAL)SH_C3[6] * (-6 * x * y);
}

/* sh_coeffs: [N,K,3]; out colors [M,3] */
void orc_view_colors_fwd(int64_t M, int64_t K, int sh_degree, const REAL *viewmats, const REAL *means,
                         const REAL *sh_coeffs, const int64_t *camera_ids, const int64_t *gaussian_ids,
                         REAL *colors) {
  for (int64_t m = 0; m < M; ++m) {
    int64_t n = gaussian_ids[m];
    REAL cp[3];
    cam_pos(viewmats + 16 * camera_ids[m], cp);
    REAL d[3] = {means[3 * n] - cp[0], means[3 * n + 1] - cp[1], means[3 * n + 2] - cp[2]};
    REAL len = (REAL)sqrt((double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
    REAL inv = len > 0 ? 1 / len : 0;
    REAL b[16];
    sh_basis(sh_degree, d[0] * inv, d[1] * inv, d[2] * inv, b);
    int nb = (sh_degree + 1) * (sh_degree + 1);
    for (int ch = 0; ch < 3; ++ch) {
      REAL acc = 0;
      for (int k = 0; k < nb; ++k) acc += b[k] * sh_coeffs[(n * K + k) * 3 + ch];
      colors[3 * m + ch] = rmax(acc + (REAL)0.5, 0);
    }
  }
}

/* outputs v_sh [N,K,3] and v_means [N,3] must be zeroed by the caller */
/* First-order bound of the fp32 evaluation error of orc_projection_2dgs_bwd: the SAME expression tree evaluated on absolute values with every
 * subtraction turned into an addition ("absolute shadow"), i.e. the sum of the absolute values of all terms that enter each output — internal
 * cancellations (the quaternion gradient's projection orthogonal to q, the mean2d terms of v_Mw) included.  An fp32 evaluation in ANY operation
 * order deviates from the exact value by at most (operation depth) x eps32 x this sum.  Outputs b_* are accumulated like the gradients.
 * `a_*` = |upstream gradient| or its own error bound (the parity tests push both through). */
void orc_projection_2dgs_bwd_bound(int64_t N, int64_t C, int64_t M, const REAL *means, const REAL *quats,
                                   const REAL *scales, const REAL *viewmats, const REAL *Ks, int W, int H,
                                   uint64_t seed, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                   const REAL *a_means2d, const REAL *a_depths, const REAL *a_ray_transforms, const REAL *a_normals,
                                   const REAL *a_samples, REAL *b_means, REAL *b_quats, REAL *b_scales) {
  (void)N; (void)C;
#define AB(x) ((REAL)fabs((double)(x)))
  for (int64_t m = 0; m < M; ++m) {
    int64_t c = camera_ids[m], n = gaussian_ids[m];
    const REAL *vm = viewmats + 16 * c, *K = Ks + 9 * c;
    proj_t p;
    project_one(means + 3 * n, quats + 4 * n, scales + 3 * n, vm, K, W, H, (REAL)-1e30, (REAL)1e30, (REAL)-1, &p);
    const REAL *Mu = p.Mu, *Mv = p.Mv, *Mw = p.Mw, *f = p.f;
    REAL vMu[3], vMv[3], vMw[3];
    for (int j = 0; j < 3; ++j) {
      vMu[j] = AB(a_ray_transforms[9 * m + j]);
      vMv[j] = AB(a_ray_transforms[9 * m + 3 + j]);
      vMw[j] = AB(a_ray_transforms[9 * m + 6 + j]);
    }
    REAL gx = AB(a_means2d[2 * m]), gy = AB(a_means2d[2 * m + 1]);
    /* f = (1,1,-1) / dist, dist = Mw0^2 + Mw1^2 - Mw2^2: a cancelling sum, relative error kd x eps32 with kd = sum Mw_j^2 / |dist| (>= 1) —
     * the recomputed FORWARD quantity every mean2d term carries; mean2d = sum_j f_j Mu_j Mw_j inherits it on the sum of its |terms| */
    const REAL sq = Mw[0] * Mw[0] + Mw[1] * Mw[1] + Mw[2] * Mw[2];
    const REAL kd = sq * AB(f[0]);
    REAL am0 = 0, am1 = 0;
    for (int j = 0; j < 3; ++j) { am0 += AB(f[j] * Mu[j] * Mw[j]); am1 += AB(f[j] * Mv[j] * Mw[j]); }
    for (int j = 0; j < 3; ++j) {
      vMu[j] += gx * AB(f[j] * Mw[j]) * kd;
      vMv[j] += gy * AB(f[j] * Mw[j]) * kd;
      vMw[j] += gx * (AB(f[j] * Mu[j]) * kd + 2 * AB(f[j] * Mw[j]) * am0 * (2 * kd)) + gy * (AB(f[j] * Mv[j]) * kd + 2 * AB(f[j] * Mw[j]) * am1 * (2 * kd));
    }
    const REAL fx = AB(K[0]), cx = AB(K[2]), fy = AB(K[4]), cy = AB(K[5]);
    REAL vH0[3], vH1[3], vH2[3];
    for (int j = 0; j < 3; ++j) {
      vH0[j] = fx * vMu[j];
      vH1[j] = fy * vMv[j];
      vH2[j] = cx * vMu[j] + cy * vMv[j] + vMw[j];
    }
    const REAL su = AB(scales[3 * n]), sv = AB(scales[3 * n + 1]);
    const REAL *c9 = p.Rc;
    REAL v_mc[3] = {vH0[2], vH1[2], vH2[2] + (a_depths ? AB(a_depths[m]) : 0)};
    REAL v_su = vH0[0] * AB(c9[0]) + vH1[0] * AB(c9[3]) + vH2[0] * AB(c9[6]);
    REAL v_sv = vH0[1] * AB(c9[1]) + vH1[1] * AB(c9[4]) + vH2[1] * AB(c9[7]);
    REAL vRc[9];
    vRc[0] = su * vH0[0]; vRc[3] = su * vH1[0]; vRc[6] = su * vH2[0];
    vRc[1] = sv * vH0[1]; vRc[4] = sv * vH1[1]; vRc[7] = sv * vH2[1];
    vRc[2] = AB(p.mult) * AB(a_normals[3 * m]); vRc[5] = AB(p.mult) * AB(a_normals[3 * m + 1]); vRc[8] = AB(p.mult) * AB(a_normals[3 * m + 2]);
    REAL Rv[9] = {AB(vm[0]), AB(vm[1]), AB(vm[2]), AB(vm[4]), AB(vm[5]), AB(vm[6]), AB(vm[8]), AB(vm[9]), AB(vm[10])};
    REAL vRq[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) vRq[3 * i + j] = Rv[0 + i] * vRc[0 + j] + Rv[3 + i] * vRc[3 + j] + Rv[6 + i] * vRc[6 + j];
    REAL v_mu[3];
    for (int i = 0; i < 3; ++i) v_mu[i] = Rv[0 + i] * v_mc[0] + Rv[3 + i] * v_mc[1] + Rv[6 + i] * v_mc[2];
    REAL eu, ev;
    sample_eps(seed, (uint32_t)n, &eu, &ev);
    if (a_samples) {
      const REAL *vs = a_samples + 3 * m;
      for (int i = 0; i < 3; ++i) {
        v_mu[i] += AB(vs[i]);
        vRq[3 * i + 0] += su * AB(eu) * AB(vs[i]);
        vRq[3 * i + 1] += sv * AB(ev) * AB(vs[i]);
        v_su += AB(eu * p.Rq[3 * i + 0]) * AB(vs[i]);
        v_sv += AB(ev * p.Rq[3 * i + 1]) * AB(vs[i]);
      }
    }
    REAL w = AB(p.qn[0]), x = AB(p.qn[1]), y = AB(p.qn[2]), z = AB(p.qn[3]);
    const REAL *g = vRq;
    REAL vq[4];
    vq[0] = 2 * (x * (g[7] + g[5]) + y * (g[2] + g[6]) + z * (g[3] + g[1]));
    vq[1] = 2 * (2 * x * (g[4] + g[8]) + y * (g[1] + g[3]) + z * (g[2] + g[6]) + w * (g[7] + g[5]));
    vq[2] = 2 * (x * (g[1] + g[3]) + 2 * y * (g[0] + g[8]) + z * (g[5] + g[7]) + w * (g[2] + g[6]));
    vq[3] = 2 * (x * (g[2] + g[6]) + y * (g[5] + g[7]) + 2 * z * (g[0] + g[4]) + w * (g[3] + g[1]));
    REAL dotq = vq[0] * w + vq[1] * x + vq[2] * y + vq[3] * z;
    const REAL qa[4] = {w, x, y, z};
    for (int i = 0; i < 4; ++i) b_quats[4 * n + i] += (vq[i] + dotq * qa[i]) * AB(p.inv_norm);
    for (int i = 0; i < 3; ++i) b_means[3 * n + i] += v_mu[i];
    b_scales[3 * n] += v_su;
    b_scales[3 * n + 1] += v_sv;
  }
#undef AB
}

void orc_view_colors_bwd(int64_t M, int64_t K, int sh_degree, const REAL *viewmats, const REAL *means,
                         const REAL *sh_coeffs, const int64_t *camera_ids, const int64_t *gaussian_ids,
                         const REAL *v_colors, REAL *v_sh, REAL *v_means) {
  for (int64_t m = 0; m < M; ++m) {
    int64_t n = gaussian_ids[m];
    REAL cp[3];
    cam_pos(viewmats + 16 * camera_ids[m], cp);
    REAL d[3] = {means[3 * n] - cp[0], means[3 * n + 1] - cp[1], means[3 * n + 2] - cp[2]};
    REAL len = (REAL)sqrt((double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
    REAL inv = len > 0 ? 1 / len : 0;
    REAL u[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
    REAL b[16], db[16][3];
    sh_basis(sh_degree, u[0], u[1], u[2], b);
    sh_basis_grad(sh_degree, u[0], u[1], u[2], db);
    int nb = (sh_degree + 1) * (sh_degree + 1);
    REAL vu[3] = {0, 0, 0};
    for (int ch = 0; ch < 3; ++ch) {
      REAL acc = 0;
      for (int k = 0; k < nb; ++k) acc += b[k] * sh_coeffs[(n * K + k) * 3 + ch];
      if (!(acc + (REAL)0.5 > 0)) continue; /* clamp_min gate */
      REAL g = v_colors[3 * m + ch];
      for (int k = 0; k < nb; ++k) {
        v_sh[(n * K + k) * 3 + ch] += g * b[k];
        REAL cf = g * sh_coeffs[(n * K + k) * 3 + ch];
        vu[0] += cf * db[k][0]; vu[1] += cf * db[k][1]; vu[2] += cf * db[k][2];
      }
    }
    /* u = d/|d| :  v_d = (v_u - (v_u.u) u)/|d| */
    REAL dot = vu[0] * u[0] + vu[1] * u[1] + vu[2] * u[2];
    for (int i = 0; i < 3; ++i) v_means[3 * n + i] += (vu[i] - dot * u[i]) * inv;
  }
}

/* absolute shadow of orc_view_colors_bwd (see orc_projection_2dgs_bwd_bound) */
void orc_view_colors_bwd_bound(int64_t M, int64_t K, int sh_degree, const REAL *viewmats, const REAL *means,
                               const REAL *sh_coeffs, const int64_t *camera_ids, const int64_t *gaussian_ids,
                               const REAL *a_colors, REAL *b_sh, REAL *b_means) {
#define AB(x) ((REAL)fabs((double)(x)))
  for (int64_t m = 0; m < M; ++m) {
    int64_t n = gaussian_ids[m];
    REAL cp[3];
    cam_pos(viewmats + 16 * camera_ids[m], cp);
    REAL d[3] = {means[3 * n] - cp[0], means[3 * n + 1] - cp[1], means[3 * n + 2] - cp[2]};
    REAL len = (REAL)sqrt((double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
    REAL inv = len > 0 ? 1 / len : 0;
    REAL u[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
    REAL b[16], db[16][3];
    sh_basis(sh_degree, u[0], u[1], u[2], b);
    sh_basis_grad(sh_degree, u[0], u[1], u[2], db);
    int nb = (sh_degree + 1) * (sh_degree + 1);
    REAL vu[3] = {0, 0, 0};
    for (int ch = 0; ch < 3; ++ch) {
      REAL acc = 0;
      for (int k = 0; k < nb; ++k) acc += b[k] * sh_coeffs[(n * K + k) * 3 + ch];
      if (!(acc + (REAL)0.5 > 0)) continue;
      REAL g = AB(a_colors[3 * m + ch]);
      for (int k = 0; k < nb; ++k) {
        b_sh[(n * K + k) * 3 + ch] += g * AB(b[k]);
        REAL cf = g * AB(sh_coeffs[(n * K + k) * 3 + ch]);
        vu[0] += cf * AB(db[k][0]); vu[1] += cf * AB(db[k][1]); vu[2] += cf * AB(db[k][2]);
      }
    }
    REAL dot = vu[0] * AB(u[0]) + vu[1] * AB(u[1]) + vu[2] * AB(u[2]);
    for (int i = 0; i < 3; ++i) b_means[3 * n + i] += (vu[i] + dot * AB(u[i])) * inv;
  }
#undef AB
}

/* ---------------------------------------------------------------------------------------
 * P3: tile binning (gsplat_cpp::tile_encode, neural_gaussian.cpp:207-209).  SPEC A.3.
 * Integer outputs: the bit-exact parity target.
 * ------------------------------------------------------------------------------------- */
static void tile_rect(const REAL *mean2d, int32_t radius, int tile_size, int tw, int th, int *x0,
                      int *y0, int *x1, int *y1) {
  REAL r = (REAL)radius / (REAL)tile_size;
  REAL tx = mean2d[0] / (REAL)tile_size, ty = mean2d[1] / (REAL)tile_size;
  REAL a;
  a = R_FLOOR(tx - r); *x0 = (int)rmin(rmax(a, 0), (REAL)tw);
  a = R_CEIL(tx + r);  *x1 = (int)rmin(rmax(a, 0), (REAL)tw);
  a = R_FLOOR(ty - r); *y0 = (int)rmin(rmax(a, 0), (REAL)th);
  a = R_CEIL(ty + r);  *y1 = (int)rmin(rmax(a, 0), (REAL)th);
}

int64_t orc_tile_count(int64_t M, int W, int H, int tile_size, const REAL *means2d,
                       const int32_t *radii, int32_t *tiles_per_gauss) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t total = 0;
  for (int64_t m = 0; m < M; ++m) {
    int32_t cnt = 0;
    if (radii[m] > 0) {
      int x0, y0, x1, y1;
      tile_rect(means2d + 2 * m, radii[m], tile_size, tw, th, &x0, &y0, &x1, &y1);
      cnt = (y1 - y0) * (x1 - x0);
    }
    tiles_per_gauss[m] = cnt;
    total += cnt;
  }
  return total;
}

static int tile_bits_for(int64_t n_tiles) {
  int b = 0;
  while ((1LL << (b + 1)) <= n_tiles) ++b; /* floor(log2) */
  return b + 1;
}

/* LSD radix sort, 8 bits per pass, stable; sorts (key,val) on bits [0,nbits) */
static void radix_sort_pairs(uint64_t *keys, int32_t *vals, int64_t n, int nbits) {
  uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
  int32_t *v2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int shift = 0; shift < nbits; shift += 8) {
    int64_t hist[257];
    memset(hist, 0, sizeof(hist));
    for (int64_t i = 0; i < n; ++i) hist[((keys[i] >> shift) & 0xFF) + 1]++;
    for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
    for (int64_t i = 0; i < n; ++i) {
      int64_t d = hist[(keys[i] >> shift) & 0xFF]++;
      k2[d] = keys[i]; v2[d] = vals[i];
    }
    memcpy(keys, k2, sizeof(uint64_t) * (size_t)n);
    memcpy(vals, v2, sizeof(int32_t) * (size_t)n);
  }
  free(k2); free(v2);
}

/* emits keys, sorts, builds offsets.  isect_ids [I] (sorted keys), flatten_ids [I],
 * isect_offsets [C*th*tw]. depths are always passed as float32 bits (depth_bits). */
void orc_tile_encode(int64_t M, int64_t C, int W, int H, int tile_size, const REAL *means2d,
                     const int32_t *radii, const uint32_t *depth_bits, const int64_t *camera_ids,
                     int64_t I, int64_t *isect_ids, int32_t *flatten_ids, int32_t *isect_offsets) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t n_tiles = (int64_t)tw * th;
  int tbits = tile_bits_for(n_tiles);
  int cbits = tile_bits_for(C);
  uint64_t *keys = (uint64_t *)isect_ids;
  int64_t pos = 0;
  for (int64_t m = 0; m < M; ++m) {
    if (radii[m] <= 0) continue;
    int x0, y0, x1, y1;
    tile_rect(means2d + 2 * m, radii[m], tile_size, tw, th, &x0, &y0, &x1, &y1);
    uint64_t cid = (uint64_t)camera_ids[m];
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        uint64_t tile_id = (uint64_t)y * tw + x;
        keys[pos] = (cid << (32 + tbits)) | (tile_id << 32) | (uint64_t)depth_bits[m];
        flatten_ids[pos] = (int32_t)m;
        ++pos;
      }
  }
  (void)I;
  radix_sort_pairs(keys, flatten_ids, pos, 32 + tbits + cbits);
  /* offsets: first sorted position whose (cam,tile) >= this tile */
  int64_t total_tiles = C * n_tiles;
  int64_t cur = 0;
  for (int64_t t = 0; t < total_tiles; ++t) {
    int64_t c = t / n_tiles, tl = t % n_tiles;
    uint64_t tk = ((uint64_t)c << tbits) | (uint64_t)tl;
    while (cur < pos && (keys[cur] >> 32) < tk) ++cur;
    isect_offsets[t] = (int32_t)cur;
  }
}

/* ---------------------------------------------------------------------------------------
 * P4: compositing forward (rasterize_to_pixels_2dgs, neural_gaussian.cpp:215-223). SPEC A.4.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int valid;      /* contributes (alpha >= 1/255) */
  int branch3d;   /* g3 <= g2 */
  int clamped;    /* opac*vis > 0.999 */
  REAL hu[3], hv[3], z[3], s[2], d[2], vis, alpha, dep;
} pix_eval_t;

static inline void eval_pair(REAL px, REAL py, const REAL *xy, REAL opac, const REAL *M,
                             pix_eval_t *e) {
  const REAL *Mu = M, *Mv = M + 3, *Mw = M + 6;
  e->valid = 0;
  for (int j = 0; j < 3; ++j) { e->hu[j] = px * Mw[j] - Mu[j]; e->hv[j] = py * Mw[j] - Mv[j]; }
  e->z[0] = e->hu[1] * e->hv[2] - e->hu[2] * e->hv[1];
  e->z[1] = e->hu[2] * e->hv[0] - e->hu[0] * e->hv[2];
  e->z[2] = e->hu[0] * e->hv[1] - e->hu[1] * e->hv[0];
  if (e->z[2] == 0) return;
  e->s[0] = e->z[0] / e->z[2]; e->s[1] = e->z[1] / e->z[2];
  REAL g3 = e->s[0] * e->s[0] + e->s[1] * e->s[1];
  e->d[0] = xy[0] - px; e->d[1] = xy[1] - py;
  REAL g2 = FILTER_INV_SQUARE * (e->d[0] * e->d[0] + e->d[1] * e->d[1]);
  e->branch3d = g3 <= g2;
  REAL sigma = (REAL)0.5 * (e->branch3d ? g3 : g2);
  e->vis = (REAL)exp(-(double)sigma);
  REAL a = opac * e->vis;
  e->clamped = a > ALPHA_MAX;
  e->alpha = rmin(ALPHA_MAX, a);
  if (!(sigma >= 0) || !(e->alpha >= TILE_ALPHA_MIN)) return;
  e->dep = e->branch3d ? (e->s[0] * Mw[0] + e->s[1] * Mw[1]) + Mw[2] : Mw[2];
  e->valid = 1;
}

/* backgrounds: [C,3] or NULL; masks: uint8 [C,th,tw] or NULL (0 => tile skipped: bg only) */
void orc_rasterize_2dgs_fwd(int64_t C, int64_t M, int64_t I, int W, int H, int tile_size,
                            const REAL *means2d, const REAL *ray_transforms, const REAL *colors,
                            const REAL *opacities, const REAL *normals, const REAL *backgrounds,
                            const uint8_t *masks, const int32_t *isect_offsets,
                            const int32_t *flatten_ids, REAL *render_colors, REAL *render_depths,
                            REAL *render_alphas, REAL *render_normals, REAL *render_median,
                            int32_t *last_ids, int32_t *median_ids, REAL *visibilities) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t n_tiles = (int64_t)tw * th;
  for (int64_t m = 0; m < M; ++m) visibilities[m] = 0;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t t = 0; t < C * n_tiles; ++t) {
    int64_t c = t / n_tiles, tl = t % n_tiles;
    int ty = (int)(tl / tw), tx = (int)(tl % tw);
    int32_t start = isect_offsets[t];
    int32_t end = (t == C * n_tiles - 1) ? (int32_t)I : isect_offsets[t + 1];
    int masked = masks && !masks[t];
    int32_t len = masked ? 0 : end - start;
    REAL *vis_local = (REAL *)calloc((size_t)(len > 0 ? len : 1), sizeof(REAL));
    for (int yy = 0; yy < tile_size; ++yy)
      for (int xx = 0; xx < tile_size; ++xx) {
        int i = ty * tile_size + yy, j = tx * tile_size + xx;
        if (i >= H || j >= W) continue;
        int64_t pid = (c * H + i) * (int64_t)W + j;
        REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5;
        REAL T = 1, col[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, dsum = 0, med = 0;
        int32_t cur = 0, med_idx = 0; /* gsplat initialises both to 0 */
        for (int32_t k = 0; k < len; ++k) {
          int32_t g = flatten_ids[start + k];
          pix_eval_t e;
          eval_pair(px, py, means2d + 2 * g, opacities[g], ray_transforms + 9 * g, &e);
          if (!e.valid) continue;
          REAL nT = T * (1 - e.alpha);
          if (nT <= T_EPS) break;
          REAL w = e.alpha * T;
          for (int ch = 0; ch < 3; ++ch) {
            col[ch] += colors[3 * g + ch] * w;
            nrm[ch] += normals[3 * g + ch] * w;
          }
          dsum += e.dep * w;
          if (T > (REAL)0.5) { med = e.dep; med_idx = start + k; }
          if (w > vis_local[k]) vis_local[k] = w;
          cur = start + k;
          T = nT;
        }
        for (int ch = 0; ch < 3; ++ch) {
          render_colors[3 * pid + ch] = backgrounds ? col[ch] + T * backgrounds[3 * c + ch] : col[ch];
          render_normals[3 * pid + ch] = nrm[ch];
        }
        render_depths[pid] = dsum;
        render_alphas[pid] = 1 - T;
        render_median[pid] = med;
        last_ids[pid] = cur;
        median_ids[pid] = med_idx;
      }
#pragma omp critical
    for (int32_t k = 0; k < len; ++k) {
      int32_t g = flatten_ids[start + k];
      if (vis_local[k] > visibilities[g]) visibilities[g] = vis_local[k];
    }
    free(vis_local);
  }
}

/* debugging aid (tools/): per-pixel (pid, v_sigma, v_dep, alpha, T, branch) of ONE splat during the next backward call */
static int64_t dbg_gid = -1, dbg_cap = 0, dbg_n = 0;
static double *dbg_buf = 0;
void orc_debug_trace_splat(int64_t gid, int64_t cap, double *buf) { dbg_gid = gid; dbg_cap = cap; dbg_buf = buf; dbg_n = 0; }
int64_t orc_debug_trace_count(void) { return dbg_n; }

/* SPEC A.5: VJP of the compositing.  Gradient buffers [M,.] must be zeroed by the caller.
 * v_means2d_abs may be NULL.  Gradient outputs are double regardless of REAL (deterministic
 * per-tile accumulation, cross-tile merge in double). */
void orc_rasterize_2dgs_bwd(int64_t C, int64_t M, int64_t I, int W, int H, int tile_size,
                            const REAL *means2d, const REAL *ray_transforms, const REAL *colors,
                            const REAL *opacities, const REAL *normals, const REAL *backgrounds,
                            const uint8_t *masks, const int32_t *isect_offsets,
                            const int32_t *flatten_ids, const REAL *render_alphas,
                            const int32_t *last_ids, const int32_t *median_ids,
                            const REAL *v_render_colors, const REAL *v_render_depths,
                            const REAL *v_render_alphas, const REAL *v_render_normals,
                            const REAL *v_render_median, double *v_means2d,
                            double *v_ray_transforms, double *v_colors, double *v_opacities,
                            double *v_normals, double *v_densify, double *v_means2d_abs,
                            double *abs_ray_transforms /* [M,9] or NULL: sum of |per-pixel contribution| */,
                            double *abs_densify /* [M,2] or NULL */) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t n_tiles = (int64_t)tw * th;
  (void)M;
  enum { NG = 33 }; /* per-splat accumulators: xy2 M9 col3 opac1 nrm3 dens2 abs2 | |M|9 |dens|2 (conditioning of the sums) */
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t t = 0; t < C * n_tiles; ++t) {
    int64_t c = t / n_tiles, tl = t % n_tiles;
    int ty = (int)(tl / tw), tx = (int)(tl % tw);
    int32_t start = isect_offsets[t];
    int32_t end = (t == C * n_tiles - 1) ? (int32_t)I : isect_offsets[t + 1];
    if (masks && !masks[t]) continue;
    int32_t len = end - start;
    if (len <= 0) continue;
    double *acc = (double *)calloc((size_t)len * NG, sizeof(double));
    for (int yy = 0; yy < tile_size; ++yy)
      for (int xx = 0; xx < tile_size; ++xx) {
        int i = ty * tile_size + yy, j = tx * tile_size + xx;
        if (i >= H || j >= W) continue;
        int64_t pid = (c * H + i) * (int64_t)W + j;
        REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5;
        const REAL T_final = 1 - render_alphas[pid];
        REAL T = T_final;
        const int32_t bin_final = last_ids[pid], med_idx = median_ids[pid];
        const REAL *vC = v_render_colors + 3 * pid, *vN = v_render_normals + 3 * pid;
        const REAL vD = v_render_depths[pid], vA = v_render_alphas[pid], vMed = v_render_median[pid];
        REAL bufC[3] = {0, 0, 0}, bufN[3] = {0, 0, 0}, bufD = 0;
        REAL bgdot = 0;
        if (backgrounds)
          bgdot = backgrounds[3 * c] * vC[0] + backgrounds[3 * c + 1] * vC[1] + backgrounds[3 * c + 2] * vC[2];
        for (int32_t idx = bin_final; idx >= start; --idx) {
          int32_t g = flatten_ids[idx];
          pix_eval_t e;
          const REAL *Mrow = ray_transforms + 9 * g;
          eval_pair(px, py, means2d + 2 * g, opacities[g], Mrow, &e);
          if (!e.valid) continue;
          double *a = acc + (size_t)(idx - start) * NG;
          const REAL ra = 1 / (1 - e.alpha);
          T *= ra;
          const REAL fac = e.alpha * T;
          REAL v_alpha = 0;
          for (int ch = 0; ch < 3; ++ch) {
            a[11 + ch] += fac * vC[ch];
            a[15 + ch] += fac * vN[ch];
            v_alpha += (colors[3 * g + ch] * T - bufC[ch] * ra) * vC[ch];
            v_alpha += (normals[3 * g + ch] * T - bufN[ch] * ra) * vN[ch];
          }
          v_alpha += (e.dep * T - bufD * ra) * vD;
          v_alpha += T_final * ra * vA;
          v_alpha += -T_final * ra * bgdot;
          REAL v_dep = fac * vD + (idx == med_idx ? vMed : 0);
          for (int ch = 0; ch < 3; ++ch) {
            bufC[ch] += colors[3 * g + ch] * fac;
            bufN[ch] += normals[3 * g + ch] * fac;
          }
          bufD += e.dep * fac;
          REAL v_sigma = 0;
          if (!e.clamped) {
            a[14] += e.vis * v_alpha;                 /* v_opacity */
            v_sigma = -opacities[g] * e.vis * v_alpha; /* d alpha / d sigma */
          }
          const REAL *Mw = Mrow + 6;
          if (g == dbg_gid && dbg_buf) {
#pragma omp critical(dbgtrace)
            if (dbg_n < dbg_cap) {
              double *r = dbg_buf + 6 * dbg_n++;
              r[0] = (double)pid; r[1] = (double)v_sigma; r[2] = (double)v_dep; r[3] = (double)e.alpha; r[4] = (double)T; r[5] = e.branch3d;
            }
          }
          if (e.branch3d) {
            /* sigma = 0.5 (s.s); dep = s.x Mw.x + s.y Mw.y + Mw.z */
            REAL v_s[2] = {v_sigma * e.s[0] + v_dep * Mw[0], v_sigma * e.s[1] + v_dep * Mw[1]};
            REAL vsx = v_s[0] / e.z[2], vsy = v_s[1] / e.z[2];
            REAL v_z[3] = {vsx, vsy, -(vsx * e.s[0] + vsy * e.s[1])};
            /* z = hu x hv */
            REAL v_hu[3] = {e.hv[1] * v_z[2] - e.hv[2] * v_z[1], e.hv[2] * v_z[0] - e.hv[0] * v_z[2],
                            e.hv[0] * v_z[1] - e.hv[1] * v_z[0]};
            REAL v_hv[3] = {v_z[1] * e.hu[2] - v_z[2] * e.hu[1], v_z[2] * e.hu[0] - v_z[0] * e.hu[2],
                            v_z[0] * e.hu[1] - v_z[1] * e.hu[0]};
            REAL vMw[3] = {px * v_hu[0] + py * v_hv[0] + v_dep * e.s[0],
                           px * v_hu[1] + py * v_hv[1] + v_dep * e.s[1],
                           px * v_hu[2] + py * v_hv[2] + v_dep};
            for (int k = 0; k < 3; ++k) {
              a[2 + k] += -v_hu[k];
              a[5 + k] += -v_hv[k];
              a[8 + k] += vMw[k];
            }
            {
              /* conditioning of the same sums: every product that enters a difference, in absolute value (the cross
               * products h x v_z cancel per pixel when h and v_z are close to parallel, v_z.z = -(v_s . s) cancels too) */
              const double za[3] = {fabs((double)v_z[0]), fabs((double)v_z[1]), fabs((double)(vsx * e.s[0])) + fabs((double)(vsy * e.s[1]))};
              for (int k = 0; k < 3; ++k) {
                const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                const double ahu = fabs((double)e.hv[k1]) * za[k2] + fabs((double)e.hv[k2]) * za[k1];
                const double ahv = fabs((double)e.hu[k1]) * za[k2] + fabs((double)e.hu[k2]) * za[k1];
                a[22 + k] += ahu;
                a[25 + k] += ahv;
                a[28 + k] += fabs((double)px) * ahu + fabs((double)py) * ahv + fabs((double)(v_dep * (k < 2 ? e.s[k] : 1)));
                if (k == 2) { a[31] += ahu * fabs((double)Mw[2]); a[32] += ahv * fabs((double)Mw[2]); }
              }
            }
            /* densification signal (2DGS: dL/dM[2]*depth, dL/dM[5]*depth) SPEC S-4 */
            a[18] += -v_hu[2] * Mw[2];
            a[19] += -v_hv[2] * Mw[2];

          } else {
            /* sigma = 0.5*2*(d.d) ; dep = Mw.z */
            REAL gx = v_sigma * FILTER_INV_SQUARE * e.d[0], gy = v_sigma * FILTER_INV_SQUARE * e.d[1];
            a[0] += gx; a[1] += gy;
            a[20] += fabs((double)gx); a[21] += fabs((double)gy);
            a[10] += v_dep;
            a[30] += fabs((double)v_dep);
          }
        }
      }
#pragma omp critical
    for (int32_t k = 0; k < len; ++k) {
      int32_t g = flatten_ids[start + k];
      const double *a = acc + (size_t)k * NG;
      v_means2d[2 * g] += a[0]; v_means2d[2 * g + 1] += a[1];
      for (int q = 0; q < 9; ++q) v_ray_transforms[9 * g + q] += a[2 + q];
      for (int q = 0; q < 3; ++q) { v_colors[3 * g + q] += a[11 + q]; v_normals[3 * g + q] += a[15 + q]; }
      v_opacities[g] += a[14];
      v_densify[2 * g] += a[18]; v_densify[2 * g + 1] += a[19];
      if (v_means2d_abs) { v_means2d_abs[2 * g] += a[20]; v_means2d_abs[2 * g + 1] += a[21]; }
      if (abs_ray_transforms) for (int q = 0; q < 9; ++q) abs_ray_transforms[9 * g + q] += a[22 + q];
      if (abs_densify) { abs_densify[2 * g] += a[31]; abs_densify[2 * g + 1] += a[32]; }
    }
    free(acc);
  }
}


/* ---------------------------------------------------------------------------------------
 * P4 fragility analysis (test infrastructure for the decision-matched parity gate).
 *
 * The compositing operator (SPEC A.4 / A.5) is piecewise smooth: per (pixel, splat) pair it takes the decisions
 *   alpha >= 1/255 (contribute), T (1 - alpha) <= 1e-4 (stop), T > 0.5 (median), g3 <= g2 (which footprint, which
 *   depth), opac * vis > 0.999 (clamp: gates d alpha).
 * Two correct fp32 evaluations of the same inputs may take a different side of a decision whose margin is inside
 * their rounding error, and then differ by O(alpha T) in that pixel and in the gradient of every splat the pixel
 * blends.  This function walks every pixel's list with the build's own (f64: exact) decisions, evaluates every pair a
 * second time in plain fp32 (the direct h_u x h_v form), and flags a pair as FRAGILE when a decision margin is
 * within `kmargin` x the observed fp32 evaluation error of the quantity that is compared (with a floor of
 * `ulp_floor` relative), or as ILL-CONDITIONED when the fp32 evaluation of its blending weight deviates by more than
 * `cond_abs` (absolute, T-weighted).  pix_flags[p] = OR over the pixel's pairs; splat_flags[g] = OR of the flags of
 * every pixel in which splat g is blended (or is itself the fragile pair).  Elements with flag 0 are the ones on
 * which a 1e-4 element-wise comparison is meaningful; the caller reports the excluded fraction.
 * A splat is additionally flagged EDGE-ON (64, splat only) when, in a pixel that blends it, the z.z component of
 * h_u x h_v = h_u.x h_v.y - h_u.y h_v.x cancels by more than `kappa_max` (kappa = (|h_u.x h_v.y| + |h_u.y h_v.x|) / |z.z|):
 * every fp32 evaluation of s = z.xy / z.z — the reference's included — then carries ~kappa x 6e-8 relative error, amplified
 * again by the 1 / z.z^2 of the gradient; its own gradient is not comparable at 1e-4 element-wise.
 * bits: 1 alpha test, 2 termination, 4 median, 8 footprint branch, 16 clamp, 32 ill-conditioned weight, 64 edge-on splat.
 * ------------------------------------------------------------------------------------- */
typedef struct { float a, g3, g2; int zero; } pair32_t;
static inline pair32_t eval_pair_f32(float px, float py, float mx, float my, float opac, const float *M) {
  pair32_t r; r.zero = 0; r.a = 0; r.g3 = 0; r.g2 = 0;
  float hu[3], hv[3];
  /* fused multiply-adds where a GPU compiler contracts them (nvcc -fmad=true, hipcc -ffp-contract=fast): h = p M_w - M is the
   * cancelling step (|p M_w.z| ~ |M.z| ~ 1e3 x |h.z|), exactly rounded with an FMA */
  for (int j = 0; j < 3; ++j) { hu[j] = fmaf(px, M[6 + j], -M[j]); hv[j] = fmaf(py, M[6 + j], -M[3 + j]); }
  float zx = fmaf(hu[1], hv[2], -(hu[2] * hv[1])), zy = fmaf(hu[2], hv[0], -(hu[0] * hv[2])), zz = fmaf(hu[0], hv[1], -(hu[1] * hv[0]));
  if (zz == 0.0f) { r.zero = 1; return r; }
  float sx = zx / zz, sy = zy / zz;
  r.g3 = sx * sx + sy * sy;
  float dx = mx - px, dy = my - py;
  r.g2 = 2.0f * (dx * dx + dy * dy);
  float sigma = 0.5f * (r.g3 <= r.g2 ? r.g3 : r.g2);
  r.a = opac * expf(-sigma);
  return r;
}

void orc_rasterize_2dgs_fragility(int64_t C, int64_t M, int64_t I, int W, int H, int tile_size,
                                  const REAL *means2d, const REAL *ray_transforms, const REAL *opacities,
                                  const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                  double kmargin, double ulp_floor, double cond_abs, double kappa_max,
                                  uint8_t *pix_flags, uint8_t *splat_flags, int64_t *counts /* [8] */) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t n_tiles = (int64_t)tw * th;
  for (int64_t m = 0; m < M; ++m) splat_flags[m] = 0;
  for (int q = 0; q < 8; ++q) counts[q] = 0;
  int64_t n_pairs = 0, n_valid = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_pairs, n_valid)
  for (int64_t t = 0; t < C * n_tiles; ++t) {
    int64_t c = t / n_tiles, tl = t % n_tiles;
    int ty = (int)(tl / tw), tx = (int)(tl % tw);
    int32_t start = isect_offsets[t];
    int32_t end = (t == C * n_tiles - 1) ? (int32_t)I : isect_offsets[t + 1];
    int masked = masks && !masks[t];
    int32_t len = masked ? 0 : end - start;
    uint8_t *sf = (uint8_t *)calloc((size_t)(len > 0 ? len : 1), 1);
    uint8_t *touched = (uint8_t *)malloc((size_t)(len > 0 ? len : 1));
    for (int yy = 0; yy < tile_size; ++yy)
      for (int xx = 0; xx < tile_size; ++xx) {
        int i = ty * tile_size + yy, j = tx * tile_size + xx;
        if (i >= H || j >= W) continue;
        int64_t pid = (c * H + i) * (int64_t)W + j;
        REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5;
        double T = 1, eT = 0;   /* transmittance and a running bound of its fp32 evaluation error */
        uint8_t pf = 0;
        int32_t prev_valid = -1;
        memset(touched, 0, (size_t)(len > 0 ? len : 1));
        for (int32_t k = 0; k < len; ++k) {
          int32_t g = flatten_ids[start + k];
          pix_eval_t e;
          eval_pair(px, py, means2d + 2 * g, opacities[g], ray_transforms + 9 * g, &e);
          float Mf[9];
          for (int q = 0; q < 9; ++q) Mf[q] = (float)ray_transforms[9 * g + q];
          pair32_t f = eval_pair_f32((float)px, (float)py, (float)means2d[2 * g], (float)means2d[2 * g + 1], (float)opacities[g], Mf);
          ++n_pairs;
          if (e.z[2] == 0 || f.zero) { if (e.z[2] != 0 || !f.zero) { pf |= 32; touched[k] = 1; } continue; }
          double a64 = (double)opacities[g] * (double)e.vis;          /* before the clamp */
          double err_a = fabs((double)f.a - a64) + ulp_floor * a64;
          double g3 = (double)(e.s[0] * e.s[0] + e.s[1] * e.s[1]);
          double g2 = 2.0 * (double)(e.d[0] * e.d[0] + e.d[1] * e.d[1]);
          double err_g = fabs((double)f.g3 - g3) + fabs((double)f.g2 - g2) + ulp_floor * (g3 > g2 ? g3 : g2);
          double alpha = a64 < 0.999 ? a64 : 0.999;
          uint8_t fl = 0;
          if (fabs(alpha - 1.0 / 255.0) <= kmargin * err_a) fl |= 1;
          int contributes = e.valid;
          if (contributes || (fl & 1)) {
            if (fabs(g3 - g2) <= kmargin * err_g) fl |= 8;
            if (fabs(a64 - 0.999) <= kmargin * err_a) fl |= 16;
            if (T * err_a > cond_abs + T * ulp_floor * a64 * 4) fl |= 32;
          }
          if (contributes) {
            double nT = T * (1 - alpha);
            double enT = eT * (1 - alpha) + T * err_a + ulp_floor * nT;
            if (fabs(nT - 1e-4) <= kmargin * enT) fl |= 2;
            if (nT <= 1e-4) { if (fl) { pf |= fl; touched[k] = 1; } break; }
            if (fabs(T - 0.5) <= kmargin * (eT + ulp_floor * T)) {
              /* median = the last pair blended while T > 0.5: a flip moves render_median (and its upstream gradient)
               * between this pair and the previous blended one; nothing else in the pixel changes */
              pf |= 4; sf[k] |= 4;
              if (prev_valid >= 0) sf[prev_valid] |= 4;
            }
            ++n_valid;
            touched[k] = 1;
            prev_valid = k;
            {
              const double t1 = fabs((double)(e.hu[0] * e.hv[1])), t2 = fabs((double)(e.hu[1] * e.hv[0]));
              if (t1 + t2 > kappa_max * fabs((double)e.z[2])) sf[k] |= 64;
            }
            T = nT; eT = enT;
          }
          if (fl) { pf |= fl; touched[k] = 1; }
        }
        pix_flags[pid] = pf;
        if (pf & ~4)
          for (int32_t k = 0; k < len; ++k)
            if (touched[k]) sf[k] |= (uint8_t)(pf & ~4);
      }
#pragma omp critical
    for (int32_t k = 0; k < len; ++k)
      if (sf[k]) splat_flags[flatten_ids[start + k]] |= sf[k];
    free(sf); free(touched);
  }
  counts[0] = n_pairs; counts[1] = n_valid;
}

int orc_real_bytes(void) { return (int)sizeof(REAL); }

/* ---------------------------------------------------------------------------------------
 * P4 DECISION-MATCHED compositing (test infrastructure; the gate of tests/util.py).
 *
 * orc_rasterize_2dgs_fragility() above names the pixels in which some decision of the operator has a margin inside fp32
 * rounding.  Instead of leaving those pixels (and the splats they blend) unchecked, the implementation under test is asked for
 * the decisions it actually took there (libgsdf_hip's gsdf_rasterize_2dgs_fwd_instr writes one byte per (traced pixel, list
 * position): bit0 blended, bit1 3-D footprint branch, bit2 alpha clamped, bit3 the pixel terminates at this pair (not blended),
 * bit4 the median is updated here) and the fp64 operator is evaluated UNDER THOSE DECISIONS: between two decision surfaces the
 * operator is smooth, so the result is the exact value the implementation approximates and the 1e-4 bar applies to it.
 * Every decision that differs from the fp64 evaluation's own is counted and its margin is reported in units of the fp32
 * evaluation error of the compared quantity (flip_worst): a flip is legitimate only inside that noise, the caller asserts it.
 *
 * What is left after matching the decisions is CONDITIONING, which no implementation escapes: exp(-|s|^2/2) with s = z.xy / z.z
 * carries kappa = (|h_u.x h_v.y| + |h_u.y h_v.x|) / |z.z| times the rounding of z.z (edge-on splats), transmittances are
 * products of (1 - alpha) with alpha up to 0.999, per-splat gradients are sums over pixels of terms that cancel.  Both
 * functions therefore return, next to every value, a first-order BOUND of its fp32 evaluation error in units of eps32
 * (pix_bound / vis_bound / cond): per pair the relative error of alpha is r = 2 + 2 sigma kappa' (kappa' = kappa in the 3-D
 * branch, 1 in the screen-space branch, r = 0 when clamped), a pixel's weights carry R = sum_j r_j max(1, alpha_j/(1-alpha_j))
 * + (number of blended pairs), s and 1/z.z add kappa each, and every sum is bounded through the sum of the absolute values of
 * its terms.  The gate is  |got - ref| <= 1e-4 max(|ref|, mean|ref|) + C eps32 bound  for EVERY element.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int valid, stop, median, forced;
  int branch3d, clamped;
  REAL hu[3], hv[3], z[3], s[2], d[2], vis, alpha, dep, sigma, a_raw, g3, g2;
  double kappa, ks, r;   /* conditioning of the pair: z.z, the vector s (relative, in norm), alpha — see the block comment */
} mpair_t;

static inline void eval_pair_matched(REAL px, REAL py, const REAL *xy, REAL opac, const REAL *M, int forced, uint8_t bits, mpair_t *e) {
  const REAL *Mu = M, *Mv = M + 3, *Mw = M + 6;
  e->valid = 0; e->stop = 0; e->median = 0; e->forced = forced;
  for (int j = 0; j < 3; ++j) { e->hu[j] = px * Mw[j] - Mu[j]; e->hv[j] = py * Mw[j] - Mv[j]; }
  e->z[0] = e->hu[1] * e->hv[2] - e->hu[2] * e->hv[1];
  e->z[1] = e->hu[2] * e->hv[0] - e->hu[0] * e->hv[2];
  e->z[2] = e->hu[0] * e->hv[1] - e->hu[1] * e->hv[0];
  e->d[0] = xy[0] - px; e->d[1] = xy[1] - py;
  e->g2 = FILTER_INV_SQUARE * (e->d[0] * e->d[0] + e->d[1] * e->d[1]);
  if (e->z[2] == 0) { e->s[0] = e->s[1] = 0; e->g3 = (REAL)INFINITY; }
  else { e->s[0] = e->z[0] / e->z[2]; e->s[1] = e->z[1] / e->z[2]; e->g3 = e->s[0] * e->s[0] + e->s[1] * e->s[1]; }
  e->branch3d = forced ? ((bits >> 1) & 1) : (e->z[2] != 0 && e->g3 <= e->g2);
  e->sigma = (REAL)0.5 * (e->branch3d ? e->g3 : e->g2);
  e->vis = (REAL)exp(-(double)e->sigma);
  e->a_raw = opac * e->vis;
  e->clamped = forced ? ((bits >> 2) & 1) : (e->a_raw > ALPHA_MAX);
  e->alpha = e->clamped ? ALPHA_MAX : e->a_raw;
  e->dep = e->branch3d ? (e->s[0] * Mw[0] + e->s[1] * Mw[1]) + Mw[2] : Mw[2];
  if (forced) { e->valid = (bits & 1) != 0; e->stop = (bits >> 3) & 1; e->median = (bits >> 4) & 1; }
  else e->valid = (e->z[2] != 0) && (e->sigma >= 0) && (e->alpha >= TILE_ALPHA_MIN);
  /* conditioning of s = z.xy / z.z in the direct form z = h_u x h_v (what the reference's kernel evaluates): every component of z is a
   * difference of two products */
  e->kappa = 1.0; e->ks = 1.0; e->r = e->clamped ? 0.0 : 2.0;
  if (e->branch3d && e->z[2] != 0) {
    const double iz = 1.0 / fabs((double)e->z[2]);
    const double Ax = fabs((double)(e->hu[1] * e->hv[2])) + fabs((double)(e->hu[2] * e->hv[1]));
    const double Ay = fabs((double)(e->hu[2] * e->hv[0])) + fabs((double)(e->hu[0] * e->hv[2]));
    const double Az = fabs((double)(e->hu[0] * e->hv[1])) + fabs((double)(e->hu[1] * e->hv[0]));
    const double sx = fabs((double)e->s[0]), sy = fabs((double)e->s[1]), sn = sqrt(sx * sx + sy * sy);
    e->kappa = Az * iz;
    const double dsx = Ax * iz + sx * e->kappa, dsy = Ay * iz + sy * e->kappa;   /* absolute error of s, eps units */
    e->ks = sn > 0 ? sqrt(dsx * dsx + dsy * dsy) / sn : e->kappa;
    if (!e->clamped) e->r = 2.0 + sx * dsx + sy * dsy;                           /* d sigma = s . ds */
  }
}

/* pix_bound [C*H*W][5] (colors, depths, alphas, normals, median), vis_bound [M]: eps32 units, may be NULL.
 * flip_counts [5] / flip_worst [5]: alpha test, branch, clamp, termination, median (may be NULL). */
void orc_rasterize_2dgs_fwd_matched(int64_t C, int64_t M, int64_t I, int W, int H, int tile_size,
                                    const REAL *means2d, const REAL *ray_transforms, const REAL *colors,
                                    const REAL *opacities, const REAL *normals, const REAL *backgrounds,
                                    const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                    const int32_t *trace_rows, int64_t trace_stride, const uint8_t *trace_bits, double ulp_floor,
                                    REAL *render_colors, REAL *render_depths, REAL *render_alphas, REAL *render_normals,
                                    REAL *render_median, int32_t *last_ids, int32_t *median_ids, REAL *visibilities,
                                    double *pix_bound, double *vis_bound, int64_t *flip_counts, double *flip_worst) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t n_tiles = (int64_t)tw * th;
  for (int64_t m = 0; m < M; ++m) { visibilities[m] = 0; if (vis_bound) vis_bound[m] = 0; }
  int64_t fc[5] = {0, 0, 0, 0, 0};
  double fw_[5] = {0, 0, 0, 0, 0};
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t t = 0; t < C * n_tiles; ++t) {
    int64_t c = t / n_tiles, tl = t % n_tiles;
    int ty = (int)(tl / tw), tx = (int)(tl % tw);
    int32_t start = isect_offsets[t];
    int32_t end = (t == C * n_tiles - 1) ? (int32_t)I : isect_offsets[t + 1];
    int masked = masks && !masks[t];
    int32_t len = masked ? 0 : end - start;
    REAL *vis_local = (REAL *)calloc((size_t)(len > 0 ? len : 1), sizeof(REAL));
    double *visb_local = (double *)calloc((size_t)(len > 0 ? len : 1), sizeof(double));
    int64_t lfc[5] = {0, 0, 0, 0, 0};
    double lfw[5] = {0, 0, 0, 0, 0};
    for (int yy = 0; yy < tile_size; ++yy)
      for (int xx = 0; xx < tile_size; ++xx) {
        int i = ty * tile_size + yy, j = tx * tile_size + xx;
        if (i >= H || j >= W) continue;
        int64_t pid = (c * H + i) * (int64_t)W + j;
        REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5;
        const int32_t row = trace_rows ? trace_rows[pid] : -1;
        const uint8_t *bits = row >= 0 ? trace_bits + (int64_t)row * trace_stride : NULL;
        REAL T = 1, col[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, dsum = 0, med = 0;
        int32_t cur = 0, med_idx = 0;
        double Racc = 0 /* sum of squares */, eT = 0, bcol = 0, bnrm = 0, bdep = 0, bmed = 0;
        for (int32_t k = 0; k < len; ++k) {
          int32_t g = flatten_ids[start + k];
          mpair_t e;
          const int forced = bits != NULL && k < trace_stride;
          eval_pair_matched(px, py, means2d + 2 * g, opacities[g], ray_transforms + 9 * g, forced, forced ? bits[k] : 0, &e);
          double err_a = 0, err_g = 0;
          if (forced) {
            /* margins of the decisions the fp64 evaluation would have taken itself, in units of the fp32 evaluation error */
            float Mf[9];
            for (int q = 0; q < 9; ++q) Mf[q] = (float)ray_transforms[9 * g + q];
            pair32_t f = eval_pair_f32((float)px, (float)py, (float)means2d[2 * g], (float)means2d[2 * g + 1], (float)opacities[g], Mf);
            const double g3o = (double)e.g3, g2o = (double)e.g2;
            const int own_b3 = e.z[2] != 0 && g3o <= g2o;
            const double a_own = (double)opacities[g] * exp(-0.5 * (own_b3 ? g3o : g2o));
            err_a = fabs((double)f.a - a_own) + ulp_floor * a_own;
            err_g = fabs((double)f.g3 - g3o) + fabs((double)f.g2 - g2o) + ulp_floor * (g3o > g2o ? g3o : g2o);
            const double al_own = a_own < 0.999 ? a_own : 0.999;
            const int own_valid = e.z[2] != 0 && al_own >= 1.0 / 255.0;
            const int f_valid = e.valid || e.stop;
            if (own_valid != f_valid) {
              double ratio = fabs(al_own - 1.0 / 255.0) / err_a;
              if (e.z[2] == 0 || f.zero) ratio = INFINITY;
              lfc[0]++; if (ratio > lfw[0]) lfw[0] = ratio;
            }
            if (f_valid && own_valid) {
              if (own_b3 != e.branch3d) { double ratio = fabs(g3o - g2o) / err_g; lfc[1]++; if (ratio > lfw[1]) lfw[1] = ratio; }
              if ((a_own > 0.999) != e.clamped) { double ratio = fabs(a_own - 0.999) / err_a; lfc[2]++; if (ratio > lfw[2]) lfw[2] = ratio; }
            }
          }
          if (!forced) {
            if (!e.valid) continue;
            if (T * (1 - e.alpha) <= T_EPS) break;
            e.median = T > (REAL)0.5;
          } else {
            if (e.valid || e.stop) {
              const double nT = (double)T * (1 - (double)e.alpha);
              const double enT = eT * (1 - (double)e.alpha) + (double)T * err_a + ulp_floor * nT;
              const int own_stop = nT <= 1e-4;
              if (own_stop != e.stop) { double ratio = fabs(nT - 1e-4) / enT; lfc[3]++; if (ratio > lfw[3]) lfw[3] = ratio; }
              if (!e.stop && (((double)T > 0.5) != e.median)) {
                double ratio = fabs((double)T - 0.5) / (eT + ulp_floor * (double)T); lfc[4]++; if (ratio > lfw[4]) lfw[4] = ratio;
              }
              if (!e.stop) eT = enT;
            }
            if (e.stop) break;
            if (!e.valid) continue;
          }
          const REAL nT = T * (1 - e.alpha);
          REAL w = e.alpha * T;
          const double rel = sqrt(Racc + e.r * e.r + 1.0);
          REAL cmax = 0, nmax = 0;
          for (int ch = 0; ch < 3; ++ch) {
            col[ch] += colors[3 * g + ch] * w;
            nrm[ch] += normals[3 * g + ch] * w;
            cmax = rmax(cmax, (REAL)fabs((double)colors[3 * g + ch])); nmax = rmax(nmax, (REAL)fabs((double)normals[3 * g + ch]));
          }
          dsum += e.dep * w;
          bcol += (double)w * (double)cmax * rel; bnrm += (double)w * (double)nmax * rel;
          const double adepv = e.branch3d ? (fabs((double)(e.s[0] * ray_transforms[9 * g + 6])) + fabs((double)(e.s[1] * ray_transforms[9 * g + 7]))) * (e.ks + 1.0) +
                                                fabs((double)ray_transforms[9 * g + 8]) : fabs((double)e.dep);   /* eps units: error of dep */
          bdep += (double)w * (fabs((double)e.dep) * rel + adepv);
          if (e.median) { med = e.dep; med_idx = start + k; bmed = adepv + fabs((double)e.dep); }
          if (w > vis_local[k]) vis_local[k] = w;
          if ((double)w * rel > visb_local[k]) visb_local[k] = (double)w * rel;
          cur = start + k;
          const double am = (double)e.alpha / (1 - (double)e.alpha), rm_ = e.r * (am > 1 ? am : 1);
          Racc += rm_ * rm_ + 1.0;
          T = nT;
        }
        for (int ch = 0; ch < 3; ++ch) {
          render_colors[3 * pid + ch] = backgrounds ? col[ch] + T * backgrounds[3 * c + ch] : col[ch];
          render_normals[3 * pid + ch] = nrm[ch];
        }
        render_depths[pid] = dsum;
        render_alphas[pid] = 1 - T;
        render_median[pid] = med;
        last_ids[pid] = cur;
        median_ids[pid] = med_idx;
        if (pix_bound) {
          double *b = pix_bound + 5 * pid;
          double bgm = 0;
          if (backgrounds) for (int ch = 0; ch < 3; ++ch) bgm = fmax(bgm, fabs((double)backgrounds[3 * c + ch]));
          const double Rf = sqrt(Racc + 1.0);
          b[0] = bcol + (double)T * Rf * bgm; b[1] = bdep; b[2] = (double)T * Rf; b[3] = bnrm; b[4] = bmed;
        }
      }
#pragma omp critical
    {
      for (int32_t k = 0; k < len; ++k) {
        int32_t g = flatten_ids[start + k];
        if (vis_local[k] > visibilities[g]) visibilities[g] = vis_local[k];
        if (vis_bound && visb_local[k] > vis_bound[g]) vis_bound[g] = visb_local[k];
      }
      for (int q = 0; q < 5; ++q) { fc[q] += lfc[q]; if (lfw[q] > fw_[q]) fw_[q] = lfw[q]; }
    }
    free(vis_local); free(visb_local);
  }
  if (flip_counts) for (int q = 0; q < 5; ++q) flip_counts[q] = fc[q];
  if (flip_worst) for (int q = 0; q < 5; ++q) flip_worst[q] = fw_[q];
}

/* VJP under the same decisions.  last_ids / median_ids / render_alphas are those of orc_rasterize_2dgs_fwd_matched.
 * Gradients [M,.] double, zeroed by the caller; cond [M][44], zeroed by the caller, may be NULL: per gradient element (layout:
 * means2d 2, M 9, colors 3, opacity 1, normals 3, densify 2, means2d_abs 2) first the sum over pixels of (F |term|)^2 — F the
 * pair's relative-error factor, root-sum-square model of independent roundings — then the plain sum of |term| (the fp32
 * accumulation itself).  The caller's bound is sqrt(first) + ACC x second, in eps32 units. */
/* Gradient accumulation of the matched backward.  Default: every term (evaluated in REAL) is summed in double, so that the f64 build is the
 * reference and the f32 builds isolate the per-term evaluation error.  -DACC_FLOAT (the builds liborc_splat_f32acc / _f32fmaacc): the sums
 * are fp32 as well, term after term in pixel order and tile after tile — a SECOND, fully-fp32 evaluation of the operator with an operation
 * order unlike the HIP kernel's (wave trees + atomics), used by the parity tests as the independent leg under the conditioning bound. */
#ifdef ACC_FLOAT
#define ACCADD(x, v) do { (x) = (double)(float)((x) + (double)(float)(v)); } while (0)
#else
#define ACCADD(x, v) do { (x) += (v); } while (0)
#endif
void orc_rasterize_2dgs_bwd_matched(int64_t C, int64_t M, int64_t I, int W, int H, int tile_size,
                                    const REAL *means2d, const REAL *ray_transforms, const REAL *colors,
                                    const REAL *opacities, const REAL *normals, const REAL *backgrounds,
                                    const uint8_t *masks, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                    const int32_t *trace_rows, int64_t trace_stride, const uint8_t *trace_bits,
                                    const REAL *render_alphas, const int32_t *last_ids, const int32_t *median_ids,
                                    const REAL *v_render_colors, const REAL *v_render_depths, const REAL *v_render_alphas,
                                    const REAL *v_render_normals, const REAL *v_render_median, double *v_means2d,
                                    double *v_ray_transforms, double *v_colors, double *v_opacities, double *v_normals,
                                    double *v_densify, double *v_means2d_abs, double *cond,
                                    double t_final_abs_err /* eps32 units: 1 for an implementation that recovers the final transmittance
                                                              as 1 - render_alphas (upstream gsplat, this file's fp32 builds), 0 for one
                                                              that saves it (libgsdf_hip's final_T) */) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t n_tiles = (int64_t)tw * th;
  (void)M;
  enum { NG = 66 }; /* 0..21 gradients (layout of cond), 22..43 sum of (F |term|)^2, 44..65 sum of |term| */
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t t = 0; t < C * n_tiles; ++t) {
    int64_t c = t / n_tiles, tl = t % n_tiles;
    int ty = (int)(tl / tw), tx = (int)(tl % tw);
    int32_t start = isect_offsets[t];
    int32_t end = (t == C * n_tiles - 1) ? (int32_t)I : isect_offsets[t + 1];
    if (masks && !masks[t]) continue;
    int32_t len = end - start;
    if (len <= 0) continue;
    double *acc = (double *)calloc((size_t)len * NG, sizeof(double));
    for (int yy = 0; yy < tile_size; ++yy)
      for (int xx = 0; xx < tile_size; ++xx) {
        int i = ty * tile_size + yy, j = tx * tile_size + xx;
        if (i >= H || j >= W) continue;
        int64_t pid = (c * H + i) * (int64_t)W + j;
        REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5;
        const int32_t row = trace_rows ? trace_rows[pid] : -1;
        const uint8_t *bits = row >= 0 ? trace_bits + (int64_t)row * trace_stride : NULL;
        const REAL T_final = 1 - render_alphas[pid];
        REAL T = T_final;
        const int32_t bin_final = last_ids[pid], med_idx = median_ids[pid];
        const REAL *vC = v_render_colors + 3 * pid, *vN = v_render_normals + 3 * pid;
        const REAL vD = v_render_depths[pid], vA = v_render_alphas[pid], vMed = v_render_median[pid];
        REAL bufC[3] = {0, 0, 0}, bufN[3] = {0, 0, 0}, bufD = 0;
        double abufC[3] = {0, 0, 0}, abufN[3] = {0, 0, 0}, abufD = 0;
        REAL bgdot = 0;
        double abgdot = 0;
        if (backgrounds) {
          bgdot = backgrounds[3 * c] * vC[0] + backgrounds[3 * c + 1] * vC[1] + backgrounds[3 * c + 2] * vC[2];
          abgdot = fabs((double)(backgrounds[3 * c] * vC[0])) + fabs((double)(backgrounds[3 * c + 1] * vC[1])) + fabs((double)(backgrounds[3 * c + 2] * vC[2]));
        }
        /* the pixel's weight conditioning R (block comment): one pass over its blended pairs */
        double Rpix = 0;
        if (cond && t_final_abs_err > 0) { const double q = t_final_abs_err / fmax((double)T_final, 1e-30); Rpix += q * q; }
        if (cond)
          for (int32_t idx = start; idx <= bin_final; ++idx) {
            int32_t g = flatten_ids[idx];
            mpair_t e;
            const int k = idx - start, forced = bits != NULL && k < trace_stride;
            eval_pair_matched(px, py, means2d + 2 * g, opacities[g], ray_transforms + 9 * g, forced, forced ? bits[k] : 0, &e);
            if (!e.valid) continue;
            const double am = (double)e.alpha / (1 - (double)e.alpha), rm_ = e.r * (am > 1 ? am : 1);
            Rpix += rm_ * rm_ + 1.0;    /* sum of squares */
          }
        for (int32_t idx = bin_final; idx >= start; --idx) {
          int32_t g = flatten_ids[idx];
          mpair_t e;
          const REAL *Mrow = ray_transforms + 9 * g;
          const int k = idx - start, forced = bits != NULL && k < trace_stride;
          eval_pair_matched(px, py, means2d + 2 * g, opacities[g], Mrow, forced, forced ? bits[k] : 0, &e);
          if (!e.valid) continue;
          double *a = acc + (size_t)(idx - start) * NG, *b = a + 22, *l = a + 44;
#define BND(slot, F, term) do { const double t_ = (term), f_ = (F) * t_; b[slot] += f_ * f_; l[slot] += t_; } while (0)
          const REAL ra = 1 / (1 - e.alpha);
          T *= ra;
          const REAL fac = e.alpha * T;
          REAL v_alpha = 0;
          double A = 0;   /* sum of the absolute values of v_alpha's terms */
          for (int ch = 0; ch < 3; ++ch) {
            ACCADD(a[11 + ch], fac * vC[ch]);
            ACCADD(a[15 + ch], fac * vN[ch]);
            BND(11 + ch, sqrt(Rpix), fabs((double)(fac * vC[ch])));
            BND(15 + ch, sqrt(Rpix), fabs((double)(fac * vN[ch])));
            v_alpha += (colors[3 * g + ch] * T - bufC[ch] * ra) * vC[ch];
            v_alpha += (normals[3 * g + ch] * T - bufN[ch] * ra) * vN[ch];
            A += (fabs((double)(colors[3 * g + ch] * T)) + abufC[ch] * (double)ra) * fabs((double)vC[ch]);
            A += (fabs((double)(normals[3 * g + ch] * T)) + abufN[ch] * (double)ra) * fabs((double)vN[ch]);
          }
          v_alpha += (e.dep * T - bufD * ra) * vD;
          v_alpha += T_final * ra * vA;
          v_alpha += -T_final * ra * bgdot;
          A += (fabs((double)(e.dep * T)) + abufD * (double)ra) * fabs((double)vD) + (double)(T_final * ra) * (fabs((double)vA) + abgdot);
          const int is_med = (idx == med_idx);
          REAL v_dep = fac * vD + (is_med ? vMed : 0);
          const double adep = fabs((double)(fac * vD)) + (is_med ? fabs((double)vMed) : 0);
          for (int ch = 0; ch < 3; ++ch) {
            bufC[ch] += colors[3 * g + ch] * fac;
            bufN[ch] += normals[3 * g + ch] * fac;
            abufC[ch] += fabs((double)(colors[3 * g + ch] * fac));
            abufN[ch] += fabs((double)(normals[3 * g + ch] * fac));
          }
          bufD += e.dep * fac;
          abufD += fabs((double)(e.dep * fac));
          REAL v_sigma = 0;
          double asig = 0;
          const double Fw = sqrt(Rpix + e.r * e.r + e.ks * e.ks);   /* ks: e.dep inside v_alpha carries the error of s */
          if (!e.clamped) {
            ACCADD(a[14], e.vis * v_alpha);
            BND(14, Fw, (double)e.vis * A);
            v_sigma = -opacities[g] * e.vis * v_alpha;
            asig = (double)opacities[g] * (double)e.vis * A;
          }
          const REAL *Mw = Mrow + 6;
          if (e.branch3d) {
            REAL v_s[2] = {v_sigma * e.s[0] + v_dep * Mw[0], v_sigma * e.s[1] + v_dep * Mw[1]};
            REAL vsx = v_s[0] / e.z[2], vsy = v_s[1] / e.z[2];
            REAL v_z[3] = {vsx, vsy, -(vsx * e.s[0] + vsy * e.s[1])};
            REAL v_hu[3] = {e.hv[1] * v_z[2] - e.hv[2] * v_z[1], e.hv[2] * v_z[0] - e.hv[0] * v_z[2],
                            e.hv[0] * v_z[1] - e.hv[1] * v_z[0]};
            REAL v_hv[3] = {v_z[1] * e.hu[2] - v_z[2] * e.hu[1], v_z[2] * e.hu[0] - v_z[0] * e.hu[2],
                            v_z[0] * e.hu[1] - v_z[1] * e.hu[0]};
            REAL vMw[3] = {px * v_hu[0] + py * v_hv[0] + v_dep * e.s[0],
                           px * v_hu[1] + py * v_hv[1] + v_dep * e.s[1],
                           px * v_hu[2] + py * v_hv[2] + v_dep};
            for (int q = 0; q < 3; ++q) { ACCADD(a[2 + q], -v_hu[q]); ACCADD(a[5 + q], -v_hv[q]); ACCADD(a[8 + q], vMw[q]); }
            ACCADD(a[18], -v_hu[2] * Mw[2]);
            ACCADD(a[19], -v_hv[2] * Mw[2]);
            if (cond) {
              const double iz = 1.0 / fabs((double)e.z[2]);
              const double sn = sqrt((double)e.g3);
              const double avs[2] = {asig * sn + adep * fabs((double)Mw[0]), asig * sn + adep * fabs((double)Mw[1])};
              const double za[3] = {avs[0] * iz, avs[1] * iz, (avs[0] + avs[1]) * iz * sn};
              const double F = sqrt(Rpix + e.r * e.r + 4.0 * e.ks * e.ks + e.kappa * e.kappa + 16.0);
              for (int q = 0; q < 3; ++q) {
                const int q1 = (q + 1) % 3, q2 = (q + 2) % 3;
                const double ahu = fabs((double)e.hv[q1]) * za[q2] + fabs((double)e.hv[q2]) * za[q1];
                const double ahv = fabs((double)e.hu[q1]) * za[q2] + fabs((double)e.hu[q2]) * za[q1];
                BND(2 + q, F, ahu);
                BND(5 + q, F, ahv);
                BND(8 + q, F, fabs((double)px) * ahu + fabs((double)py) * ahv + adep * (q < 2 ? fabs((double)e.s[q]) : 1.0));
                if (q == 2) { BND(18, F, ahu * fabs((double)Mw[2])); BND(19, F, ahv * fabs((double)Mw[2])); }
              }
            }
          } else {
            REAL gx = v_sigma * FILTER_INV_SQUARE * e.d[0], gy = v_sigma * FILTER_INV_SQUARE * e.d[1];
            ACCADD(a[0], gx); ACCADD(a[1], gy);
            ACCADD(a[20], fabs((double)gx)); ACCADD(a[21], fabs((double)gy));
            ACCADD(a[10], v_dep);
            if (cond) {
              const double F = sqrt(Rpix + e.r * e.r + 16.0);
              const double agx = asig * 2.0 * fabs((double)e.d[0]), agy = asig * 2.0 * fabs((double)e.d[1]);
              BND(0, F, agx); BND(1, F, agy); BND(20, F, agx); BND(21, F, agy);
              BND(10, sqrt(Rpix + 4.0), adep);
            }
          }
        }
      }
#pragma omp critical
    for (int32_t k = 0; k < len; ++k) {
      int32_t g = flatten_ids[start + k];
      const double *a = acc + (size_t)k * NG;
      ACCADD(v_means2d[2 * g], a[0]); ACCADD(v_means2d[2 * g + 1], a[1]);
      for (int q = 0; q < 9; ++q) ACCADD(v_ray_transforms[9 * g + q], a[2 + q]);
      for (int q = 0; q < 3; ++q) { ACCADD(v_colors[3 * g + q], a[11 + q]); ACCADD(v_normals[3 * g + q], a[15 + q]); }
      ACCADD(v_opacities[g], a[14]);
      ACCADD(v_densify[2 * g], a[18]); ACCADD(v_densify[2 * g + 1], a[19]);
      if (v_means2d_abs) { ACCADD(v_means2d_abs[2 * g], a[20]); ACCADD(v_means2d_abs[2 * g + 1], a[21]); }
      if (cond) for (int q = 0; q < 22; ++q) { cond[44 * (int64_t)g + q] += a[22 + q]; cond[44 * (int64_t)g + 22 + q] += a[44 + q]; }
    }
#undef BND
    free(acc);
  }
}

#undef ACCADD
/* The decision record of THIS build's own evaluation (same byte layout as gsdf_rasterize_2dgs_fwd_instr writes): used by the
 * CPU self-check of the decision-matched gate, where the fp32 build stands in for the implementation under test. */
void orc_rasterize_2dgs_trace(int64_t C, int64_t M, int64_t I, int W, int H, int tile_size, const REAL *means2d,
                              const REAL *ray_transforms, const REAL *opacities, const uint8_t *masks,
                              const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *trace_rows,
                              int64_t trace_stride, uint8_t *trace_bits) {
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  int64_t n_tiles = (int64_t)tw * th;
  (void)M;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t t = 0; t < C * n_tiles; ++t) {
    int64_t c = t / n_tiles, tl = t % n_tiles;
    int ty = (int)(tl / tw), tx = (int)(tl % tw);
    int32_t start = isect_offsets[t];
    int32_t end = (t == C * n_tiles - 1) ? (int32_t)I : isect_offsets[t + 1];
    int32_t len = (masks && !masks[t]) ? 0 : end - start;
    for (int yy = 0; yy < tile_size; ++yy)
      for (int xx = 0; xx < tile_size; ++xx) {
        int i = ty * tile_size + yy, j = tx * tile_size + xx;
        if (i >= H || j >= W) continue;
        int64_t pid = (c * H + i) * (int64_t)W + j;
        if (trace_rows[pid] < 0) continue;
        uint8_t *bits = trace_bits + (int64_t)trace_rows[pid] * trace_stride;
        REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5, T = 1;
        for (int32_t k = 0; k < len && k < trace_stride; ++k) {
          int32_t g = flatten_ids[start + k];
          pix_eval_t e;
          eval_pair(px, py, means2d + 2 * g, opacities[g], ray_transforms + 9 * g, &e);
          if (!e.valid) continue;
          uint8_t b = (uint8_t)((e.branch3d ? 2 : 0) | (e.clamped ? 4 : 0));
          REAL nT = T * (1 - e.alpha);
          if (nT <= T_EPS) { bits[k] = b | 8; break; }
          bits[k] = b | 1 | (T > (REAL)0.5 ? 16 : 0);
          T = nT;
        }
      }
  }
}
