/*
 * sdf_oracle.c — CPU restatement of the hash-grid SDF path of GS-SDF.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): checker for tests/, smoke() and the
 * cpu_baseline leg of bench.py; never linked or called by the product.
 *
 * PARITY UNPINNED.  The reference calls NVlabs/tiny-cuda-nn through the un-vendored submodule
 * jianhengLiu/tcnn_binding (/root/reference/.gitmodules, empty directory, SHA unknown) and
 * graphdeco simple-knn; no tests/golden vectors exist.  This file restates the published
 * algorithms anchored on the reference's call sites:
 *   - multiresolution hash encoding, config {Grid, Hash, n_levels 16, n_features_per_level 2,
 *     log2_hashmap_size 19, base_resolution 32, per_level_scale 2.0, Linear}:
 *       /root/reference/include/neural_net/encoding_map.cpp:15-26 (config), :59 (forward)
 *       tiny-cuda-nn include/tiny-cuda-nn/encodings/grid.h (grid_scale, grid_resolution,
 *       pos_fract with the +0.5 offset, coherent prime hash {1, 2654435761, 805459861},
 *       dense index while stride <= hashmap_size, params per level rounded up to 8)
 *   - decoder MLP: Linear(32,64)+ReLU, 3x[Linear(64,64)+ReLU], Linear(64,2)
 *       /root/reference/include/neural_net/local_map.cpp:29-42 (torch) / :44-55 (tcnn, bias free)
 *   - SDF head: sdf = out[0], isigma = 1 + softplus_{beta=100}(out[1]) * (1/bce_sigma)
 *       /root/reference/include/neural_net/local_map.cpp:97-102
 *   - distCUDA2: mean squared distance to the 3 nearest neighbours
 *       /root/reference/include/neural_gaussian/neural_gaussian.cpp:314
 * Compiled twice (-DREAL=float / double) like splat_oracle.c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#define MAX_LEVELS 32

typedef struct {
  int n_levels, n_feat, log2_hashmap, base_res;
  float per_level_scale;
} grid_cfg_t;

/* tiny-cuda-nn: grid_scale / grid_resolution are evaluated in fp32 */
static float level_scale(const grid_cfg_t *c, int l) {
  return exp2f((float)l * log2f(c->per_level_scale)) * (float)c->base_res - 1.0f;
}
static uint32_t level_res(float scale) { return (uint32_t)ceilf(scale) + 1u; }

/* offsets[l] in ENTRIES (not floats); offsets[n_levels] = total entries.  Returns total. */
int64_t orc_grid_offsets(int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                         int64_t *offsets) {
  grid_cfg_t c = {n_levels, n_feat, log2_hashmap, base_res, per_level_scale};
  int64_t off = 0;
  for (int l = 0; l < n_levels; ++l) {
    uint32_t res = level_res(level_scale(&c, l));
    double dense = pow((double)res, 3.0);
    uint64_t max_params = 0xFFFFFFFFu / 2;
    uint64_t p = dense > (double)max_params ? max_params : (uint64_t)res * res * res;
    p = (p + 7) / 8 * 8;
    uint64_t cap = 1ull << log2_hashmap;
    if (p > cap) p = cap;
    offsets[l] = off;
    off += (int64_t)p;
  }
  offsets[n_levels] = off;
  return off;
}

static inline uint32_t grid_index(uint32_t hashmap_size, uint32_t res, const uint32_t g[3]) {
  uint32_t stride = 1, index = 0;
  for (int d = 0; d < 3 && stride <= hashmap_size; ++d) {
    index += g[d] * stride;
    stride *= res;
  }
  if (hashmap_size < stride) index = (g[0] * 1u) ^ (g[1] * 2654435761u) ^ (g[2] * 805459861u);
  return index % hashmap_size;
}

typedef struct {
  uint32_t idx[8];
  REAL w[8];        /* trilinear weight */
  REAL dw[8][3];    /* d w / d pos_d (before the *scale) */
  REAL ddw[8][3][3];/* d2 w / d pos_d d pos_e */
  REAL scale;
} corner_t;

static void level_corners(const grid_cfg_t *c, const int64_t *offsets, int l, const REAL x[3], corner_t *o) {
  float scale = level_scale(c, l);
  uint32_t res = level_res(scale);
  uint32_t hsize = (uint32_t)(offsets[l + 1] - offsets[l]);
  REAL fr[3];
  uint32_t g0[3];
  for (int d = 0; d < 3; ++d) {
#ifdef REAL_IS_FLOAT
    REAL pos = fmaf(scale, x[d], 0.5f); /* tiny-cuda-nn pos_fract uses fmaf */
#else
    REAL pos = fma((double)scale, x[d], 0.5);
#endif
    REAL fl = (REAL)floor((double)pos);
    g0[d] = (uint32_t)(int32_t)fl;
    fr[d] = pos - fl;
  }
  o->scale = (REAL)scale;
  for (int k = 0; k < 8; ++k) {
    uint32_t g[3];
    REAL wd[3], sg[3];
    for (int d = 0; d < 3; ++d) {
      int hi = (k >> d) & 1;
      g[d] = g0[d] + (uint32_t)hi;
      wd[d] = hi ? fr[d] : 1 - fr[d];
      sg[d] = hi ? (REAL)1 : (REAL)-1;
    }
    o->idx[k] = grid_index(hsize, res, g);
    o->w[k] = wd[0] * wd[1] * wd[2];
    for (int d = 0; d < 3; ++d) {
      REAL p = sg[d];
      for (int e = 0; e < 3; ++e)
        if (e != d) p *= wd[e];
      o->dw[k][d] = p;
      for (int e = 0; e < 3; ++e) {
        if (e == d) { o->ddw[k][d][e] = 0; continue; }
        int r = 3 - d - e; /* remaining dim */
        o->ddw[k][d][e] = sg[d] * sg[e] * wd[r];
      }
    }
  }
}

/* S1 forward: x [B,3] -> feat [B, L*F] ; optional dfeat_dx [B, L*F, 3] */
void orc_grid_fwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                  const int64_t *offsets, const REAL *x, const REAL *table, REAL *feat, REAL *dfeat_dx) {
  grid_cfg_t c = {n_levels, n_feat, log2_hashmap, base_res, per_level_scale};
  const int F = n_feat;
#pragma omp parallel for
  for (int64_t b = 0; b < B; ++b)
    for (int l = 0; l < n_levels; ++l) {
      corner_t cr;
      level_corners(&c, offsets, l, x + 3 * b, &cr);
      for (int f = 0; f < F; ++f) {
        REAL acc = 0, g[3] = {0, 0, 0};
        for (int k = 0; k < 8; ++k) {
          REAL v = table[(offsets[l] + cr.idx[k]) * F + f];
          acc += cr.w[k] * v;
          for (int d = 0; d < 3; ++d) g[d] += cr.dw[k][d] * v;
        }
        feat[b * n_levels * F + l * F + f] = acc;
        if (dfeat_dx)
          for (int d = 0; d < 3; ++d) dfeat_dx[(b * n_levels * F + l * F + f) * 3 + d] = cr.scale * g[d];
      }
    }
}

/* S1' backward: v_feat [B,L*F] -> v_table (ACCUMULATE, double), v_x [B,3] */
void orc_grid_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                  const int64_t *offsets, const REAL *x, const REAL *table, const REAL *v_feat, double *v_table,
                  REAL *v_x) {
  grid_cfg_t c = {n_levels, n_feat, log2_hashmap, base_res, per_level_scale};
  const int F = n_feat;
  /* threaded over the points (round 5: the cpu_baseline leg of bench.py times this function); the table gradient is a double sum whose order
   * then varies by ~1e-16 relative */
#pragma omp parallel for
  for (int64_t b = 0; b < B; ++b) {
    REAL gx[3] = {0, 0, 0};
    for (int l = 0; l < n_levels; ++l) {
      corner_t cr;
      level_corners(&c, offsets, l, x + 3 * b, &cr);
      for (int f = 0; f < F; ++f) {
        REAL vf = v_feat[b * n_levels * F + l * F + f];
        for (int k = 0; k < 8; ++k) {
          int64_t e = (offsets[l] + cr.idx[k]) * F + f;
          if (v_table) {
            const double add = (double)(cr.w[k] * vf);
#pragma omp atomic
            v_table[e] += add;
          }
          for (int d = 0; d < 3; ++d) gx[d] += vf * cr.scale * cr.dw[k][d] * table[e];
        }
      }
    }
    if (v_x)
      for (int d = 0; d < 3; ++d) v_x[3 * b + d] = gx[d];
  }
}

/* S1'' double backward of  v_x = J(x,table)^T v_feat : given vv_x [B,3] (gradient w.r.t. v_x) returns
 *   g_vfeat [B,L*F]  = d/d v_feat,   g_table (ACCUMULATE, double) = d/d table,   g_x [B,3] = d/d x */
void orc_grid_bwd_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                      const int64_t *offsets, const REAL *x, const REAL *table, const REAL *v_feat,
                      const REAL *vv_x, REAL *g_vfeat, double *g_table, REAL *g_x) {
  grid_cfg_t c = {n_levels, n_feat, log2_hashmap, base_res, per_level_scale};
  const int F = n_feat;
  for (int64_t b = 0; b < B; ++b) {
    REAL gx[3] = {0, 0, 0};
    const REAL *vv = vv_x + 3 * b;
    for (int l = 0; l < n_levels; ++l) {
      corner_t cr;
      level_corners(&c, offsets, l, x + 3 * b, &cr);
      for (int f = 0; f < F; ++f) {
        REAL vf = v_feat[b * n_levels * F + l * F + f];
        REAL gvf = 0;
        for (int k = 0; k < 8; ++k) {
          int64_t e = (offsets[l] + cr.idx[k]) * F + f;
          REAL t = 0; /* sum_d vv_d * scale * dw_d */
          for (int d = 0; d < 3; ++d) t += vv[d] * cr.scale * cr.dw[k][d];
          gvf += t * table[e];
          if (g_table) g_table[e] += (double)(t * vf);
          for (int ee = 0; ee < 3; ++ee) {
            REAL s2 = 0;
            for (int d = 0; d < 3; ++d) s2 += vv[d] * cr.ddw[k][d][ee];
            gx[ee] += s2 * cr.scale * cr.scale * vf * table[e];
          }
        }
        if (g_vfeat) g_vfeat[b * n_levels * F + l * F + f] = gvf;
      }
    }
    if (g_x)
      for (int d = 0; d < 3; ++d) g_x[3 * b + d] = gx[d];
  }
}

/* S1 third order: the backward of the double backward above.  With (g_vfeat, g_table, g_x) = orc_grid_bwd_bwd(x, table, v_feat, vv_x), given
 * lam_x [B,3] (the gradient arriving at g_x) and mu [B,L*F] (arriving at g_vfeat; NULL = zero) returns
 *   t_vfeat [B,L*F] = d/d v_feat,  t_table (ACCUMULATE, double) = d/d table,  t_vv [B,3] = d/d vv_x,  t_x [B,3] = d/d x.
 * What a loss on the analytic Hessian needs (LocalMap::get_gradient with hessian = true, numerical_grad = 0, then curvate_loss:
 * /root/reference/include/neural_net/local_map.cpp:151-168, include/neural_mapping/neural_mapping.cpp:117-121).  Trilinear weights: the second
 * derivatives are the mixed ones only, the third derivative is d3w/dxdydz = +-scale^3. */
void orc_grid_bwd3(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                   const int64_t *offsets, const REAL *x, const REAL *table, const REAL *v_feat, const REAL *vv_x,
                   const REAL *lam_x, const REAL *mu, REAL *t_vfeat, double *t_table, REAL *t_vv, REAL *t_x) {
  grid_cfg_t c = {n_levels, n_feat, log2_hashmap, base_res, per_level_scale};
  const int F = n_feat;
  for (int64_t b = 0; b < B; ++b) {
    REAL tv[3] = {0, 0, 0}, tx[3] = {0, 0, 0};
    const REAL *vv = vv_x + 3 * b, *lam = lam_x + 3 * b;
    for (int l = 0; l < n_levels; ++l) {
      corner_t cr;
      level_corners(&c, offsets, l, x + 3 * b, &cr);
      const REAL s1 = cr.scale, s2 = cr.scale * cr.scale, s3 = cr.scale * cr.scale * cr.scale;
      for (int f = 0; f < F; ++f) {
        const REAL vf = v_feat[b * n_levels * F + l * F + f];
        const REAL m = mu ? mu[b * n_levels * F + l * F + f] : 0;
        REAL tvf = 0;
        for (int k = 0; k < 8; ++k) {
          const int64_t e = (offsets[l] + cr.idx[k]) * F + f;
          const REAL th = table[e];
          REAL A = 0, Bk = 0;
          for (int d = 0; d < 3; ++d) {
            A += vv[d] * cr.dw[k][d];
            for (int ee = 0; ee < 3; ++ee) Bk += vv[d] * lam[ee] * cr.ddw[k][d][ee];
          }
          A *= s1; Bk *= s2;
          if (t_table) t_table[e] += (double)(vf * Bk + m * A);
          tvf += Bk * th;
          /* sign of the third derivative: the product of the three corner signs */
          const REAL sg3 = (REAL)((((k >> 0) & 1) ? 1 : -1) * (((k >> 1) & 1) ? 1 : -1) * (((k >> 2) & 1) ? 1 : -1));
          for (int d = 0; d < 3; ++d) {
            REAL a2 = 0, a4 = 0;
            for (int ee = 0; ee < 3; ++ee) {
              a2 += lam[ee] * cr.ddw[k][d][ee];            /* d/d vv_d of B */
              a4 += vv[ee] * cr.ddw[k][ee][d];             /* d/d x_d of A (the mu term) */
            }
            /* d/d x_d of B: the pairs (p, q) with {p, q, d} = {0, 1, 2} */
            const int p = (d + 1) % 3, q = (d + 2) % 3;
            const REAL a3 = sg3 * (vv[p] * lam[q] + vv[q] * lam[p]);
            tv[d] += vf * th * s2 * a2 + m * th * s1 * cr.dw[k][d];
            tx[d] += vf * th * s3 * a3 + m * th * s2 * a4;
          }
        }
        if (t_vfeat) t_vfeat[b * n_levels * F + l * F + f] = tvf;
      }
    }
    if (t_vv) for (int d = 0; d < 3; ++d) t_vv[3 * b + d] = tv[d];
    if (t_x) for (int d = 0; d < 3; ++d) t_x[3 * b + d] = tx[d];
  }
}

/* ---------------------------------------------------------------------------------------
 * S2 decoder MLP.  n_layers linear layers, dims[0..n_layers]; weights row-major [out][in]
 * concatenated, biases concatenated (NULL = bias free); ReLU after every layer but the last.
 * acts (optional) receives the post-ReLU hidden activations, layer after layer: [B, dims[1]] ...
 * ------------------------------------------------------------------------------------- */
void orc_mlp_fwd(int64_t B, int n_layers, const int *dims, const REAL *weights, const REAL *biases, const REAL *in,
                 REAL *out, REAL *acts) {
  int maxd = 0;
  for (int l = 0; l <= n_layers; ++l) if (dims[l] > maxd) maxd = dims[l];
  int64_t act_stride = 0;
  for (int l = 1; l < n_layers; ++l) act_stride += dims[l];
#pragma omp parallel for
  for (int64_t b = 0; b < B; ++b) {
    REAL *cur = (REAL *)malloc(sizeof(REAL) * maxd), *nxt = (REAL *)malloc(sizeof(REAL) * maxd);
    memcpy(cur, in + b * dims[0], sizeof(REAL) * dims[0]);
    const REAL *W = weights, *bi = biases;
    int64_t aoff = 0;
    for (int l = 0; l < n_layers; ++l) {
      int I = dims[l], O = dims[l + 1];
      for (int o = 0; o < O; ++o) {
        REAL acc = bi ? bi[o] : 0;
        for (int i = 0; i < I; ++i) acc += W[o * I + i] * cur[i];
        if (l < n_layers - 1 && acc < 0) acc = 0;
        nxt[o] = acc;
      }
      if (l < n_layers - 1 && acts) { memcpy(acts + b * act_stride + aoff, nxt, sizeof(REAL) * O); aoff += O; }
      W += (int64_t)O * I;
      if (bi) bi += O;
      REAL *t = cur; cur = nxt; nxt = t;
    }
    memcpy(out + b * dims[n_layers], cur, sizeof(REAL) * dims[n_layers]);
    free(cur); free(nxt);
  }
}

/* v_out [B,dims[n]] -> v_in [B,dims[0]], v_weights / v_biases (ACCUMULATE, double) */
void orc_mlp_bwd(int64_t B, int n_layers, const int *dims, const REAL *weights, const REAL *biases, const REAL *in,
                 const REAL *v_out, REAL *v_in, double *v_weights, double *v_biases) {
  int maxd = 0;
  int64_t nw = 0, nb = 0;
  for (int l = 0; l <= n_layers; ++l) if (dims[l] > maxd) maxd = dims[l];
  for (int l = 0; l < n_layers; ++l) { nw += (int64_t)dims[l] * dims[l + 1]; nb += dims[l + 1]; }
  /* threaded over the points (round 5: timed by bench.py's cpu_baseline leg): every thread accumulates its own weight / bias gradient and the
   * partial sums are added in turn (double sums: the order changes the result by ~1e-16 relative) */
#pragma omp parallel
  {
  REAL *h = (REAL *)malloc(sizeof(REAL) * (size_t)maxd * (n_layers + 1));
  REAL *g = (REAL *)malloc(sizeof(REAL) * maxd), *g2 = (REAL *)malloc(sizeof(REAL) * maxd);
  double *tw = v_weights ? (double *)calloc((size_t)nw, sizeof(double)) : NULL;
  double *tb = (v_biases && biases) ? (double *)calloc((size_t)nb, sizeof(double)) : NULL;
#pragma omp for
  for (int64_t b = 0; b < B; ++b) {
    memcpy(h, in + b * dims[0], sizeof(REAL) * dims[0]);
    const REAL *W = weights, *bi = biases;
    for (int l = 0; l < n_layers; ++l) {
      int I = dims[l], O = dims[l + 1];
      const REAL *cur = h + (size_t)l * maxd;
      REAL *nxt = h + (size_t)(l + 1) * maxd;
      for (int o = 0; o < O; ++o) {
        REAL acc = bi ? bi[o] : 0;
        for (int i = 0; i < I; ++i) acc += W[o * I + i] * cur[i];
        if (l < n_layers - 1 && acc < 0) acc = 0;
        nxt[o] = acc;
      }
      W += (int64_t)O * I;
      if (bi) bi += O;
    }
    memcpy(g, v_out + b * dims[n_layers], sizeof(REAL) * dims[n_layers]);
    int64_t woff = 0, boff = 0;
    for (int l = 0; l < n_layers; ++l) { woff += (int64_t)dims[l] * dims[l + 1]; boff += dims[l + 1]; }
    for (int l = n_layers - 1; l >= 0; --l) {
      int I = dims[l], O = dims[l + 1];
      woff -= (int64_t)O * I; boff -= O;
      const REAL *Wl = weights + woff;
      const REAL *hin = h + (size_t)l * maxd, *hout = h + (size_t)(l + 1) * maxd;
      if (l < n_layers - 1)
        for (int o = 0; o < O; ++o) if (!(hout[o] > 0)) g[o] = 0; /* ReLU mask */
      for (int i = 0; i < I; ++i) g2[i] = 0;
      for (int o = 0; o < O; ++o) {
        if (tb) tb[boff + o] += (double)g[o];
        for (int i = 0; i < I; ++i) {
          if (tw) tw[woff + (int64_t)o * I + i] += (double)(g[o] * hin[i]);
          g2[i] += Wl[o * I + i] * g[o];
        }
      }
      REAL *t = g; g = g2; g2 = t;
    }
    if (v_in) memcpy(v_in + b * dims[0], g, sizeof(REAL) * dims[0]);
  }
#pragma omp critical
  {
    if (tw) for (int64_t k = 0; k < nw; ++k) v_weights[k] += tw[k];
    if (tb) for (int64_t k = 0; k < nb; ++k) v_biases[k] += tb[k];
  }
  free(h); free(g); free(g2); free(tw); free(tb);
  }
}

/* Double backward of the decoder (the analytic eikonal term of the reference's DEFAULT configuration:
 * LocalMap::get_gradient, /root/reference/include/neural_net/local_map.cpp:151-172, torch::autograd::grad(create_graph=true)
 * through the torch::nn::Sequential of :29-42).  The first backward maps v_out to v_in = W_0^T D_0 W_1^T D_1 ... v_out
 * (D_l = ReLU mask of layer l's output).  Given vv_in = dL/d v_in this returns
 *   g_vout [B,dims[n]]  = dL/d v_out  = W_{n-1} D_{n-2} ... D_0 W_0 vv_in     (a masked, bias-free forward pass),
 *   g_weights (ACCUMULATE, double): dL/dW_0 = delta_0 (x) vv_in,  dL/dW_l = delta_l (x) (D_{l-1} t_{l-1}),  t_0 = W_0 vv_in,
 *                                   t_l = W_l D_{l-1} t_{l-1},  delta_l = the first backward's gradient at layer l's pre-activation.
 * A ReLU network is piecewise linear: nothing flows to the network input or to the biases. */
void orc_mlp_bwd_bwd(int64_t B, int n_layers, const int *dims, const REAL *weights, const REAL *biases, const REAL *in,
                     const REAL *v_out, const REAL *vv_in, REAL *g_vout, double *g_weights) {
  int maxd = 0;
  for (int l = 0; l <= n_layers; ++l) if (dims[l] > maxd) maxd = dims[l];
  REAL *h = (REAL *)malloc(sizeof(REAL) * (size_t)maxd * (n_layers + 1));      /* forward activations */
  REAL *dl = (REAL *)malloc(sizeof(REAL) * (size_t)maxd * (n_layers + 1));     /* delta_l at slot l+1 (pre-activation gradients) */
  REAL *t = (REAL *)malloc(sizeof(REAL) * maxd), *t2 = (REAL *)malloc(sizeof(REAL) * maxd);
  for (int64_t b = 0; b < B; ++b) {
    memcpy(h, in + b * dims[0], sizeof(REAL) * dims[0]);
    const REAL *W = weights, *bi = biases;
    for (int l = 0; l < n_layers; ++l) {
      int I = dims[l], O = dims[l + 1];
      const REAL *cur = h + (size_t)l * maxd;
      REAL *nxt = h + (size_t)(l + 1) * maxd;
      for (int o = 0; o < O; ++o) {
        REAL acc = bi ? bi[o] : 0;
        for (int i = 0; i < I; ++i) acc += W[o * I + i] * cur[i];
        if (l < n_layers - 1 && acc < 0) acc = 0;
        nxt[o] = acc;
      }
      W += (int64_t)O * I;
      if (bi) bi += O;
    }
    /* first backward: delta_{n-1} = v_out, delta_{l-1} = D_{l-1} W_l^T delta_l */
    memcpy(dl + (size_t)n_layers * maxd, v_out + b * dims[n_layers], sizeof(REAL) * dims[n_layers]);
    int64_t woff = 0;
    for (int l = 0; l < n_layers; ++l) woff += (int64_t)dims[l] * dims[l + 1];
    for (int l = n_layers - 1; l >= 1; --l) {
      int I = dims[l], O = dims[l + 1];
      woff -= (int64_t)O * I;
      const REAL *Wl = weights + woff, *d = dl + (size_t)(l + 1) * maxd, *hout = h + (size_t)l * maxd;
      REAL *dn = dl + (size_t)l * maxd;
      for (int i = 0; i < I; ++i) {
        REAL acc = 0;
        for (int o = 0; o < O; ++o) acc += Wl[o * I + i] * d[o];
        dn[i] = hout[i] > 0 ? acc : 0;
      }
    }
    /* second pass, forward: u = vv_in; dW_l += delta_l (x) u; u <- D_l (W_l u) */
    memcpy(t, vv_in + b * dims[0], sizeof(REAL) * dims[0]);
    woff = 0;
    for (int l = 0; l < n_layers; ++l) {
      int I = dims[l], O = dims[l + 1];
      const REAL *Wl = weights + woff, *d = dl + (size_t)(l + 1) * maxd, *hout = h + (size_t)(l + 1) * maxd;
      for (int o = 0; o < O; ++o) {
        REAL acc = 0;
        for (int i = 0; i < I; ++i) {
          if (g_weights) g_weights[woff + (int64_t)o * I + i] += (double)(d[o] * t[i]);
          acc += Wl[o * I + i] * t[i];
        }
        t2[o] = (l < n_layers - 1 && !(hout[o] > 0)) ? 0 : acc;
      }
      woff += (int64_t)O * I;
      REAL *sw = t; t = t2; t2 = sw;
    }
    if (g_vout) memcpy(g_vout + b * dims[n_layers], t, sizeof(REAL) * dims[n_layers]);
  }
  free(h); free(dl); free(t); free(t2);
}

/* SDF head (local_map.cpp:97-102): out [B,2] -> sdf [B], isigma [B] ; softplus beta=100, threshold 20 (torch) */
void orc_sdf_head(int64_t B, REAL inv_bce_sigma, const REAL *out, REAL *sdf, REAL *isigma) {
  for (int64_t b = 0; b < B; ++b) {
    sdf[b] = out[2 * b];
    REAL z = out[2 * b + 1] * 100;
    REAL sp = z > 20 ? out[2 * b + 1] : (REAL)(log1p(exp((double)z)) / 100.0);
    isigma[b] = 1 + sp * inv_bce_sigma;
  }
}

/* K1 distCUDA2: mean of the squared distances to the 3 nearest neighbours (self excluded), brute force */
void orc_knn_mean_dist2(int64_t N, const REAL *pts, REAL *out) {
#pragma omp parallel for
  for (int64_t i = 0; i < N; ++i) {
    double best[3] = {1e300, 1e300, 1e300};
    for (int64_t j = 0; j < N; ++j) {
      if (j == i) continue;
      double dx = (double)pts[3 * i] - pts[3 * j], dy = (double)pts[3 * i + 1] - pts[3 * j + 1],
             dz = (double)pts[3 * i + 2] - pts[3 * j + 2];
      double d = dx * dx + dy * dy + dz * dz;
      if (d < best[2]) {
        if (d < best[0]) { best[2] = best[1]; best[1] = best[0]; best[0] = d; }
        else if (d < best[1]) { best[2] = best[1]; best[1] = d; }
        else best[2] = d;
      }
    }
    int cnt = N - 1 < 3 ? (int)(N - 1) : 3;
    double s = 0;
    for (int k = 0; k < cnt; ++k) s += best[k];
    out[i] = (REAL)(N > 1 ? s / 3.0 : 0.0);
  }
}
