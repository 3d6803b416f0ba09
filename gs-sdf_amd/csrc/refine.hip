// refine.hip — a18: one refinement step of the splat set on the trainer's FIELD-MAJOR flat buffers, in two passes over the rows.
// Replaces, per refinement step, the reference's three re-materialisations of every parameter and of both Adam moments
//   NeuralGS::grow_gs -> duplicate -> split, prune_gs, zero_state   /root/reference/include/neural_gaussian/neural_gaussian.cpp:690-890
//   prune / cat / prune_cat_tensors_to_optimizer                    /root/reference/include/optimizer/optimizer_utils/optimizer_utils.cpp:5-165
// (~40 index_select / cat launches per stage and three `.sum().item<int>()` host round trips, neural_gaussian.cpp:723,763,831).
//
// The three stages compose into ONE row map.  With d_i = duplicate, s_i = split (both decided on the OLD rows, :690-738), p_i = prune test of
// row i's own values and q_i = prune test of its split children (same opacity, scale / 1.6; :856-890, applied to the set AFTER growing), the
// reference's final set is, in this order:
//   A  the old rows that are neither split nor pruned                         (their Adam moments travel with them)
//   B  the copies of the duplicated rows that are not pruned (index order)    (zero moments: optimizer_utils.cpp cat_tensors_to_optimizer)
//   C  the first  child of every split row that is not pruned                 (zero moments)
//   D  the second child of every split row that is not pruned                 (zero moments)
// plan: flags + counts per row for A, B, C and the rank of a row among the split rows (the reference draws randn [2, n_split, 3] for ALL split
//       rows, pruned children included); one scan gives every destination; four totals land in host-visible words (no .item()).
// apply: every old row writes its surviving images — parameters, both moments, anchor, the densification statistics — straight into the new
//       buffers: each byte of the old set is read once and each byte of the new set written once.
// mode 1 = prune by a caller-supplied mask (prune_invisible_gs :892-905, prune_nan_gs :907-916): segment A only.
#include "common.h"
#include "scan.h"

namespace gsdf {

static constexpr int RF_FIELDS = 6;   // offsets 3, scaling 3, quaternion 4, opacity 1, features_dc 3, features_rest 3 r
struct RefineLayout {
  int width[RF_FIELDS];
  int64_t n;   // rows of the old set
};
struct RefineThresholds {
  float grow_grad2d, grow_scale3d, grow_scale2d, prune_opa, prune_scale_min, prune_scale3d;
  int use_radii, use_prune_scale3d;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// flags of row i: bit 0 = A (kept), bit 1 = B (surviving copy), bit 2 = C/D (surviving children), bit 3 = split
__device__ __forceinline__ unsigned refine_flags(const RefineThresholds &t, float grad2d, float count, float radius, float s0, float s1, float opac_logit) {
  const float grads = grad2d / fmaxf(count, 1.0f);
  const bool high = grads > t.grow_grad2d;
  const float e0 = expf(s0), e1 = expf(s1);
  const bool small = fmaxf(e0, e1) <= t.grow_scale3d;
  const bool dupli = high && small;
  bool split = high && !small;
  if (t.use_radii) split = split || radius > t.grow_scale2d;
  const bool opa_low = sigmoid_f(opac_logit) < t.prune_opa;
  bool p = opa_low || fminf(e0, e1) < t.prune_scale_min;
  if (t.use_prune_scale3d) p = p || fmaxf(e0, e1) > t.prune_scale3d;
  // children: scaling = log(scale * (1 / 1.6)) (torch divides by a scalar through its reciprocal), read back through exp like any row
  const float inv16 = 1.0f / 1.6f;
  const float c0 = expf(logf(e0 * inv16)), c1 = expf(logf(e1 * inv16));
  bool q = opa_low || fminf(c0, c1) < t.prune_scale_min;
  if (t.use_prune_scale3d) q = q || fmaxf(c0, c1) > t.prune_scale3d;
  return (!split && !p ? 1u : 0u) | (dupli && !p ? 2u : 0u) | (split && !q ? 4u : 0u) | (split ? 8u : 0u);
}

// counts: segment-major [4][n] = A, B, C, split
__global__ void __launch_bounds__(256)
    refine_plan_kernel(RefineLayout L, RefineThresholds t, int mode, const uint8_t *__restrict__ mask, const float *__restrict__ flat,
                       const float *__restrict__ grad2d, const float *__restrict__ count, const float *__restrict__ radii,
                       int32_t *__restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L.n) return;
  unsigned f;
  if (mode == 1) {
    f = mask[i] ? 0u : 1u;
  } else {
    const float *scaling = flat + (int64_t)L.width[0] * L.n;
    const float *opacity = scaling + (int64_t)(L.width[1] + L.width[2]) * L.n;
    f = refine_flags(t, grad2d[i], count[i], radii ? radii[i] : 0.f, scaling[3 * i], scaling[3 * i + 1], opacity[i]);
  }
  counts[i] = (int)(f & 1u);
  counts[L.n + i] = (int)((f >> 1) & 1u);
  counts[2 * L.n + i] = (int)((f >> 2) & 1u);
  counts[3 * L.n + i] = (int)((f >> 3) & 1u);
}

// totals[0..3] = nA, nB, nC, n_split from the inclusive scan (device or host-visible words)
__global__ void refine_totals_kernel(int64_t n, const int64_t *__restrict__ incl, int64_t *__restrict__ totals) {
  const int k = threadIdx.x;
  if (k < 4) {
    const int64_t hi = incl[(int64_t)(k + 1) * n - 1], lo = k ? incl[(int64_t)k * n - 1] : 0;
    store_host_visible(totals + k, hi - lo);
  }
}

struct RefineSrc {
  const float *flat, *m, *v, *anchors, *state[4];   // state: grad2d, count, vis, radii (any may be NULL)
};
struct RefineDst {
  float *flat, *m, *v, *anchors, *state[4];
};

__device__ __forceinline__ void copy_row(const RefineLayout &L, int64_t n_dst, const float *__restrict__ src, float *__restrict__ dst, int64_t i, int64_t j) {
  int64_t so = 0, dof = 0;
#pragma unroll
  for (int f = 0; f < RF_FIELDS; ++f) {
    const int w = L.width[f];
    for (int c = 0; c < w; ++c) dst[dof + j * w + c] = src[so + i * w + c];
    so += (int64_t)w * L.n; dof += (int64_t)w * n_dst;
  }
}
__device__ __forceinline__ void zero_row(const RefineLayout &L, int64_t n_dst, float *__restrict__ dst, int64_t j) {
  int64_t dof = 0;
#pragma unroll
  for (int f = 0; f < RF_FIELDS; ++f) {
    const int w = L.width[f];
    for (int c = 0; c < w; ++c) dst[dof + j * w + c] = 0.f;
    dof += (int64_t)w * n_dst;
  }
}

// One lane per old row.  (Rows are 14 + 3 r floats in 6 field segments: consecutive lanes write consecutive rows of every segment.)
__global__ void __launch_bounds__(256)
    refine_apply_kernel(RefineLayout L, int64_t n_dst, int64_t nA, int64_t nB, int64_t nC, int64_t n_split, const int32_t *__restrict__ counts,
                        const int64_t *__restrict__ incl, const float *__restrict__ randn, RefineSrc S, RefineDst D) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L.n) return;
  const int64_t n = L.n;
  const bool a = counts[i] != 0, b = counts[n + i] != 0, c = counts[2 * n + i] != 0;
  // exclusive positions inside the segments (the scan runs over the segment-major array: subtract what the earlier segments hold)
  const int64_t jA = incl[i] - (a ? 1 : 0);
  const int64_t jB = nA + (incl[n + i] - nA) - (b ? 1 : 0);
  const int64_t jC = nA + nB + (incl[2 * n + i] - nA - nB) - (c ? 1 : 0);
  const int64_t jD = jC + nC;
  auto carry_aux = [&](int64_t j) {
    D.anchors[3 * j] = S.anchors[3 * i]; D.anchors[3 * j + 1] = S.anchors[3 * i + 1]; D.anchors[3 * j + 2] = S.anchors[3 * i + 2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (D.state[k]) D.state[k][j] = S.state[k][i];
  };
  if (a) {
    copy_row(L, n_dst, S.flat, D.flat, i, jA);
    if (D.m) { copy_row(L, n_dst, S.m, D.m, i, jA); copy_row(L, n_dst, S.v, D.v, i, jA); }
    carry_aux(jA);
  }
  if (b) {
    copy_row(L, n_dst, S.flat, D.flat, i, jB);
    if (D.m) { zero_row(L, n_dst, D.m, jB); zero_row(L, n_dst, D.v, jB); }
    carry_aux(jB);
  }
  if (c) {
    // NeuralGS::split (:767-827): scales = (exp s0, exp s1, 0); offsets_child = R(q / |q|) (scales * (scales * randn)) + offsets;
    // scaling_child = log(scales / 1.6) (third: log 0 = -inf, the disc has no extent along its normal); the rest is the parent's
    const int64_t rank = incl[3 * n + i] - nA - nB - nC - 1;      // this row among the split rows
    const float *off = S.flat, *scl = S.flat + (int64_t)L.width[0] * n, *qt = scl + (int64_t)L.width[1] * n;
    const float e0 = expf(scl[3 * i]), e1 = expf(scl[3 * i + 1]);
    float qw = qt[4 * i], qx = qt[4 * i + 1], qy = qt[4 * i + 2], qz = qt[4 * i + 3];
    const float nrm = fmaxf(sqrtf(qw * qw + qx * qx + qy * qy + qz * qz), 1e-12f);
    qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
    const float R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                        2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                        2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
    const float inv16 = 1.0f / 1.6f;
    const float cs[3] = {logf(e0 * inv16), logf(e1 * inv16), logf(0.0f * inv16)};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t j = k ? jD : jC;
      copy_row(L, n_dst, S.flat, D.flat, i, j);
      const float *r = randn + ((int64_t)k * n_split + rank) * 3;
      const float v0 = e0 * (e0 * r[0]), v1 = e1 * (e1 * r[1]);     // third component: scale 0
      float *doff = D.flat, *dscl = D.flat + (int64_t)L.width[0] * n_dst;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        doff[3 * j + x] = (R[3 * x] * v0 + R[3 * x + 1] * v1) + off[3 * i + x];
        dscl[3 * j + x] = cs[x];
      }
      if (D.m) { zero_row(L, n_dst, D.m, j); zero_row(L, n_dst, D.v, j); }
      carry_aux(j);
    }
  }
}

}  // namespace gsdf

using namespace gsdf;

static int refine_layout(const gsdf_refine_args *g, RefineLayout *L, RefineThresholds *t, const char *who) {
  GSDF_REQUIRE(g != nullptr, "%s: null arguments", who);
  GSDF_REQUIRE(g->n >= 0 && g->n < ((int64_t)1 << 31) / 4, "%s: row count %lld out of range", who, (long long)g->n);
  GSDF_REQUIRE(g->n_rest_cols >= 0 && g->n_rest_cols % 3 == 0 && g->n_rest_cols <= 3 * 24, "%s: features_rest columns must be 3 r", who);
  GSDF_REQUIRE(g->mode == 0 || g->mode == 1, "%s: mode %d", who, g->mode);
  const int w[RF_FIELDS] = {3, 3, 4, 1, 3, g->n_rest_cols};
  for (int f = 0; f < RF_FIELDS; ++f) L->width[f] = w[f];
  L->n = g->n;
  t->grow_grad2d = g->grow_grad2d; t->grow_scale3d = g->grow_scale3d; t->grow_scale2d = g->grow_scale2d; t->prune_opa = g->prune_opa;
  t->prune_scale_min = g->prune_scale_min; t->prune_scale3d = g->prune_scale3d; t->use_radii = g->use_radii; t->use_prune_scale3d = g->use_prune_scale3d;
  return GSDF_OK;
}

extern "C" size_t gsdf_refine_ws_bytes(int64_t n) { return scan_ws_bytes(4 * n) + 256; }

extern "C" int gsdf_refine_plan(const gsdf_refine_args *g, int32_t *counts, int64_t *offsets_incl, int64_t *totals, void *ws, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_refine_plan");
  RefineLayout L;
  RefineThresholds t;
  if (int rc = refine_layout(g, &L, &t, "refine_plan")) return rc;
  GSDF_REQUIRE(totals != nullptr, "refine_plan: null totals");
  if (L.n == 0) {
    GSDF_HIP(hipMemsetAsync(totals, 0, 4 * sizeof(int64_t), stream), "refine_plan memset");
    return GSDF_OK;
  }
  GSDF_REQUIRE(counts && offsets_incl && ws && g->flat, "refine_plan: null buffer");
  GSDF_REQUIRE(g->mode == 1 ? g->mask != nullptr : (g->grad2d && g->count && (!g->use_radii || g->radii)), "refine_plan: statistics / mask missing");
  refine_plan_kernel<<<(unsigned)((L.n + 255) / 256), 256, 0, stream>>>(L, t, g->mode, g->mask, g->flat, g->grad2d, g->count, g->use_radii ? g->radii : nullptr, counts);
  GSDF_CHECK_LAUNCH("refine_plan_kernel");
  // the grand total goes to the workspace (unused: the four segment totals are what the caller needs)
  int64_t *grand = (int64_t *)((char *)ws + scan_ws_bytes(4 * L.n));
  if (int rc = scan_inclusive_i32_i64(counts, offsets_incl, 4 * L.n, ws, grand, stream)) return rc;
  refine_totals_kernel<<<1, 64, 0, stream>>>(L.n, offsets_incl, totals);
  GSDF_CHECK_LAUNCH("refine_totals_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_refine_apply(const gsdf_refine_args *g, const int32_t *counts, const int64_t *offsets_incl, const int64_t *totals_host,
                                 const float *randn, float *flat_new, float *m_new, float *v_new, float *anchors_new, float *const *state_new,
                                 gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_refine_apply");
  RefineLayout L;
  RefineThresholds t;
  if (int rc = refine_layout(g, &L, &t, "refine_apply")) return rc;
  if (L.n == 0) return GSDF_OK;
  GSDF_REQUIRE(totals_host != nullptr, "refine_apply: null totals");
  const int64_t nA = totals_host[0], nB = totals_host[1], nC = totals_host[2], nS = totals_host[3];
  GSDF_REQUIRE(nA >= 0 && nA <= L.n && nB >= 0 && nB <= L.n && nC >= 0 && nC <= nS && nS <= L.n, "refine_apply: implausible totals");
  const int64_t n_dst = nA + nB + 2 * nC;
  if (n_dst == 0) return GSDF_OK;
  GSDF_REQUIRE(counts && offsets_incl && flat_new && anchors_new && g->flat && g->anchors && (nC == 0 || randn), "refine_apply: null buffer");
  GSDF_REQUIRE((m_new == nullptr) == (v_new == nullptr) && (m_new == nullptr || (g->adam_m && g->adam_v)), "refine_apply: Adam moments come in pairs");
  RefineSrc S{g->flat, g->adam_m, g->adam_v, g->anchors, {g->grad2d, g->count, g->vis, g->radii}};
  RefineDst D{flat_new, m_new, v_new, anchors_new, {nullptr, nullptr, nullptr, nullptr}};
  for (int k = 0; k < 4; ++k) {
    D.state[k] = state_new ? state_new[k] : nullptr;
    GSDF_REQUIRE(D.state[k] == nullptr || S.state[k] != nullptr, "refine_apply: a statistic is carried over only when the old one exists");
  }
  refine_apply_kernel<<<(unsigned)((L.n + 255) / 256), 256, 0, stream>>>(L, n_dst, nA, nB, nC, nS, counts, offsets_incl, randn, S, D);
  GSDF_CHECK_LAUNCH("refine_apply_kernel");
  return GSDF_OK;
}
