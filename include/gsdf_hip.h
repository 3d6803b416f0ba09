/*
 * gsdf_hip.h — C ABI of the MI355X-native GS-SDF hot path (libgsdf_hip.so).
 *
 * This is the drop-in boundary.  Each entry point replaces one operator of the reference's
 * un-vendored CUDA submodules, as called from the reference's host code (citations are
 * /root/reference/<file>:<line>).  The reference-side binding a maintainer adds is the thin
 * libtorch wrapper shown in INTEGRATION.md (gs-sdf_amd/host/ ships it): same C++ names and
 * argument order as the reference's call sites, converting torch::Tensor <-> raw pointers.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named *_host;
 *     all floating data is fp32, row-major, contiguous; ids are int32/int64 as documented;
 *   - no allocation inside: the caller owns every buffer, workspaces are sized by *_ws_bytes();
 *   - dynamic sizes (M = visible splats, I = tile intersections) are produced by a "count" call
 *     that writes the size to device memory; the caller reads it back and allocates (the same
 *     two host syncs the reference has: neural_gaussian.cpp:188-209);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing syncs;
 *   - return value: 0 = ok, negative = error (gsdf_last_error() gives the message, thread-local);
 *   - re-entrant, no global state; gradient outputs documented "accumulate" must be zeroed by
 *     the caller, all others are fully overwritten.
 */
#ifndef GSDF_HIP_H
#define GSDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSDF_OK 0
#define GSDF_ERR_INVALID_ARG (-1)
#define GSDF_ERR_LAUNCH (-2)
#define GSDF_ERR_UNSUPPORTED (-3)

typedef void *gsdf_stream_t;

const char *gsdf_last_error(void);
/* ABI version; bumped on any signature change.  Every binding compares gsdf_abi_version() of the library it loaded with the
 * GSDF_ABI_VERSION of the header it was written against and refuses to run on a mismatch (gs_sdf_amd/capi.py: lib(); the C++
 * operator layer: gsplat_ops.cpp static initialiser): a stale libgsdf_hip.so fails at load, not on the device. */
#define GSDF_ABI_VERSION 10
int gsdf_abi_version(void);

/* Deterministic mode (process-wide — a step's backward kernels are launched by the autograd engine's thread, not by the caller's; on != 0 switches
 * it on, on == 0 off, on < 0 only asks; returns the previous setting).  While it is on, the entry points accumulate in an order-independent way, so that
 * two runs from the same state produce the same bits: the compositing backward sums its records in 64-bit fixed point (unit 2^-34 of the
 * launch's largest upstream gradient; a tile sum of 2^13 times that maximum or more poisons the outputs with NaN), the loss values are reduced in
 * a fixed order (one device-global slot per kernel: such a kernel must not run on two streams at once), the decoder's weight gradients leave
 * through per-wave partial buffers.  The table scatter is order-independent in either mode; the atomic fall-backs for tiny batches
 * (gsdf_hashgrid_bwd, the fp32-pipe decoder kernels) and the float atomics that sum a splat's gradient over SEVERAL cameras of one call (C > 1 in
 * the projection / view-colour backward, densify statistics) are not covered: the host layers route around the former while the mode is on, the
 * training loop renders one camera per call.  The integer
 * outputs and every forward pass are deterministic in either mode.  Costs about 2 ms per step at the headline workload: a validation mode. */
int gsdf_deterministic(int on);

/* Optional per-entry-point device timing (bench.py's roofline leg, for callers in any language): between gsdf_timing_begin and
 * gsdf_timing_end every timed entry point records a HIP event pair on ITS OWN stream around everything it launches (an operator
 * such as gsdf_hashgrid_bwd_binned2 is several kernels).  only_csv: NULL = all, else a comma-separated list of entry-point names.
 * gsdf_timing_end stops collecting, waits for the recorded events and writes one line per entry point,
 * "name calls total_ms min_ms max_ms median_ms\n", into buf (NUL-terminated, truncated to cap); returns the bytes the full report needs.
 * ~10 us of host time per timed call; nothing is recorded (one atomic load per call) while timing is off. */
int gsdf_timing_begin(const char *only_csv);
size_t gsdf_timing_end(char *buf, size_t cap);
/* the same collection as a device timeline: one line per timed call in call order, "name begin_ms end_ms" relative to the first call's begin event
 * (stops the timing like gsdf_timing_end; returns the bytes needed) */
size_t gsdf_timing_trace(char *buf, size_t cap);

/* Host-visible count words.  The packed operators hand the HOST a size between two launches (P1: visible splats, P3: intersections, the
 * visible set of the joint iteration): a device scalar read back with a copy + a stream synchronisation costs 30-50 us of idle queue per
 * read on this path (profiles/r04_bench_cfg3_step_timeline.txt).  Any `int64_t *` count output of this ABI (n_visible, n_isects, n_out ...)
 * may instead point at words of pinned, device-mapped, fine-grained host memory: the kernel's store lands in host memory, the host arms the
 * word with a sentinel before the launch and polls it — no copy kernel, no synchronisation, and whatever the host queues after the launch
 * runs while it waits.  gsdf_host_words_alloc returns the same words under both addresses (host view, device view); free with the host view. */
int gsdf_host_words_alloc(int n_words, int64_t **host_view, int64_t **device_view);
int gsdf_host_words_free(int64_t *host_view);

/* ------------------------------------------------------------------------------------------
 * P1  fully_fused_projection_2dgs(means, quats, scales, viewmats, Ks, W, H, near, far,
 *                                 radius_clip, packed=true, sparse_grad=false)
 *     reference call: include/neural_gaussian/neural_gaussian.cpp:188-192
 * Two phases (packed mode): cull -> (host reads M) -> fill.
 * ---------------------------------------------------------------------------------------- */
size_t gsdf_projection_2dgs_ws_bytes(int64_t n_gauss, int64_t n_cams);

/* radii_dense int32[C*N] (0 = culled), ws (>= ws_bytes), n_visible int64[1] (device). */
int gsdf_projection_2dgs_cull(int64_t n_gauss, int64_t n_cams, const float *means, const float *quats,
                              const float *scales, const float *viewmats, const float *Ks, int width,
                              int height, float near_plane, float far_plane, float radius_clip,
                              int32_t *radii_dense, void *ws, int64_t *n_visible, gsdf_stream_t stream);

/* Packed outputs, M rows in increasing (camera, gaussian) order:
 * camera_ids i64[M], gaussian_ids i64[M], radii i32[M], means2d [M,2], depths [M],
 * ray_transforms [M,3,3] (rows M_u, M_v, M_w of K*[s_u t_u, s_v t_v, mu_c]), normals [M,3]
 * (camera space, facing the camera), samples [M,3], samples_weights [M,1].
 * sample_seed == 0: samples = centres, weights = 1 (the `center_reg` mode,
 * neural_gaussian.cpp:259-265); otherwise one hashed N(0,1)^2 point on the splat disk. */
int gsdf_projection_2dgs_fill(int64_t n_gauss, int64_t n_cams, const float *means, const float *quats,
                              const float *scales, const float *viewmats, const float *Ks, int width,
                              int height, uint64_t sample_seed, const int32_t *radii_dense, const void *ws,
                              int64_t n_visible, int64_t *camera_ids, int64_t *gaussian_ids, int32_t *radii,
                              float *means2d, float *depths, float *ray_transforms, float *normals,
                              float *samples, float *samples_weights, gsdf_stream_t stream);

/* VJP (implicit via autograd in the reference).  v_samples and v_depths may be NULL (= zero).  Dense outputs
 * v_means [N,3], v_quats [N,4], v_scales [N,3]: ACCUMULATE (zero them first). */
int gsdf_projection_2dgs_bwd(int64_t n_gauss, int64_t n_cams, int64_t n_visible, const float *means,
                             const float *quats, const float *scales, const float *viewmats, const float *Ks,
                             int width, int height, uint64_t sample_seed, const int64_t *camera_ids,
                             const int64_t *gaussian_ids, const float *v_means2d, const float *v_depths,
                             const float *v_ray_transforms, const float *v_normals, const float *v_samples,
                             float *v_means, float *v_quats, float *v_scales, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * P2  gsplat_cpp::get_view_colors(viewmats, means, radii, colors, camera_ids, gaussian_ids,
 *                                 sh_degree)      reference call: neural_gaussian.cpp:199-200
 *     rgb = max(SH_deg(dir) . coeffs + 0.5, 0), dir = normalise(mean - campos); sh [N,K,3].
 * ---------------------------------------------------------------------------------------- */
int gsdf_view_colors_fwd(int64_t n_visible, int64_t n_sh_bases, int sh_degree, const float *viewmats,
                         const float *means, const float *sh_coeffs, const int64_t *camera_ids,
                         const int64_t *gaussian_ids, float *colors, gsdf_stream_t stream);
/* v_sh [N,K,3], v_means [N,3]: ACCUMULATE.  unique_gaussians != 0 (single camera) allows plain
 * read-modify-write instead of atomics. */
int gsdf_view_colors_bwd(int64_t n_visible, int64_t n_sh_bases, int sh_degree, const float *viewmats,
                         const float *means, const float *sh_coeffs, const int64_t *camera_ids,
                         const int64_t *gaussian_ids, const float *v_colors, float *v_sh, float *v_means,
                         int unique_gaussians, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * P3  gsplat_cpp::tile_encode(W, H, tile, means2d, radii, depths, packed, C, camera_ids,
 *                             gaussian_ids)        reference call: neural_gaussian.cpp:207-209
 * count -> (host reads I) -> encode.  Integer outputs are the bit-exact parity target.
 * ---------------------------------------------------------------------------------------- */
size_t gsdf_tile_count_ws_bytes(int64_t n_visible);
/* tiles_per_gauss i32[M], cum_tiles i64[M] (inclusive scan), n_isects i64[1] (device). */
int gsdf_tile_count(int64_t n_visible, int width, int height, int tile_size, const float *means2d,
                    const int32_t *radii, int32_t *tiles_per_gauss, int64_t *cum_tiles, void *ws,
                    int64_t *n_isects, gsdf_stream_t stream);
size_t gsdf_tile_encode_ws_bytes(int64_t n_visible, int64_t n_isects);
/* isect_ids i64[I] sorted keys (cam | tile | fp32 depth bits), flatten_ids i32[I],
 * isect_offsets i32[C*tile_h*tile_w]. */
int gsdf_tile_encode(int64_t n_visible, int64_t n_cams, int64_t n_isects, int width, int height, int tile_size,
                     const float *means2d, const int32_t *radii, const float *depths,
                     const int64_t *camera_ids, const int64_t *cum_tiles, void *ws, int64_t *isect_ids,
                     int32_t *flatten_ids, int32_t *isect_offsets, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * P4  rasterize_to_pixels_2dgs(means2d, ray_transforms, colors, opacities, normals, densify,
 *                              W, H, tile, isect_offsets, flatten_ids, backgrounds, masks,
 *                              packed, means2d_absgrad, distloss)
 *     reference call: neural_gaussian.cpp:215-223 ; grads consumed at :626-633
 * tile_size must be 16.  backgrounds [C,3] / masks u8[C,th,tw] may be NULL.
 * ws >= gsdf_rasterize_2dgs_fwd_ws_bytes(M, I) (ABI 8): the forward first packs every visible splat into one 128-byte record and
 * every (tile, splat) pair into a 64-bit reach mask (csrc/raster_quad.h) and leaves both there; handing the same buffer to the
 * backward as `fwd_ws` saves it the two passes.
 * ---------------------------------------------------------------------------------------- */
size_t gsdf_rasterize_2dgs_fwd_ws_bytes(int64_t n_visible, int64_t n_isects);
int gsdf_rasterize_2dgs_fwd(int64_t n_cams, int64_t n_visible, int64_t n_isects, int width, int height,
                            int tile_size, const float *means2d, const float *ray_transforms,
                            const float *colors, const float *opacities, const float *normals,
                            const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                            const int32_t *flatten_ids, float *render_colors /*[C,H,W,3]*/,
                            float *render_depths /*[C,H,W,1]*/, float *render_alphas /*[C,H,W,1]*/,
                            float *render_normals /*[C,H,W,3]*/, float *render_median /*[C,H,W,1]*/,
                            int32_t *last_ids /*[C,H,W]*/, int32_t *median_ids /*[C,H,W]*/,
                            float *visibilities /*[M,1], fully written*/,
                            float *final_T /*[C,H,W] or NULL: the transmittance after the last blended splat, saved for the
                                             backward (render_alphas = 1 - T loses it to rounding once T << 1)*/,
                            void *ws, gsdf_stream_t stream);

/* Instrumented launches of the compositing kernels (tests / diagnostics; the product never passes an instr block).  No process-wide
 * state: the instrumentation is an argument of the call.
 *   counters: device pointer to 16 zero-initialised uint64, the launch ADDS [0] (wave, splat) visits after the per-quadrant reach mask,
 *     [1] lanes of those visits whose pixel is still live, [2] lanes passing the alpha test, [3] lanes that blend (forward); [4] visits,
 *     [5] lanes replaying, [6] lanes blending (backward); [7] forward visits in which no lane passed the alpha test; [8..10] what-if
 *     iteration counts of tools/exp_raster_pairs.py.
 *   trace_rows / trace_stride / trace_bits (forward only): the DECISION RECORD of the parity gate (tests/util.py).  trace_rows int32
 *     [C,H,W]: row of the record of that pixel or -1; trace_bits uint8 [rows, trace_stride], zeroed by the caller: byte k of a row =
 *     the decisions the kernel took for the pixel at position k of its tile's list: bit0 blended, bit1 3-D footprint branch (g3 <= g2),
 *     bit2 alpha clamped at 0.999, bit3 the pixel terminates at this pair (T (1 - alpha) <= 1e-4; not blended), bit4 the median is
 *     updated here (T > 0.5); 0 = the pair does not contribute (alpha < 1/255, or after termination).
 * counters and the trace are separate launches (one of the two per call). */
typedef struct gsdf_raster_instr {
  unsigned long long *counters;
  const int32_t *trace_rows;
  int32_t trace_stride;
  uint8_t *trace_bits;
} gsdf_raster_instr;
int gsdf_rasterize_2dgs_fwd_instr(int64_t n_cams, int64_t n_visible, int64_t n_isects, int width, int height,
                                  int tile_size, const float *means2d, const float *ray_transforms, const float *colors,
                                  const float *opacities, const float *normals, const float *backgrounds, const uint8_t *masks,
                                  const int32_t *isect_offsets, const int32_t *flatten_ids, float *render_colors,
                                  float *render_depths, float *render_alphas, float *render_normals, float *render_median,
                                  int32_t *last_ids, int32_t *median_ids, float *visibilities, float *final_T, void *ws,
                                  const gsdf_raster_instr *instr /*host pointer*/, gsdf_stream_t stream);

/* All gradient outputs are fully written.  v_means2d_abs may be NULL.  ws >= *_bwd_ws_bytes(M, I): the kernel
 * accumulates one packed 84-byte gradient record per splat there (line-coalesced atomics) and unpacks it.
 * fwd_ws: the workspace the forward of the SAME inputs filled (records + reach masks), or NULL — the backward then runs the pack and
 * mask passes into its own workspace.
 * final_T: the forward's saved transmittance, or NULL (then T_final = 1 - render_alphas as upstream gsplat does, which is
 * only accurate to 6e-8 ABSOLUTE: a 6e-4 relative error of every weight of a pixel that ended at T = 1e-4). */
size_t gsdf_rasterize_2dgs_bwd_ws_bytes(int64_t n_visible, int64_t n_isects);
int gsdf_rasterize_2dgs_bwd(int64_t n_cams, int64_t n_visible, int64_t n_isects, int width, int height,
                            int tile_size, const float *means2d, const float *ray_transforms,
                            const float *colors, const float *opacities, const float *normals,
                            const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                            const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                            const int32_t *median_ids, const float *v_render_colors,
                            const float *v_render_depths, const float *v_render_alphas,
                            const float *v_render_normals, const float *v_render_median, float *v_means2d,
                            float *v_ray_transforms, float *v_colors, float *v_opacities, float *v_normals,
                            float *v_densify, float *v_means2d_abs, void *ws, const float *final_T, const void *fwd_ws,
                            gsdf_stream_t stream);
/* the same with instr->counters (see gsdf_raster_instr; the trace is a forward-only facility) */
int gsdf_rasterize_2dgs_bwd_instr(int64_t n_cams, int64_t n_visible, int64_t n_isects, int width, int height,
                                  int tile_size, const float *means2d, const float *ray_transforms,
                                  const float *colors, const float *opacities, const float *normals,
                                  const float *backgrounds, const uint8_t *masks, const int32_t *isect_offsets,
                                  const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                                  const int32_t *median_ids, const float *v_render_colors,
                                  const float *v_render_depths, const float *v_render_alphas,
                                  const float *v_render_normals, const float *v_render_median, float *v_means2d,
                                  float *v_ray_transforms, float *v_colors, float *v_opacities, float *v_normals,
                                  float *v_densify, float *v_means2d_abs, void *ws, const float *final_T, const void *fwd_ws,
                                  const gsdf_raster_instr *instr /*host pointer*/, gsdf_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * a3  per-pixel epilogue of rasterization_2dgs_sdf (include/neural_gaussian/neural_gaussian.cpp:229-240):
 *     renders [n,4] = cat(colours, expected_depth ? nan_to_num(depth/alpha) : depth);
 *     normals_world [n,3] = normals @ inverse(viewmats)[0,:3,:3]^T   (camera 0's pose, as the reference does).
 * viewmat0: the first 4x4 world->camera matrix (device).  n_pix = C*H*W.
 * ---------------------------------------------------------------------------------------- */
int gsdf_render_post_fwd(int64_t n_pix, int expected_depth, const float *viewmat0, const float *render_colors,
                         const float *render_depths, const float *render_alphas, const float *render_normals,
                         float *renders, float *normals_world, float *color3 /*[n,3] or NULL*/,
                         float *depth1 /*[n,1] or NULL: the two slices the caller takes of `renders`, :533-543*/,
                         gsdf_stream_t stream);
/* any of the four upstream gradients may be NULL (treated as zero) */
int gsdf_render_post_bwd(int64_t n_pix, int expected_depth, const float *viewmat0, const float *render_depths,
                         const float *render_alphas, const float *v_renders, const float *v_normals_world,
                         const float *v_color3, const float *v_depth1, float *v_render_colors, float *v_render_depths, float *v_render_alphas,
                         float *v_render_normals, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * S1  TCNNEncoding::forward — multiresolution hash grid (tiny-cuda-nn "Grid"/"Hash"/"Linear")
 *     reference: include/neural_net/encoding_map.cpp:15-26 (config {n_levels 16, n_features_per_level 2,
 *     log2_hashmap_size 19, base_resolution 32, per_level_scale 2.0}), :59 (forward);
 *     second order (autograd::grad with create_graph) at include/neural_net/local_map.cpp:151-172.
 * x [B,3] in [0,1]; table fp32 [n_entries, 2] (levels concatenated); feat [B, n_levels*2].
 * n_features_per_level must be 2 and n_levels <= 16.
 * ---------------------------------------------------------------------------------------- */
/* host helper: entry offsets per level, offsets_host[n_levels+1]; returns total entries (-1 on error). */
int64_t gsdf_hashgrid_offsets(int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                              int64_t *offsets_host);
int gsdf_hashgrid_fwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                      const float *x, const float *table, float *feat, gsdf_stream_t stream);
/* First-order fast path for d/dx: the forward also stores jac [B, n_levels*n_feat, 3] = d feat / d x (384 B per point at
 * 16x2), and v_x[b] = sum_k v_feat[b][k] * jac[b][k] is then a dense contraction (no second pass over the table). */
int gsdf_hashgrid_fwd_jac(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                          const float *x, const float *table, float *feat, float *jac, gsdf_stream_t stream);
/* The same with the Jacobian of the first jac_rows rows only (jac [jac_rows, n_levels*n_feat, 3]): a batch whose tail is
 * the central-difference stencil of its head needs d/dx of the head alone (neural_mapping.cpp:436-451). */
int gsdf_hashgrid_fwd_jac_rows(int64_t B, int64_t jac_rows, int n_levels, int n_feat, int log2_hashmap, int base_res,
                               float per_level_scale, const float *x, const float *table, float *feat, float *jac,
                               gsdf_stream_t stream);
/* The same for a stencil batch: x = stencil_n base rows followed by the 6 blocks of stencil_n central-difference rows
 * (B = 7 * stencil_n; the layout LocalMap::get_gradient's numerical branch evaluates, include/neural_net/local_map.cpp:110-150,
 * and neural_mapping.cpp:436-451).  Same features bit for bit; the 7 rows of a group are gathered together (register reuse
 * where they share a grid cell, cache reuse where their cells are neighbours).  jac_rows = 0 or stencil_n. */
int gsdf_hashgrid_fwd_stencil(int64_t B, int64_t stencil_n, int64_t jac_rows, int n_levels, int n_feat, int log2_hashmap,
                              int base_res, float per_level_scale, const float *x, const float *table, float *feat,
                              float *jac, gsdf_stream_t stream);
/*  gsdf_hashgrid_fwd_stencil_points: gsdf_sdf_query_points2 (stencil rows) and gsdf_hashgrid_fwd_stencil in ONE launch — the joint iteration's SDF
 *     batch starts from world points (n_a rows of xyz_a, then rows ids_b[j] — or j — of xyz_b), and the 7 encoder rows of a point are made inside the
 *     encoder's kernel with the query-points kernel's own operations (the same bits) instead of being written by one launch and read back by the next
 *     (18 of a lane's 21 position loads go, the kernel drops from 110 to 83 registers).  x_out [7 n, 3] receives exactly what gsdf_sdf_query_points2
 *     writes (the backward's table scatter reads its base rows); feat [7 n, L F], jac [n, L F, 3] (want_jac != 0) as gsdf_hashgrid_fwd_stencil. */
int gsdf_hashgrid_fwd_stencil_points(int64_t n_a, const float *xyz_a, int64_t n_b, const float *xyz_b, const int64_t *ids_b, float delta,
                                     const float *origin_host, float map_size_inv, int want_jac, int n_levels, int n_feat, int log2_hashmap,
                                     int base_res, float per_level_scale, const float *table, float *x_out, float *feat, float *jac,
                                     gsdf_stream_t stream);
/* Launch hint for gsdf_hashgrid_fwd_stencil on the CALLING THREAD (round 5): wgs_per_cu > 0 launches a RESIDENT grid of that many
 * 256-thread workgroups per CU which walks the batch, instead of one workgroup per chunk; 0 = the full grid; -1 = back to the default
 * (environment GSDF_HASHGRID_RESIDENT, else 0).  The gathers are bound by the L1's miss queue, which two waves per SIMD keep nearly as
 * full as four: a caller that runs OTHER kernels on a second stream at the same time (gsdf_extras::JointIteration's splat leg) asks for 2,
 * so that half of every SIMD's wave slots and registers stay free for them and the two legs really share the CUs instead of taking turns
 * (measured at the headline workload: the launch alone 1.67 -> 2.2 ms, the two-stream step 4.91 -> 4.72 ms).  A caller with the chip to
 * itself leaves the default.  Results are bit-identical either way.  Returns the previous value. */
int gsdf_hashgrid_fwd_stencil_resident(int wgs_per_cu);
int gsdf_hashgrid_bwd_jac(int64_t B, int n_levels, int n_feat, const float *jac, const float *v_feat, float *v_x,
                          gsdf_stream_t stream);
/* out[ids[b]] += scale * (J_b^T v_feat_b)  (ids NULL: row b): the same contraction with the chain-rule scale of the world -> unit-cube map
 * and the scatter to the selected sample rows in one launch; the rows named by ids must be distinct. */
int gsdf_hashgrid_bwd_jac_scatter(int64_t B, int n_levels, int n_feat, const float *jac, const float *v_feat, float scale, const int64_t *ids,
                                  float *out, gsdf_stream_t stream);
/* v_table ACCUMULATES (zero it first), v_x is overwritten; either may be NULL. */
int gsdf_hashgrid_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                      const float *x, const float *table, const float *v_feat, float *v_table, float *v_x,
                      gsdf_stream_t stream);
/* The table gradient alone, without global atomics (count -> plan -> emit 12-byte records bucketed by 64 KB table tile ->
 * accumulate each tile in LDS): the scatter of the joint iteration's large batches (7 x (ray points + visible splat
 * samples), neural_mapping.cpp:106-136,448-451).  v_table ACCUMULATES and must not be written by anything else while
 * the call runs (tiles are added with plain read-modify-writes).  ws: gsdf_hashgrid_bwd_binned_ws_bytes(...) bytes,
 * 256-byte aligned; that function returns 0 for grids whose levels have more than 128 tiles of 4096 entries
 * (log2_hashmap_size > 19), for which only gsdf_hashgrid_bwd exists.  Each tile's sum is accumulated in 64-bit fixed
 * point (exact to 2^-41 of the level's largest |v_feat| per contribution, independent of the order of the
 * contributions), then rounded to fp32 once: results agree with gsdf_hashgrid_bwd's to fp32 summation error and are
 * bit-reproducible from run to run unless a tile receives more than 2^19 contributions. */
size_t gsdf_hashgrid_bwd_binned_ws_bytes(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                                         float per_level_scale);
int gsdf_hashgrid_bwd_binned(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                             const float *x, const float *v_feat, float *v_table, void *ws, size_t ws_bytes,
                             gsdf_stream_t stream);
/* The same for a batch with the stencil structure gsdf_sdf_query_points writes (rows [0, stencil_n) base points, then six blocks
 * of stencil_n central-difference points; B == 7 * stencil_n): at the levels [0, merge_levels) the rows of a group that fall
 * into the base point's grid cell touch the same 8 entries and are summed in registers before they become records (-23 % records
 * at the reference configuration with delta = 0.02 m in a 16 m map).  Same result up to the fp32 sum of <= 7 products. */
int gsdf_hashgrid_bwd_binned_stencil(int64_t B, int64_t stencil_n, int merge_levels, int n_levels, int n_feat, int log2_hashmap,
                                     int base_res, float per_level_scale, const float *x, const float *v_feat, float *v_table,
                                     void *ws, size_t ws_bytes, gsdf_stream_t stream);
/* The binned scatter with the SECOND-ORDER term of the analytic eikonal regulariser folded in (the reference's default
 * configuration, numerical_grad: 0): v_table += (d feat / d table)^T v_feat  [v_feat may be NULL]
 *                                             + d/d table [ (J(x, table)^T v_feat2) . vv_x ]
 * i.e. gsdf_hashgrid_bwd's and gsdf_hashgrid_bwd_bwd's table gradients of the same points in ONE pass without global atomics
 * (every corner's record carries w * v_feat + (d w / d x . vv_x) * v_feat2).  v_feat2 [B, L*F], vv_x [B,3]. */
int gsdf_hashgrid_bwd_binned2(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                              const float *x, const float *v_feat, const float *v_feat2, const float *vv_x, float *v_table,
                              void *ws, size_t ws_bytes, gsdf_stream_t stream);
/* Double backward of v_x = J(x,table)^T v_feat: given vv_x [B,3] returns d/d v_feat (g_vfeat, overwritten),
 * d/d table (g_table, ACCUMULATES) and d/d x (g_x, overwritten); any may be NULL. */
int gsdf_hashgrid_bwd_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                          float per_level_scale, const float *x, const float *table, const float *v_feat,
                          const float *vv_x, float *g_vfeat, float *g_table, float *g_x, gsdf_stream_t stream);
/* Third order: the backward of gsdf_hashgrid_bwd_bwd.  With (g_vfeat, g_table, g_x) its outputs for (v_feat, vv_x), given lam_x [B,3] (the gradient
 * arriving at g_x) and mu_vfeat [B, L*F] (arriving at g_vfeat; NULL = zero) returns d/d v_feat (t_vfeat, overwritten), d/d table (t_table,
 * ACCUMULATES), d/d vv_x (t_vv, overwritten) and d/d x (t_x, overwritten); any may be NULL.  What a loss on the analytic Hessian needs:
 * LocalMap::get_gradient(hessian = true, numerical_grad = 0) + curvate_loss, include/neural_net/local_map.cpp:151-168,
 * include/neural_mapping/neural_mapping.cpp:117-121 (curvate_weight > 0). */
int gsdf_hashgrid_bwd_bwd_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                              const float *x, const float *table, const float *v_feat, const float *vv_x, const float *lam_x,
                              const float *mu_vfeat, float *t_vfeat, float *t_table, float *t_vv, float *t_x, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * S2  TCNNNetwork::forward — fully fused width-64 ReLU MLP (fp32 MFMA)
 *     reference: include/neural_net/local_map.cpp:44-55 (tcnn FullyFusedMLP, bias free), :94 (forward);
 *     same kernel serves the default torch decoder topology with biases (local_map.cpp:29-42).
 * dims_host[n_layers+1] (HOST ints): input 32|64, hidden 64, output <= 32.  weights: torch Linear layout,
 * row-major [out][in] per layer, layers concatenated; biases concatenated or NULL.
 * acts (gsdf_mlp_acts_floats(B, n_layers) floats; post-ReLU hidden activations in an OPAQUE tile layout that only
 * gsdf_mlp_bwd* read) is written when non-NULL and required by bwd.
 * ---------------------------------------------------------------------------------------- */
int gsdf_mlp_fwd(int64_t B, int n_layers, const int *dims_host, const float *weights, const float *biases,
                 const float *in, float *out, float *acts, gsdf_stream_t stream);
size_t gsdf_mlp_acts_floats(int64_t B, int n_layers);
size_t gsdf_mlp_bwd_ws_bytes(int64_t B, int n_layers);
/* Workspace gsdf_mlp_bwd needs for THIS call.  When both gradients are requested and the topology takes the one-pass backward
 * (input width 32, 4 or 5 layers, <= 16 outputs: v_pre never leaves the registers): the waves' partial weight gradients (plain stores,
 * summed by two small kernels; with ws == NULL the same call still works and each wave leaves through ~14.5 K atomics instead: 0.3 ms
 * slower per launch at 0.5 M points).  Else the per-layer gradient images that travel to gsdf_mlp_bwd_weights.
 * gsdf_mlp_bwd_ws_bytes(B, n_layers) is an upper bound of both for callers that do not have the widths at hand. */
size_t gsdf_mlp_bwd_ws_bytes_for(int64_t B, int n_layers, const int *dims_host, int want_weights);
/* 1 when gsdf_mlp_bwd with both gradients takes the one-pass kernel for this topology (gsdf_mlp_bwd_weights is then never needed), else 0 */
int gsdf_mlp_bwd_is_one_pass(int n_layers, const int *dims_host);
/* v_in [B,dims[0]] overwritten (may be NULL); v_weights / v_biases ACCUMULATE (may be NULL). */
int gsdf_mlp_bwd(int64_t B, int n_layers, const int *dims_host, const float *weights, const float *biases,
                 const float *in, const float *acts, const float *v_out, float *v_in, float *v_weights,
                 float *v_biases, void *ws, gsdf_stream_t stream);
/* Double backward of the decoder: the analytic eikonal term of the reference's default configuration differentiates
 * v_in = d sdf / d features (torch::autograd::grad(..., create_graph=true), include/neural_net/local_map.cpp:151-172) again.
 * Inputs: `acts` of gsdf_mlp_fwd, and of the FIRST backward gsdf_mlp_bwd(..., v_weights = NULL, ws = bwd_ws) its upstream
 * v_out and its workspace bwd_ws (kept by the caller); vv_in [B, dims[0]] = dL/d v_in.
 * Outputs: g_vout [B, dims[n]] = dL/d v_out (required; overwritten) and g_weights (same layout as weights; ACCUMULATES; may be
 * NULL).  Nothing flows to the biases or to the network input (a ReLU network is piecewise linear).
 * bwd_ws = NULL (round 4; topologies gsdf_mlp_bwd_is_one_pass covers): nothing of the first backward is needed — the chain v_out -> v_pre_l
 * is recomputed in registers from the ReLU masks inside the pass that accumulates g_weights, and the matching first backward may be the
 * lean one (gsdf_mlp_bwd with v_weights = NULL and ws = NULL: input gradient only, nothing saved).
 * ws: gsdf_mlp_bwd_bwd_ws_bytes(B, n_layers) bytes. */
size_t gsdf_mlp_bwd_bwd_ws_bytes(int64_t B, int n_layers);
int gsdf_mlp_bwd_bwd(int64_t B, int n_layers, const int *dims_host, const float *weights, const float *acts,
                     const float *v_out, const void *bwd_ws, const float *vv_in, float *g_vout, float *g_weights, void *ws,
                     gsdf_stream_t stream);
/* The weight-gradient half alone, for callers that run it on another stream: `ws` must hold the result of a preceding
 * gsdf_mlp_bwd(..., v_weights = NULL, ..., ws) on the same arguments.  v_weights / v_biases ACCUMULATE. */
int gsdf_mlp_bwd_weights(int64_t B, int n_layers, const int *dims_host, int has_biases, const float *in,
                         const float *acts, const float *v_out, const void *ws, float *v_weights, float *v_biases,
                         gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1  distCUDA2(points) (simple-knn)          reference call: include/neural_gaussian/neural_gaussian.cpp:314
 *     out[i] = mean of the squared distances from point i to its 3 nearest neighbours (exact).
 * ---------------------------------------------------------------------------------------- */
size_t gsdf_knn_ws_bytes(int64_t n_points);
int gsdf_knn_mean_dist2(int64_t n_points, const float *points, float *out, void *ws, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * O1  fused Adam step over a flat parameter buffer (torch.optim.Adam semantics, no amsgrad / weight decay):
 *     replaces the unfused libtorch Adam of include/neural_mapping/neural_mapping.cpp:466-469 (groups and learning
 *     rates: include/neural_gaussian/neural_gaussian.cpp:434-453).  Segments = parameter groups laid out back to
 *     back in the flat buffer: seg_begin_host[k] (element offset, sorted, [0] == 0) and seg_lr_host[k] (HOST arrays).
 *     `step` counts from 1 (bias correction).  params / exp_avg / exp_avg_sq are updated in place.
 * ---------------------------------------------------------------------------------------- */
int gsdf_adam_step(int64_t n, int n_segments, const int64_t *seg_begin_host, const float *seg_lr_host, float *params,
                   const float *grads, float *exp_avg, float *exp_avg_sq, float beta1, float beta2, float eps,
                   int64_t step, gsdf_stream_t stream);
/* the same, and the gradient buffer is zeroed as it is consumed (optimizer.zero_grad() of the next iteration, neural_mapping.cpp:466, without
 * a pass of its own) */
int gsdf_adam_step_zero_grad(int64_t n, int n_segments, const int64_t *seg_begin_host, const float *seg_lr_host, float *params, float *grads,
                             float *exp_avg, float *exp_avg_sq, float beta1, float beta2, float eps, int64_t step, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * S3  the per-ray SDF batch's elementwise ends (replace ~100 eager libtorch kernels per step):
 *   gsdf_sdf_query_points: world points -> unit-cube encoder inputs, out = 0.5*(((x - origin)*2)*map_size_inv) + 0.5
 *     (SubMap::xyz_to_zp1_pts, include/neural_net/sub_map.cpp:82-97); with stencil != 0 the output holds 7n rows: the n
 *     base points, then the 6 central-difference points x +/- delta e_axis of LocalMap::get_gradient in its order
 *     (+x,-x,+y,-y,+z,-z; include/neural_net/local_map.cpp:110-124), each block n rows.  origin_host: 3 HOST floats.
 *   gsdf_sdf_ray_loss: attr [(stencil?7:1)*n, ld] = decoder output (col 0 sdf, col 1 raw isigma) on those rows.
 *     loss[0] = mean_n BCEWithLogits(-sdf*is, clamp(sigmoid(-gt*is),1e-7,1-1e-7)),  is = min(1+softplus_100(raw)*bce_isigma, 5e2)
 *               (loss::sdf_loss, include/optimizer/loss.cpp:49-79; isigma: local_map.cpp:87-103)
 *             + w_eik * mean_n (|g|-1)^2,  g = (sdf(+d) - sdf(-d)) / (2 delta)   (loss::eikonal_loss, loss.cpp:81-83)
 *     v_attr [same shape] = d loss[0] / d attr (every column written).
 * ---------------------------------------------------------------------------------------- */
int gsdf_sdf_query_points(int64_t n, int stencil, const float *xyz, float delta, const float *origin_host,
                          float map_size_inv, float *out, gsdf_stream_t stream);
/* The same for the batch [xyz_a (n_a rows); xyz_b[ids_b] (n_b rows; ids_b NULL: the first n_b rows)]: gather + concatenation + query_points
 * of the joint iteration's SDF batch (per-ray points, then the visible splats' samples) in one launch. */
int gsdf_sdf_query_points2(int64_t n_a, const float *xyz_a, int64_t n_b, const float *xyz_b, const int64_t *ids_b, int stencil, float delta,
                           const float *origin_host, float map_size_inv, float *out, gsdf_stream_t stream);
 /*  gsdf_gs_sdf_loss: loss[0] = scale * 0.5 * sum_i w_i * attr[i][0]^2 (loss::gs_sdf_loss, loss.cpp:7-11), w_i =
 *     weights[ids[i]] (the row selection of neural_mapping.cpp:436-437; ids NULL: weights[i]); v_attr = d loss / d attr. */
int gsdf_gs_sdf_loss(int64_t n, const float *attr, int ld, const float *weights, const int64_t *ids, float scale,
                     float *loss, float *v_attr, gsdf_stream_t stream);
/*  gsdf_gs_sdf_eik_loss: the same with stencil != 0 on attr [7n, ld] (rows as gsdf_sdf_query_points writes them): adds the
 *     eikonal regulariser of the visible splats' samples, w_eik * mean_n (|g|-1)^2 with the central-difference gradient g
 *     (NeuralSLAM::sdf_regularization(gs_samples.detach(), ...), neural_mapping.cpp:448-451 -> :106-116). */
int gsdf_gs_sdf_eik_loss(int64_t n, int stencil, const float *attr, int ld, const float *weights, const int64_t *ids,
                         float scale, float delta, float w_eik, float *loss, float *v_attr, gsdf_stream_t stream);
/*  gsdf_sdf_analytic_loss: the reference's DEFAULT SDF regulariser (numerical_grad: 0, config/base.yaml:13): eikonal on the
 *     ANALYTIC gradient g = map_size_inv * J^T g0 (LocalMap::get_gradient's autograd branch, local_map.cpp:151-172; g0 [n,32] =
 *     d sdf / d features from gsdf_mlp_bwd with v_out = (1, 0), jac [n,32,3] from gsdf_hashgrid_fwd_jac_rows) plus the align term
 *     w_align * mean |g - g_num.detach()| against the central differences of the stencil rows (neural_mapping.cpp:126-134),
 *     on top of the data terms of ONE batch that holds both of the iteration's point sets: rows [0, n_ray) = the per-ray batch,
 *     w_sdf * loss::sdf_loss against gt_sdf [n_ray]; rows [n_ray, n) = the visible splats' samples, w_gs * loss::gs_sdf_loss with
 *     weights[ids[i - n_ray]] (ids NULL: weights[i - n_ray]).  The regularisers are means over each set separately, as the two
 *     sdf_regularization calls of the iteration are (neural_mapping.cpp:183-186, 448-451).  attr [(stencil ? 7 : 1) n, ld].
 *     Outputs: loss[0]; v_attr [n, ld] (may be NULL) = d loss / d attr of the BASE rows (the stencil rows are detached); vv_x [n,3] =
 *     d loss / d (J^T g0); u0 [n,32] = J vv_x = d loss / d g0. */
int gsdf_sdf_analytic_loss(int64_t n, int64_t n_ray, int stencil, const float *attr, int ld, const float *g0, int n_feat,
                           const float *jac, const float *gt_sdf, const float *weights, const int64_t *ids, float bce_isigma,
                           float w_sdf, float w_gs, float map_size_inv, float delta, float w_eik, float w_align, float *loss,
                           float *v_attr, float *vv_x, float *u0, gsdf_stream_t stream);
/*  gsdf_sdf_data_term_grad: v_attr [n, ld] of gsdf_sdf_analytic_loss ALONE (the same bits) — d (data terms) / d attr of the base rows needs the
 *     decoder's output at the base rows and nothing else (neither the stencil rows, nor g0, nor the Jacobian).  A caller on whose critical path the
 *     samples' gradient lies (the joint iteration: decoder backward -> Jacobian contraction -> the splats' optimizer -> the next render) launches
 *     this first and passes v_attr = NULL to gsdf_sdf_analytic_loss later. */
int gsdf_sdf_data_term_grad(int64_t n, int64_t n_ray, const float *attr, int ld, const float *gt_sdf, const float *weights, const int64_t *ids,
                            float bce_isigma, float w_sdf, float w_gs, float *v_attr, gsdf_stream_t stream);
int gsdf_sdf_ray_loss(int64_t n, int stencil, const float *attr, int ld, const float *gt_sdf, float bce_isigma,
                      float delta, float w_eik, float *loss, float *v_attr, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A1  occupancy acceleration structure: replaces the OctreeAS of the absent kaolin_wisp_cpp submodule as the reference's
 *     call sites use it (SubMap::update_octree_as include/neural_net/sub_map.cpp:22-35; get_valid_mask :76-80;
 *     LocalMap::sample / filter_sample include/neural_net/local_map.cpp:449-516; get_quantized_points
 *     include/neural_mapping/neural_mapping.cpp:755-758).  Coordinates are kaolin's: the map cube is [-1,1]^3,
 *     q = clamp(floor(2^level (x+1)/2), 0, 2^level-1).  `grid` = caller-owned device buffer of gsdf_occ_bytes(level)
 *     bytes: a bit pyramid (level l: 2^(3l) bits, x fastest, uint32 words; levels 0..level back to back; parent = OR of
 *     its 8 children).  level in [1,12].  Semantics: DESIGN.md SPEC A.9.
 *   build:     clears the pyramid, sets the level-`level` voxel of every point (dilate27 != 0: and its 26 neighbours,
 *              clamped to the cube = points_to_neighbors(...).clamp(0,res-1)), then the coarser levels.
 *   query:     mask[i] = 1 iff xyz_i is inside [-1,1]^3 and its cell at query_level (<0: level) is occupied
 *              (= query(xyz, level).pidx > -1).
 *   voxel_counts / voxel_list: two-phase get_quantized_points(): word_counts[w] = popcount of the w-th word of the
 *              finest level (2^(3 level)/32 words); the caller scans them (exclusive, int64) and passes word_offsets;
 *              voxels[V,3] int16 (x,y,z) in x-fastest order.
 *   raymarch:  two-phase "voxel" ray march.  count: counts[r] = number of occupied finest-level voxels ray r
 *              (origin + t*dir, t >= 0) crosses, front to back.  fill: voxel_offsets = exclusive scan of counts (int64);
 *              for the v-th voxel of ray r and k < num_samples, row j = (voxel_offsets[r]+v)*num_samples + k:
 *              ridx[j] = r, depth_samples[j] = t_in + (t_out-t_in)(k+1/2)/num_samples, samples[j] = origin + t*dir.
 *              One wave per ray, one slab of the ray's major axis per lane (<= 3 cells per slab): csrc/occupancy.hip.
 * ---------------------------------------------------------------------------------------- */
size_t gsdf_occ_bytes(int level);
int gsdf_occ_build(int level, int64_t n_points, const float *xyz_m1p1, int dilate27, void *grid, gsdf_stream_t stream);
int gsdf_occ_query(int level, int query_level, int64_t n, const float *xyz_m1p1, const void *grid, uint8_t *mask,
                   gsdf_stream_t stream);
/* query with the SubMap::xyz_to_m1p1_pts transform ((x - origin) * 2) * map_size_inv folded in (origin_host: 3 HOST floats) */
int gsdf_occ_query_world(int level, int query_level, int64_t n, const float *xyz_world, const float *origin_host,
                         float map_size_inv, const void *grid, uint8_t *mask, gsdf_stream_t stream);
/* The visible, occupancy-valid splat samples of the joint iteration (neural_mapping.cpp:423-437: samples_weights * visibilities,
 * get_valid_mask(samples) & (visibilities > k_visible_thr), nonzero) in three launches: w_all[i] = samples_weights[i] * visibilities[i]
 * for every row; ids[0 .. *count) = the rows with visibilities > vis_thresh whose sample lies in an occupied voxel, in increasing order;
 * *count (device int64) = their number.  ids has room for n entries.  ws: gsdf_visible_set_ws_bytes(n). */
size_t gsdf_visible_set_ws_bytes(int64_t n);
int gsdf_visible_set(int level, int query_level, int64_t n, const float *xyz_world, const float *origin_host, float map_size_inv,
                     const void *grid, const float *visibilities, const float *samples_weights, float vis_thresh, float *w_all,
                     int64_t *ids, int64_t *count, void *ws, gsdf_stream_t stream);
int gsdf_occ_voxel_counts(int level, const void *grid, int32_t *word_counts, gsdf_stream_t stream);
int gsdf_occ_voxel_list(int level, const void *grid, const int64_t *word_offsets, int16_t *voxels, gsdf_stream_t stream);
int gsdf_occ_raymarch_count(int level, int64_t n_rays, const float *origins_m1p1, const float *dirs, const void *grid,
                            int32_t *counts, gsdf_stream_t stream);
int gsdf_occ_raymarch_fill(int level, int64_t n_rays, const float *origins_m1p1, const float *dirs, const void *grid,
                           const int64_t *voxel_offsets, int num_samples, int32_t *ridx, float *samples,
                           float *depth_samples, gsdf_stream_t stream);

/* The reference's per-ray SDF batch in two passes over the rays instead of ~60 libtorch launches:
 *   NeuralSLAM::sample          include/neural_mapping/neural_mapping.cpp:73-104
 *   LocalMap::sample            include/neural_net/local_map.cpp:449-509   (1 sample per occupied voxel crossed + free samples, ray_sdf > 0 kept)
 *   utils::sample_surface_pts / sample_free_pts   include/utils/utils.cpp:336-393
 *   SubMap::get_inrange_mask    include/neural_net/sub_map.cpp:37-45
 * Rays: origin / direction [n,3], depth [n,1], end_xyz [n,3] in the WORLD frame.  The random numbers stay the caller's (torch's generator,
 * drawn in the reference's order): rand_free [n, free_sample_num] U[0,1), randn_surf [n, surface_sample_num] N(0,1).  map_origin (3 host
 * floats), map_size_inv, map_half = 1 / map_size_inv define SubMap's frames: m1p1 = ((x - origin) * 2) * map_size_inv,
 * world = (m1p1 * 0.5) * map_half + origin.  range_lo / range_hi: get_inrange_mask keeps lo < xyz < hi (its own padded bounds).
 * Output rows, in the reference's order: [voxel samples: rays in order, front to back | free samples, ray-major | surface samples,
 * ray-major | ray end points], the first two segments filtered by ray_sdf > 0, targets of the first three truncated to +-truncated_dis,
 * everything filtered by the in-range test.  Every elementwise operation is the reference's, in its order, in fp32.
 *   count: counts [4][n] int32 (kept rows per segment and ray), offsets_incl [4 n] int64 = inclusive scan of counts, *total = rows
 *          (total may be a host-visible word, gsdf_host_words_alloc).  Two launches.
 *   fill:  writes the rows (xyz [B,3], ray_sdf [B,1], ridx [B] int64, origin / direction [B,3], depth [B,1]).  One launch. */
typedef struct {
  int level;
  int64_t n_rays;
  const float *origin, *direction, *depth, *end_xyz;
  const void *grid;
  const float *rand_free, *randn_surf;
  int free_sample_num, surface_sample_num;
  float map_origin[3], map_size_inv, map_half;
  float range_lo[3], range_hi[3];
  float sample_std, truncated_dis;
} gsdf_ray_sampler_args;
int gsdf_ray_sampler_count(const gsdf_ray_sampler_args *args, int32_t *counts, int64_t *offsets_incl, int64_t *total, gsdf_stream_t stream);
int gsdf_ray_sampler_fill(const gsdf_ray_sampler_args *args, const int32_t *counts, const int64_t *offsets_incl, float *xyz, float *ray_sdf,
                          int64_t *ridx, float *origin, float *direction, float *depth, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * M1  marching cubes on a dense scalar grid [res_x][res_y][res_z] (x slowest): replaces mc::marching_cubes of the
 *     reference's in-tree CUDA mesher (include/mesher/cumcubes/src/cumcubes_kernel.cu:7-282, include/cumcubes.hpp:10-19).
 *     inside <=> value > thresh; one vertex per grid edge whose end points straddle thresh, at index + (thresh-d0)/(d1-d0)
 *     along the edge, mapped to world as v * (upper-lower)/res + lower; faces index the vertices (int32, 0-based).
 *     Two-phase: count -> n_vert[c] (0..3 edges owned by cell c) and n_tri[c] (0..5) per cell, c = (x*res_y + y)*res_z + z;
 *     the caller scans both (exclusive, int64) and allocates; emit writes vertices [V,3] (by cell, then axis x,y,z) and
 *     faces [F,3] (by cell, then table order).  table: GSDF_MC_TABLE_REFERENCE = the reference's in-tree triangle table
 *     (include/mesher/cumcubes/include/utils.cuh:31-289): per cell the same triangles as the reference's mesh;
 *     GSDF_MC_TABLE_WATERTIGHT = the derived table (tools/gen_mc_table.py): same vertices, triangulation that closes the
 *     classic table's ambiguous-face cracks.
 * ---------------------------------------------------------------------------------------- */
#define GSDF_MC_TABLE_REFERENCE 0
#define GSDF_MC_TABLE_WATERTIGHT 1
int gsdf_mc_count(int res_x, int res_y, int res_z, int table, const float *grid, float thresh, int32_t *n_vert,
                  int32_t *n_tri, gsdf_stream_t stream);
int gsdf_mc_emit(int res_x, int res_y, int res_z, int table, const float *grid, float thresh, const int64_t *v_offsets,
                 const int64_t *t_offsets, const float *lower_host, const float *upper_host, float *vertices,
                 int32_t *faces, gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a2  splat parameter activations (NeuralGS::generate_gaussian / get_xyz / get_scale / get_opacity,
 *     include/neural_gaussian/neural_gaussian.cpp:463-492): xyz = anchors + offsets, scales = exp(log_scales),
 *     opacities = sigmoid(logit_opacities).  bwd ACCUMULATES into g_offsets / g_log_scales / g_logit_opacities (the
 *     parameter-gradient buffers); any of the three upstream gradients may be NULL (treated as zero).
 * ---------------------------------------------------------------------------------------- */
int gsdf_splat_activations_fwd(int64_t n, const float *anchors, const float *offsets, const float *log_scales,
                               const float *logit_opacities, float *xyz, float *scales, float *opacities,
                               gsdf_stream_t stream);
int gsdf_splat_activations_bwd(int64_t n, const float *scales, const float *opacities, const float *v_xyz,
                               const float *v_scales, const float *v_opacities, float *g_offsets, float *g_log_scales,
                               float *g_logit_opacities, gsdf_stream_t stream);

/* Row gathers / scatters of the visible set (ABI 8) — what the caller writes as xyz.index_select(0, gaussian_ids), opacities.index_select,
 * torch::ones and index_add_ (include/neural_gaussian/neural_gaussian.cpp:259-262 centre samples; the samples' gradient back into the offsets):
 *   visible_gather: xyz_rows[m] = xyz[ids[m]], opacity_rows[m] = opacities[ids[m]], ones[m] = 1 (any output may be NULL);
 *   rows_scatter_add: dst[ids[m], :] += src[m, :], cols = 1 or 3; ids_unique != 0 (one camera: every splat appears once) -> plain
 *   read-modify-write, else atomics. */
int gsdf_visible_gather(int64_t n_visible, const int64_t *gaussian_ids, const float *xyz, const float *opacities, float *xyz_rows,
                        float *opacity_rows, float *ones, gsdf_stream_t stream);
int gsdf_rows_scatter_add(int64_t n_rows, int cols, const int64_t *ids, int ids_unique, const float *src, float *dst, gsdf_stream_t stream);

/* isotropic regulariser of the visible splats (include/neural_mapping/neural_mapping.cpp:268-276):
 *     loss[0] = mean over [M,2] of |scale - mean(scale, -1)| with scale = scales[gaussian_ids][:, 0:2]  (= sum |s_u - s_v| / 2M);
 *     bwd ACCUMULATES v_loss[0] * d loss / d scales into v_scales [N,3]. */
int gsdf_isotropic_loss_fwd(int64_t n_visible, const float *scales, const int64_t *gaussian_ids, float *loss, gsdf_stream_t stream);
int gsdf_isotropic_loss_bwd(int64_t n_visible, const float *scales, const int64_t *gaussian_ids, const float *v_loss, float *v_scales,
                            gsdf_stream_t stream);
/* value and gradient in ONE launch (the joint iteration's step: the value is a number only the log reads — no launch and no memset of its
 * own): `loss` [1] ACCUMULATES (the caller zeroes it, e.g. together with the buffer it lives in), v_scales as in _bwd. */
int gsdf_isotropic_loss_fwd_bwd(int64_t n_visible, const float *scales, const int64_t *gaussian_ids, const float *v_loss, float *loss,
                                float *v_scales, gsdf_stream_t stream);
/* NeuralGS::prune_nan_gs's per-iteration test (include/neural_gaussian/neural_gaussian.cpp:907-916): count[0] = number of splats
 * with a NaN in offsets [n,3] / scaling [n,3] / quaternion [n,4]; mask u8 [n] (optional) marks them. */
int gsdf_nan_rows(int64_t n, const float *offsets, const float *scaling, const float *quaternion, int32_t *count, uint8_t *mask,
                  gsdf_stream_t stream);
/* the same test ADDED to a running total (no memset, no separate add: one launch per iteration in the joint step) */
int gsdf_nan_rows_accumulate(int64_t n, const float *offsets, const float *scaling, const float *quaternion, int32_t *total,
                             gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a8  NeuralGS::update_state (include/neural_gaussian/neural_gaussian.cpp:626-680) in one launch: the densification
 *     statistics after every backward pass.  grad [M,2] = .grad() of the `densify` (or `absgrad`) leaf, scaled by
 *     (W/2*C, H/2*C) and L2-normed (:660-665); grad2d/count [N] accumulate; vis/radii [N] take the maximum
 *     (radii_px int32 [M] / max(W,H); radii and radii_px may both be NULL = criterion off, :675-679).
 *     n_cameras == 1: rows have unique ids (packed projection) and are updated with plain read-modify-writes.
 * ---------------------------------------------------------------------------------------- */
int gsdf_densify_stats(int64_t M, int64_t N, int n_cameras, int width, int height, const float *grad,
                       const int64_t *gaussian_ids, const float *visibilities, const int32_t *radii_px, float *grad2d,
                       float *count, float *vis, float *radii, gsdf_stream_t stream);
/* a18 row surgery of the refinement steps on a FIELD-MAJOR flat buffer [field 0: n x w0 | field 1: n x w1 | ...] (the splat
 *     parameters and both Adam moments): dst rows 0..n_keep-1 = src rows keep_idx[r] (NULL: the first n_keep rows), all fields
 *     in one launch; dst has n_dst >= n_keep rows per field (the caller fills the appended rows).  Replaces the index_select
 *     half of include/optimizer/optimizer_utils/optimizer_utils.cpp:5-165.  widths_host: n_fields HOST ints. */
int gsdf_flat_rows_gather(int n_fields, const int32_t *widths_host, int64_t n_src, int64_t n_dst, int64_t n_keep,
                          const int64_t *keep_idx, const float *src, float *dst, gsdf_stream_t stream);

/* a18 one refinement step of the splat set on the field-major flat buffers [offsets n x 3 | scaling n x 3 | quaternion n x 4 | opacity n |
 *     features_dc n x 3 | features_rest n x n_rest_cols], in two passes over the rows: NeuralGS::grow_gs (duplicate, split), prune_gs and the
 *     Adam-state surgery under them (include/neural_gaussian/neural_gaussian.cpp:690-890; include/optimizer/optimizer_utils/
 *     optimizer_utils.cpp:5-165) composed into one row map.  The new set is, in the reference's order: A the old rows neither split nor
 *     pruned (moments kept) | B the unpruned copies of the duplicated rows | C, D the unpruned first / second children of the split rows
 *     (B, C, D: zero moments).  Decisions (mode 0): grads = grad2d / max(count, 1) > grow_grad2d; small = max(exp(scaling[:2])) <=
 *     grow_scale3d (the caller passes k_grow_scale3d * spatial_scale); duplicate = high & small; split = high & !small (| radii >
 *     grow_scale2d when use_radii); prune = sigmoid(opacity) < prune_opa | min(exp(scaling[:2])) < prune_scale_min (| max > prune_scale3d when
 *     use_prune_scale3d), tested on the rows AFTER growing (children: scale / 1.6).  mode 1: prune the rows with mask[i] != 0 only
 *     (prune_invisible_gs :892-905, prune_nan_gs :907-916).
 *   plan:  counts [4][n] int32 (A, B, C flags and the split flag), offsets_incl [4 n] int64 (their inclusive scan, segment-major),
 *          totals [4] int64 = { nA, nB, nC, n_split } (device memory or host-visible words, gsdf_host_words_alloc).  4 launches.
 *   apply: totals_host = the four totals as HOST values; randn [2, n_split, 3] N(0,1) (the reference's torch::randn({2, n, 3}), :781);
 *          new buffers of n_new = nA + nB + 2 nC rows: flat_new, m_new / v_new (both or neither), anchors_new [n_new,3], state_new[4] =
 *          grad2d, count, vis, radii images (each may be NULL; children and copies inherit the parent's values, as the reference's
 *          cat(index_select) does — the caller zeroes what zero_state() zeroes).  1 launch. */
typedef struct {
  int64_t n;
  int n_rest_cols;                         /* 3 x (SH bases - 1) */
  int mode;                                /* 0 = grow + prune, 1 = prune by mask */
  const float *flat, *adam_m, *adam_v;     /* adam_m / adam_v may be NULL (no optimizer state yet) */
  const float *anchors;                    /* [n,3] */
  const float *grad2d, *count, *vis, *radii;
  const uint8_t *mask;                     /* mode 1 */
  float grow_grad2d, grow_scale3d, grow_scale2d, prune_opa, prune_scale_min, prune_scale3d;
  int use_radii, use_prune_scale3d;
} gsdf_refine_args;
size_t gsdf_refine_ws_bytes(int64_t n);
int gsdf_refine_plan(const gsdf_refine_args *args, int32_t *counts, int64_t *offsets_incl, int64_t *totals, void *ws, gsdf_stream_t stream);
int gsdf_refine_apply(const gsdf_refine_args *args, const int32_t *counts, const int64_t *offsets_incl, const int64_t *totals_host,
                      const float *randn, float *flat_new, float *m_new, float *v_new, float *anchors_new, float *const *state_new,
                      gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hint for the XCD-aware kernels (compositing: one band of tiles per XCD; hash-grid forward: one group of levels per
 * XCD): how many XCDs the queue behind `stream` can use.  Default 8; a caller that launches on a CU-masked stream
 * (hipExtStreamCreateWithCUMask) registers the number of XCDs its mask leaves enabled; 0 forgets the stream.
 * Locality only: results do not depend on it.
 * ---------------------------------------------------------------------------------------- */
int gsdf_stream_set_xcds(gsdf_stream_t stream, int n_xcds);

/* ------------------------------------------------------------------------------------------
 * O2  fused photometric loss  L = w_l1 * mean|I-G| + w_ssim * (1 - mean SSIM(I,G))  on [H,W,3] images:
 *     loss::rgb_loss + loss::dssim_loss (include/optimizer/loss/loss.cpp:22-47) with loss_utils::ssim
 *     (include/optimizer/loss_utils/loss_utils.cpp:71-117; 11-tap window of loss_utils.cpp:6-14 passed by the host,
 *     zero padding 5, C1 = 0.01^2, C2 = 0.03^2), called at include/neural_mapping/neural_mapping.cpp:237-240.
 * fwd: sums[2] (device) = { sum |I-G|, sum SSIM } over the 3*H*W pixel-channels; maps [3,H,W,3] (device, may be
 *      NULL for evaluation) keeps dSSIM/d(mu1, E[x^2], E[xy]) for the backward.
 * bwd: v_img [H,W,3] = dL/dI given the upstream scalar gradient v_loss (device pointer).
 * ---------------------------------------------------------------------------------------- */
int gsdf_l1_dssim_fwd(int height, int width, const float *img, const float *gt, const float *window11_host,
                      float *sums, float *maps, gsdf_stream_t stream);
int gsdf_l1_dssim_bwd(int height, int width, const float *img, const float *gt, const float *window11_host,
                      const float *maps, const float *v_loss, float w_l1, float w_ssim, float *v_img,
                      gsdf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * O3  fused depth->normal + normal-consistency loss
 *     sensor::depth_to_normal (include/utils/sensor_utils/cameras.hpp:176-226) and
 *     normal_error = mean(alpha^2 - nan_to_num((n_depth * alpha) . render_normal)) (include/neural_mapping/neural_mapping.cpp:243-266).
 * intrinsics4_host = {fx, fy, cx, cy}, pose_c2w_host = row-major [3,4] camera->world (HOST arrays);
 * depth [H,W,1], alpha [H,W,1] (treated as constant, the reference detaches it), render_normal [H,W,3] (world).
 * fwd: loss[1] (device).  bwd: v_depth [H,W,1], v_render_normal [H,W,3] given the upstream scalar v_loss (device).
 * ---------------------------------------------------------------------------------------- */
int gsdf_normal_consistency_fwd(int height, int width, const float *intrinsics4_host, const float *pose_c2w_host,
                                const float *depth, const float *alpha, const float *render_normal, float *loss,
                                gsdf_stream_t stream);
int gsdf_normal_consistency_bwd(int height, int width, const float *intrinsics4_host, const float *pose_c2w_host,
                                const float *depth, const float *alpha, const float *render_normal, const float *v_loss,
                                float *v_depth, float *v_render_normal, gsdf_stream_t stream);
/* value and gradients in ONE launch: `loss` [1] ACCUMULATES (the caller zeroes it); v_depth / v_render_normal as in _bwd. */
int gsdf_normal_consistency_fwd_bwd(int height, int width, const float *intrinsics4_host, const float *pose_c2w_host,
                                    const float *depth, const float *alpha, const float *render_normal, const float *v_loss,
                                    float *loss, float *v_depth, float *v_render_normal, gsdf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSDF_HIP_H */
