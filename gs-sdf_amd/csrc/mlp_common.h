// mlp_common.h — descriptors and the "register image" layouts shared by the decoder kernels (mlp.hip: fp32 MFMA,
// mlp_split.hip: split-bf16 MFMA).
#pragma once
#include <stdlib.h>

#include "common.h"

namespace gsdf {

typedef float v16f __attribute__((ext_vector_type(16)));

static constexpr int MLP_THREADS = 256;
static constexpr int HID = 64;
static constexpr int MAX_LAYERS = 8;

struct MlpDesc {
  int n_layers;           // linear layers
  int d_in;               // 32 or 64
  int d_out;              // <= 32
  int w_off[MAX_LAYERS];  // float offset of layer l in the torch-layout weight blob
  int b_off[MAX_LAYERS];
  int lds_off[MAX_LAYERS];  // float offset of layer l in the permuted LDS image
  int has_bias;
};

// neuron held by (register r, half h) of a 32-row D tile
__device__ __forceinline__ int d_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// K permutation of a 64-wide hidden operand: k-step s in [0,32) -> neuron
__device__ __forceinline__ int perm_hidden(int s, int h) { return 32 * (s >> 4) + d_row(s & 15, h); }

// "Register image" layout of the tensors that only travel between these kernels (saved activations, v_pre):
// [slot][tile of 32 points][half t][q 0..3][lane 0..63][4 floats]: float4 q of a lane = its accumulator registers 4q..4q+3,
// i.e. neurons 32 t + d_row(4 q + j, lane >> 5) of point 32 tile + (lane & 31).  A wave stores / loads a D tile with four
// instructions of 64 lanes x 16 contiguous bytes (whole 1 KiB runs, every HBM line written by a single instruction).
// Rows are padded to whole tiles: gsdf_mlp_acts_floats().  img_off() = address of float4 0 of `lane`; float4 q at + IMG_Q * q.
static constexpr int IMG_Q = 256;   // floats between consecutive float4s of a lane
__device__ __forceinline__ int64_t img_off(int64_t slot, int64_t n_tiles, int64_t tile, int t, int lane) {
  return (((slot * n_tiles + tile) * 2 + t) * 1024 + lane * 4);
}
// ReLU masks (bit r of a uint16 = accumulator register r of that lane was > 0): stored behind the images in the same
// buffer, [slot][tile][t][lane]; the data-gradient kernel reads 2 B per lane instead of the 64 B of activations.
__device__ __forceinline__ int64_t mask_off(int64_t slot, int64_t n_tiles, int64_t tile, int t, int lane) {
  return ((slot * n_tiles + tile) * 2 + t) * 64 + lane;
}
// the lane/register that holds neuron-in-tile j (0..31) of a point: lane half = (j >> 2) & 1, r = (j & 3) + 4 (j >> 3)
__device__ __forceinline__ int img_half_of(int j) { return (j >> 2) & 1; }
__device__ __forceinline__ int img_reg_of(int j) { return (j & 3) + 4 * (j >> 3); }

__device__ __forceinline__ v16f mfma32(float a, float b, v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// mlp_split.hip: the same operators on the bf16 MFMA pipe (exact 3-term operand split, fp32 accuracy).
// Return 1 if launched, 0 if the path does not cover the call (topology, LDS, GSDF_MLP_MFMA=f32), < 0 on error.
int mlp_fwd_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *bias, const float *in, float *out, float *acts,
                         hipStream_t stream);
// does mlp_bwd_split_launch cover this topology (both gradients requested)?
bool mlp_bwd_split_covers(const MlpDesc &d);
// data and weight gradients in one pass (v_W required; v_in, v_b optional); v_W / v_b ACCUMULATE.  ws (mlp_bwd_split_ws_bytes, may be
// NULL): the waves' weight-gradient tiles leave as plain stores into per-wave partial buffers and two small kernels sum them, instead
// of one round of ~14.5 K atomics per wave on the same 58 KB (measured 0.3 us per wave: 0.3 ms of a 0.55 ms launch at 494 K points)
int mlp_bwd_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *in, const float *acts, const float *v_out,
                         float *v_in, float *v_W, float *v_b, void *ws, hipStream_t stream);
// round 4, the analytic configuration's e0 backward and double backward on the same pipe (mlp_split.hip: BWD_DATA / BWD_TANGENT, MASKED):
int mlp_bwd_data_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *acts, const float *v_out, float *v_in, hipStream_t stream);
int mlp_fwd_masked_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *vv_in, float *out, float *tangent, const float *mask_acts,
                                hipStream_t stream);
int mlp_bwd_tangent_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *vv_in, const float *tangent, const float *mask_acts,
                                 const float *v_out, float *g_W, void *ws, hipStream_t stream);
size_t mlp_bwd_split_ws_bytes(int64_t B, const MlpDesc &d);
size_t mlp_bwd_split_ws_bytes_bound(int64_t B, int n_layers);   // for callers that do not know the widths

}  // namespace gsdf
