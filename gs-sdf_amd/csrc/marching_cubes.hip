// marching_cubes.hip — M1: iso-surface extraction of a dense scalar grid (mesh export of the SDF).
// Replaces mc::marching_cubes of the reference's in-tree CUDA mesher
// (/root/reference/include/mesher/cumcubes/src/cumcubes_kernel.cu:7-282; caller include/neural_net/local_map.cpp:258-300):
// same vertex definition (one vertex per grid edge whose end points straddle `thresh`, owned by the edge's lower cell,
// position = index + (thresh - d0)/(d1 - d0) on the edge's axis, then * (upper - lower)/res + lower, :97-139,260-274),
// same corner / edge numbering (:169-193), inside <=> value > thresh.
// MI355X design differences: (1) no global atomic counters — per-cell counts, an exclusive scan by the caller, then a
// deterministic emit (vertices ordered by cell then axis, faces by cell then table order; the reference's order is
// whatever its atomics produce); (2) no [X,Y,Z,3] vertex-index grid (12 B/cell of traffic and memory): a face looks its
// vertex ids up as v_offsets[owner cell] + rank of the edge's axis among the owner's crossings, recomputed from the
// grid.  Triangle table: `table` = GSDF_MC_TABLE_REFERENCE (0, default of the mirrors): the reference's own in-tree table
// (utils.cuh:31-289 packed into mc_table_ref.h) -> the same triangles per cell as the reference's mesh;
// GSDF_MC_TABLE_WATERTIGHT (1): the derived table of tools/gen_mc_table.py (same crossing edges in every configuration,
// triangulation that is watertight across ambiguous faces, which the classic table is not).
// Compiled with -ffp-contract=off: vertex positions are bit-identical to the numpy restatement (oracle/mc_oracle.py).
#include "common.h"
#include "mc_table.h"
#include "mc_table_ref.h"

namespace gsdf {

struct McDims {
  int rx, ry, rz;
};
__device__ __forceinline__ int64_t mc_lin(const McDims &d, int x, int y, int z) { return ((int64_t)x * d.ry + y) * d.rz + z; }

// crossing flags of the three edges a cell owns (bit 0: +x edge, bit 1: +y, bit 2: +z)
__device__ __forceinline__ int mc_owned(const McDims &d, const float *__restrict__ g, float thresh, int x, int y, int z) {
  const bool in0 = g[mc_lin(d, x, y, z)] > thresh;
  int m = 0;
  if (x < d.rx - 1 && (g[mc_lin(d, x + 1, y, z)] > thresh) != in0) m |= 1;
  if (y < d.ry - 1 && (g[mc_lin(d, x, y + 1, z)] > thresh) != in0) m |= 2;
  if (z < d.rz - 1 && (g[mc_lin(d, x, y, z + 1)] > thresh) != in0) m |= 4;
  return m;
}
__device__ __forceinline__ int mc_mask(const McDims &d, const float *__restrict__ g, float thresh, int x, int y, int z) {
  int mask = 0;
  if (g[mc_lin(d, x, y, z)] > thresh) mask |= 1;
  if (g[mc_lin(d, x + 1, y, z)] > thresh) mask |= 2;
  if (g[mc_lin(d, x + 1, y + 1, z)] > thresh) mask |= 4;
  if (g[mc_lin(d, x, y + 1, z)] > thresh) mask |= 8;
  if (g[mc_lin(d, x, y, z + 1)] > thresh) mask |= 16;
  if (g[mc_lin(d, x + 1, y, z + 1)] > thresh) mask |= 32;
  if (g[mc_lin(d, x + 1, y + 1, z + 1)] > thresh) mask |= 64;
  if (g[mc_lin(d, x, y + 1, z + 1)] > thresh) mask |= 128;
  return mask;
}
__device__ __forceinline__ int mc_edge(uint64_t packed, int k) { return (int)((packed >> (4 * k)) & 0xFull); }  // 15: end
__device__ __forceinline__ uint64_t mc_word(int table, int mask) { return table == 0 ? MC_TRI_PACKED_REF[mask] : MC_TRI_PACKED[mask]; }
__device__ __forceinline__ int mc_tri_count(int table, int mask) {
  const uint64_t w = mc_word(table, mask);
  int n = 0;
  while (n < 15 && mc_edge(w, n) != 15) n += 3;
  return n / 3;
}

__global__ void __launch_bounds__(256)
    mc_count_kernel(McDims d, int table, const float *__restrict__ g, float thresh, int32_t *__restrict__ n_vert,
                    int32_t *__restrict__ n_tri) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= (int64_t)d.rx * d.ry * d.rz) return;
  const int z = (int)(c % d.rz), y = (int)((c / d.rz) % d.ry), x = (int)(c / ((int64_t)d.rz * d.ry));
  n_vert[c] = __popc(mc_owned(d, g, thresh, x, y, z));
  n_tri[c] = (x < d.rx - 1 && y < d.ry - 1 && z < d.rz - 1) ? mc_tri_count(table, mc_mask(d, g, thresh, x, y, z)) : 0;
}

// owner cell offset and axis of the 12 cube edges (cumcubes_kernel.cu:180-191)
__device__ __constant__ static const int8_t MC_EDGE_OWNER[12][4] = {
    {0, 0, 0, 0}, {1, 0, 0, 1}, {0, 1, 0, 0}, {0, 0, 0, 1}, {0, 0, 1, 0}, {1, 0, 1, 1},
    {0, 1, 1, 0}, {0, 0, 1, 1}, {0, 0, 0, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {0, 1, 0, 2}};

__global__ void __launch_bounds__(256)
    mc_emit_kernel(McDims d, int table, const float *__restrict__ g, float thresh, const int64_t *__restrict__ v_off,
                   const int64_t *__restrict__ t_off, float sx, float sy, float sz, float lx, float ly, float lz,
                   float *__restrict__ vertices, int32_t *__restrict__ faces) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= (int64_t)d.rx * d.ry * d.rz) return;
  const int z = (int)(c % d.rz), y = (int)((c / d.rz) % d.ry), x = (int)(c / ((int64_t)d.rz * d.ry));
  // ---- vertices on the edges this cell owns, in axis order
  const int own = mc_owned(d, g, thresh, x, y, z);
  if (own) {
    const float d0 = g[c];
    int64_t v = v_off[c];
    const float scale[3] = {sx, sy, sz}, lower[3] = {lx, ly, lz};
    const int nb[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!(own & (1 << a))) continue;
      const float d1 = g[mc_lin(d, x + nb[a][0], y + nb[a][1], z + nb[a][2])];
      const float dt = (thresh - d0) / (d1 - d0);
      float pos[3] = {(float)x, (float)y, (float)z};
      pos[a] = pos[a] + dt;
#pragma unroll
      for (int k = 0; k < 3; ++k) vertices[3 * v + k] = pos[k] * scale[k] + lower[k];
      ++v;
    }
  }
  // ---- faces of the cube whose lowest corner this cell is
  if (x < d.rx - 1 && y < d.ry - 1 && z < d.rz - 1) {
    const int mask = mc_mask(d, g, thresh, x, y, z);
    int64_t f = t_off[c];
    const uint64_t tri = mc_word(table, mask);
    for (int k = 0; k < 15 && mc_edge(tri, k) != 15; ++k) {
      const int e = mc_edge(tri, k);
      const int ox = x + MC_EDGE_OWNER[e][0], oy = y + MC_EDGE_OWNER[e][1], oz = z + MC_EDGE_OWNER[e][2];
      const int axis = MC_EDGE_OWNER[e][3];
      const int o_own = mc_owned(d, g, thresh, ox, oy, oz);
      const int rank = __popc(o_own & ((1 << axis) - 1));
      faces[3 * f + k] = (int32_t)(v_off[mc_lin(d, ox, oy, oz)] + rank);
    }
  }
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_mc_count(int res_x, int res_y, int res_z, int table, const float *grid, float thresh, int32_t *n_vert,
                             int32_t *n_tri, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_mc_count");
  GSDF_REQUIRE(res_x >= 1 && res_y >= 1 && res_z >= 1, "mc_count: bad resolution");
  GSDF_REQUIRE(table == GSDF_MC_TABLE_REFERENCE || table == GSDF_MC_TABLE_WATERTIGHT, "mc_count: unknown triangle table %d", table);
  GSDF_REQUIRE(grid && n_vert && n_tri, "mc_count: null buffer");
  const McDims d = {res_x, res_y, res_z};
  const int64_t n = (int64_t)res_x * res_y * res_z;
  mc_count_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d, table, grid, thresh, n_vert, n_tri);
  GSDF_CHECK_LAUNCH("mc_count_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_mc_emit(int res_x, int res_y, int res_z, int table, const float *grid, float thresh, const int64_t *v_offsets,
                            const int64_t *t_offsets, const float *lower_host, const float *upper_host, float *vertices,
                            int32_t *faces, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_mc_emit");
  GSDF_REQUIRE(res_x >= 1 && res_y >= 1 && res_z >= 1, "mc_emit: bad resolution");
  GSDF_REQUIRE(table == GSDF_MC_TABLE_REFERENCE || table == GSDF_MC_TABLE_WATERTIGHT, "mc_emit: unknown triangle table %d", table);
  GSDF_REQUIRE(grid && v_offsets && t_offsets && lower_host && upper_host, "mc_emit: null buffer");
  const McDims d = {res_x, res_y, res_z};
  const int64_t n = (int64_t)res_x * res_y * res_z;
  const float sx = (upper_host[0] - lower_host[0]) / (float)res_x, sy = (upper_host[1] - lower_host[1]) / (float)res_y,
              sz = (upper_host[2] - lower_host[2]) / (float)res_z;
  mc_emit_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d, table, grid, thresh, v_offsets, t_offsets, sx, sy, sz,
                                                                  lower_host[0], lower_host[1], lower_host[2], vertices,
                                                                  faces);
  GSDF_CHECK_LAUNCH("mc_emit_kernel");
  return GSDF_OK;
}
