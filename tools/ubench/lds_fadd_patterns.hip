// micro-benchmark (round 6): what an LDS float accumulation costs on gfx950 by ACCESS PATTERN — design input for the compositing backward, whose
// per-(row, splat) gradient record is accumulated with ds_add_f32.  lds_atomic.hip measured 0.33 lanes per clock per CU on random addresses; here:
//   0 ds_add_f32, every lane its own word (conflict-free, linear)
//   1 ds_add_f32, the backward's pattern: 16-lane row r adds fields 0..15 of record t_r (21-float records, 4 different records per instruction)
//   2 ds_add_f32, quad pattern: lane quad q adds 4 fields of record t_q per instruction (16 different records per instruction)
//   3 ds_add_f32, random words
//   4 ds_add_f64 linear         5 ds_add_u32 linear        6 ds_add_u64 linear
//   7 plain read-modify-write (ds_read_b32 + v_add + ds_write_b32), linear
//   8 plain RMW in the backward's pattern (pattern 1 without atomicity)
//   9 ds_add_rtn_f32 linear
// Every lane issues ITER operations; the record / word it touches rotates with the iteration so that consecutive operations of a lane do not
// hit the same address.  Output: lane-operations per clock per CU (2.4 GHz assumed) and T lane-ops/s chip-wide.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
static constexpr int NREC = 216, NF = 21, ITER = 4096;
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out) {
  __shared__ double tile64[NREC * NF / 2 + 64];
  float *tile = reinterpret_cast<float *>(tile64);
  uint32_t *ti = reinterpret_cast<uint32_t *>(tile64);
  unsigned long long *tl = reinterpret_cast<unsigned long long *>(tile64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NREC * NF; i += 256) tile[i] = 0.f;
  __syncthreads();
  float acc = 0.f;
  for (int it = 0; it < ITER; ++it) {
    const uint32_t h = mix((uint32_t)it * 2654435761u + blockIdx.x);
    const float v = 1.0f + (float)(it & 3);
    if (MODE == 0) atomicAdd(&tile[(tid + 256 * (it & 7)) % (NREC * NF)], v);
    else if (MODE == 1 || MODE == 8) {
      const int row = lane >> 4, f = lane & 15;
      const int t = (int)((h >> (4 * row + 2 * wave)) + 53u * row + 17u * wave) % NREC;
      if (MODE == 1) atomicAdd(&tile[t * NF + f], v);
      else tile[t * NF + f] += v;
    } else if (MODE == 2) {
      const int q = lane >> 2, j = lane & 3;
      const int t = (int)((h >> (q & 7)) + 13u * q + 17u * wave) % NREC;
      atomicAdd(&tile[t * NF + 4 * j + (it & 3)], v);
    } else if (MODE == 3) atomicAdd(&tile[mix(h + tid) % (NREC * NF)], v);
    else if (MODE == 4) atomicAdd(&tile64[(tid + 256 * (it & 7)) % (NREC * NF / 2)], (double)v);
    else if (MODE == 5) atomicAdd(&ti[(tid + 256 * (it & 7)) % (NREC * NF)], (uint32_t)it);
    else if (MODE == 6) atomicAdd(&tl[(tid + 256 * (it & 7)) % (NREC * NF / 2)], (unsigned long long)it);
    else if (MODE == 7) tile[(tid + 256 * (it & 7)) % (NREC * NF)] += v;
    else if (MODE == 9) acc += atomicAdd(&tile[(tid + 256 * (it & 7)) % (NREC * NF)], v);
  }
  __syncthreads();
  float s = acc;
  for (int i = tid; i < NREC * NF; i += 256) s += tile[i];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float *out; hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 8;
  const char *names[] = {"ds_add_f32 linear", "ds_add_f32 bwd rows (4 rec x 16 fields)", "ds_add_f32 quads (16 rec x 4 fields)", "ds_add_f32 random",
                         "ds_add_f64 linear", "ds_add_u32 linear", "ds_add_u64 linear", "plain RMW linear", "plain RMW bwd rows", "ds_add_rtn_f32 linear"};
  for (int mode = 0; mode < 10; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      switch (mode) {
        case 0: k<0><<<blocks, 256>>>(out); break; case 1: k<1><<<blocks, 256>>>(out); break; case 2: k<2><<<blocks, 256>>>(out); break;
        case 3: k<3><<<blocks, 256>>>(out); break; case 4: k<4><<<blocks, 256>>>(out); break; case 5: k<5><<<blocks, 256>>>(out); break;
        case 6: k<6><<<blocks, 256>>>(out); break; case 7: k<7><<<blocks, 256>>>(out); break; case 8: k<8><<<blocks, 256>>>(out); break;
        default: k<9><<<blocks, 256>>>(out);
      }
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    const double ops = (double)blocks * 256 * ITER;
    printf("%-42s: %8.3f ms  %7.3f T lane-ops/s  %6.2f lanes/clk/CU\n", names[mode], ms, ops / ms / 1e9, ops / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
