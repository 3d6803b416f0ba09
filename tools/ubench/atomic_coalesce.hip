// micro-benchmark: does the fp32 atomic path process a cache line per request?  lanes grouped in runs of G
// consecutive floats at a random (line-aligned) base inside a 64 MB footprint.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t mix(uint32_t h){h^=h>>16;h*=0x85EBCA6Bu;h^=h>>13;h*=0xC2B2AE35u;h^=h>>16;return h;}
template<int G> __global__ void k(float* buf, uint32_t mask, int per_thread){
  uint32_t t = blockIdx.x*blockDim.x+threadIdx.x;
  uint32_t grp = t / G, lane_in = t % G;
  for(int i=0;i<per_thread;++i){
    uint32_t base = (mix(grp*977u+i*131071u) & mask) & ~(uint32_t)(G-1);
    atomicAdd(buf+base+lane_in, 1.0f);
  }
}
int main(){
  const size_t bytes = 64u<<20; float* buf; hipMalloc(&buf,bytes); hipMemset(buf,0,bytes);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks=256*16, threads=256, per=16; uint32_t mask=(uint32_t)(bytes/4-1);
#define RUN(G) for(int rep=0;rep<2;++rep){hipEventRecord(a); k<G><<<blocks,threads>>>(buf,mask,per); hipEventRecord(b); hipEventSynchronize(b);} \
  { float ms; hipEventElapsedTime(&ms,a,b); double n=double(blocks)*threads*per; printf("run of %2d consecutive floats per group: %.3f ms  %.1f G atomics/s\n",G,ms,n/ms/1e6);}
  RUN(1) RUN(2) RUN(4) RUN(8) RUN(16) RUN(32) RUN(64)
  return 0;
}
