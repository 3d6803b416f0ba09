// micro-benchmark: integer atomics on a SMALL set of counters (design input for a segmented tile sort):
// 2.4 M increments spread over 8160 counters, non-returning (histogram) and returning (slot reservation).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t mix(uint32_t h){h^=h>>16;h*=0x85EBCA6Bu;h^=h>>13;h*=0xC2B2AE35u;h^=h>>16;return h;}
template<int RET> __global__ void k(int* cnt, int n_cnt, int64_t n, int* sink){
  int64_t t = (int64_t)blockIdx.x*blockDim.x+threadIdx.x;
  if(t>=n) return;
  // a splat covers ~3 neighbouring tiles: emulate with a random base tile per group of 3
  uint32_t idx = (mix((uint32_t)(t/3)) % (uint32_t)(n_cnt-2)) + (uint32_t)(t%3);
  if(RET){ int s = atomicAdd(cnt+idx,1); if(s==0x7fffffff) sink[0]=s; }
  else atomicAdd(cnt+idx,1);
}
int main(){
  const int n_cnt=8160; const int64_t n=2400000;
  int *cnt,*sink; hipMalloc(&cnt,n_cnt*4); hipMalloc(&sink,4); hipMemset(cnt,0,n_cnt*4);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int ret=0;ret<2;++ret){
    float ms=0;
    for(int rep=0;rep<3;++rep){
      hipEventRecord(a);
      if(ret) k<1><<<(unsigned)((n+255)/256),256>>>(cnt,n_cnt,n,sink); else k<0><<<(unsigned)((n+255)/256),256>>>(cnt,n_cnt,n,sink);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b);
    }
    printf("%s atomics: %lld increments on %d counters: %.3f ms (%.1f G/s)\n", ret?"returning":"non-returning",(long long)n,n_cnt,ms,n/ms/1e6);
  }
  return 0;
}
