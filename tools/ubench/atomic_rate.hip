// micro-benchmark: fp32 global atomic-add throughput vs footprint and scope (design input for hashgrid bwd)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t h){h^=h>>16;h*=0x85EBCA6Bu;h^=h>>13;h*=0xC2B2AE35u;h^=h>>16;return h;}
template<int SCOPE> __global__ void k(float* buf, uint32_t mask, int per_thread, int pair){
  uint32_t t = blockIdx.x*blockDim.x+threadIdx.x;
  for(int i=0;i<per_thread;++i){
    uint32_t idx = mix(t*977u+i*131071u) & mask;
    if(pair) idx &= ~1u;
    if(SCOPE==0){ atomicAdd(buf+idx, 1.0f); if(pair) atomicAdd(buf+idx+1, 1.0f);}
    else if(SCOPE==1){ __hip_atomic_fetch_add(buf+idx,1.0f,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); if(pair) __hip_atomic_fetch_add(buf+idx+1,1.0f,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP);}
    else { buf[idx] += 1.0f; if(pair) buf[idx+1]+=1.0f; }   // plain RMW (racy) as an upper bound of the memory path
  }
}
int main(){
  const size_t maxb = 256u<<20; float* buf; hipMalloc(&buf,maxb); hipMemset(buf,0,maxb);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks=256*16, threads=256, per=16;
  for(int pair=0;pair<2;++pair) for(int scope=0;scope<3;++scope) for(size_t bytes: {size_t(64)<<10,size_t(1)<<20,size_t(4)<<20,size_t(32)<<20,size_t(64)<<20,size_t(256)<<20}){
    uint32_t mask = (uint32_t)(bytes/4-1);
    for(int rep=0;rep<2;++rep){
      hipEventRecord(a);
      if(scope==0) k<0><<<blocks,threads>>>(buf,mask,per,pair); else if(scope==1) k<1><<<blocks,threads>>>(buf,mask,per,pair); else k<2><<<blocks,threads>>>(buf,mask,per,pair);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms,a,b);
    double n = double(blocks)*threads*per*(pair?2:1);
    printf("pair=%d scope=%s footprint=%6zu KB : %.3f ms  %.1f G atomics/s\n",pair,scope==0?"agent":scope==1?"wg   ":"plain",bytes>>10,ms,n/ms/1e6);
  }
  return 0;
}
