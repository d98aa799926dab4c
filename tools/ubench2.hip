// VALU issue rate vs waves/SIMD with in-kernel cycle counters (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template<int MODE> __global__ void k_rate(float* out, long long* cyc, int iters, float e){
  float a0=threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
  long long c0 = __builtin_readcyclecounter();
  long long w0 = wall_clock64();
  for (int i=0;i<iters;i++){
    if (MODE==0){ a0+=e;a1+=e;a2+=e;a3+=e;a4+=e;a5+=e;a6+=e;a7+=e; a0+=e;a1+=e;a2+=e;a3+=e;a4+=e;a5+=e;a6+=e;a7+=e;}
    else if (MODE==1){ a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e; }
    else if (MODE==2){
      a0=fmaxf(fmaxf(a1,a2),a0+e); a0=fmaxf(fmaxf(a3,a4),a0+e); a0=fmaxf(fmaxf(a5,a6),a0+e); a0=fmaxf(fmaxf(a7,a1),a0+e);
      a0=fmaxf(fmaxf(a1,a2),a0+e); a0=fmaxf(fmaxf(a3,a4),a0+e); a0=fmaxf(fmaxf(a5,a6),a0+e); a0=fmaxf(fmaxf(a7,a1),a0+e);}
    else if (MODE==3){
      a0=fmaxf(fmaxf(a2,a3),a0+e); a1=fmaxf(fmaxf(a4,a5),a1+e); a0=fmaxf(fmaxf(a6,a7),a0+e); a1=fmaxf(fmaxf(a2,a3),a1+e);
      a0=fmaxf(fmaxf(a2,a3),a0+e); a1=fmaxf(fmaxf(a4,a5),a1+e); a0=fmaxf(fmaxf(a6,a7),a0+e); a1=fmaxf(fmaxf(a2,a3),a1+e);}
    else if (MODE==4){
      a0=fmaxf(fmaxf(a4,a5),a0+e); a1=fmaxf(fmaxf(a6,a7),a1+e); a2=fmaxf(fmaxf(a4,a5),a2+e); a3=fmaxf(fmaxf(a6,a7),a3+e);
      a0=fmaxf(fmaxf(a4,a5),a0+e); a1=fmaxf(fmaxf(a6,a7),a1+e); a2=fmaxf(fmaxf(a4,a5),a2+e); a3=fmaxf(fmaxf(a6,a7),a3+e);}
  }
  long long c1 = __builtin_readcyclecounter();
  long long w1 = wall_clock64();
  out[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
  if (threadIdx.x==0 && blockIdx.x==0){ cyc[0]=c1-c0; cyc[1]=w1-w0; }
}
template<int MODE> void run(const char* name, int wps){
  float* d; hipMalloc(&d, (size_t)256*wps*256*4); long long* dc; hipMalloc(&dc,16);
  int iters=2000000/ (wps>4?2:1);
  dim3 grid(256*wps), block(256);
  k_rate<MODE><<<grid,block>>>(d, dc, 1000, 1e-3f); hipDeviceSynchronize();
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); k_rate<MODE><<<grid,block>>>(d, dc, iters, 1e-3f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); long long h[2]; hipMemcpy(h,dc,16,hipMemcpyDeviceToHost);
  double ninstr=(double)iters*16;
  printf("%-22s w/simd=%d  %.2f ms  shader-cycles/instr(one wave)=%.2f  => per-SIMD cycles/instr=%.2f  clk=%.2f GHz (wallclk ticks %lld)\n",
    name,wps,ms,h[0]/ninstr,h[0]/ninstr/wps, h[0]/(ms*1e6), h[1]);
  hipFree(d); hipFree(dc);
}
int main(){
  for (int w: {1,2,3,4,8}) { run<0>("16 indep v_add",w); run<1>("16 dep v_add",w); run<2>("8x dep(add,max3)",w); run<3>("2 chains(add,max3)",w); run<4>("4 chains(add,max3)",w); }
}
