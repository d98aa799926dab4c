// micro checks for gfx950: DPP wave_shr semantics + VALU issue rates used by the DP kernel design
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_dpp(int* out){
  int l = threadIdx.x;
  int v = __builtin_amdgcn_update_dpp(-7, l*10, 0x138, 0xF, 0xF, false);
  out[l] = v;
}
template<int MODE> __global__ void k_rate(float* out, int iters, float e){
  float a0=threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
  for (int i=0;i<iters;i++){
    if (MODE==0){ // 8 independent adds
      a0+=e;a1+=e;a2+=e;a3+=e;a4+=e;a5+=e;a6+=e;a7+=e;
    } else if (MODE==1){ // dependent chain of adds
      a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;a0+=e;
    } else if (MODE==2){ // dependent add->max3 chain (DP column chain)
      a0=fmaxf(fmaxf(a1,a2),a0+e); a0=fmaxf(fmaxf(a3,a4),a0+e); a0=fmaxf(fmaxf(a5,a6),a0+e); a0=fmaxf(fmaxf(a7,a1),a0+e);
    } else if (MODE==3){ // two interleaved chains
      a0=fmaxf(fmaxf(a2,a3),a0+e); a1=fmaxf(fmaxf(a4,a5),a1+e); a0=fmaxf(fmaxf(a6,a7),a0+e); a1=fmaxf(fmaxf(a2,a3),a1+e);
    } else if (MODE==4){ // packed adds: 4 x float2
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 x0={a0,a1},x1={a2,a3},x2={a4,a5},x3={a6,a7}; f2 ee={e,e};
      x0+=ee;x1+=ee;x2+=ee;x3+=ee; a0=x0.x;a1=x0.y;a2=x1.x;a3=x1.y;a4=x2.x;a5=x2.y;a6=x3.x;a7=x3.y;
    } else if (MODE==5){ // independent max3
      a0=fmaxf(fmaxf(a0,a1),e);a2=fmaxf(fmaxf(a2,a3),e);a4=fmaxf(fmaxf(a4,a5),e);a6=fmaxf(fmaxf(a6,a7),e);
      a1=fmaxf(fmaxf(a1,a2),e);a3=fmaxf(fmaxf(a3,a4),e);a5=fmaxf(fmaxf(a5,a6),e);a7=fmaxf(fmaxf(a7,a0),e);
    }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7;
}
template<int MODE> void run(const char* name, int waves_per_simd, int ops_per_iter){
  float* d; hipMalloc(&d, 256*4*8*64*4);
  int iters=200000;
  dim3 grid(256* waves_per_simd), block(256);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k_rate<MODE><<<grid,block>>>(d, 1000, 1e-3f); hipDeviceSynchronize();
  hipEventRecord(a); k_rate<MODE><<<grid,block>>>(d, iters, 1e-3f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double instr_per_wave=(double)iters*ops_per_iter;
  double cyc=ms*1e-3*2.4e9; // at nominal 2.4GHz
  printf("%-28s waves/simd=%d  %.3f ms  -> %.2f cycles(@2.4GHz)/wave-instr/SIMD-slot\n", name, waves_per_simd, ms, cyc/(instr_per_wave*waves_per_simd));
  hipFree(d);
}
int main(){
  int* d; hipMalloc(&d,64*4); k_dpp<<<1,64>>>(d); std::vector<int> h(64); hipMemcpy(h.data(),d,256,hipMemcpyDeviceToHost);
  bool ok=true; for(int l=0;l<64;l++){int want = l==0? -7 : (l-1)*10; if(h[l]!=want){ok=false; printf("dpp lane %d got %d want %d\n",l,h[l],want);} }
  printf("dpp wave_shr:1 %s\n", ok?"OK":"BROKEN");
  for (int w: {1,2}) {
    run<0>("8 indep v_add", w, 8);
    run<1>("8 dep v_add chain", w, 8);
    run<2>("4x dep (add,max3)", w, 8);
    run<3>("2 chains (add,max3)", w, 8);
    run<4>("4 v_pk_add (8 adds)", w, 4);
    run<5>("8 v_max3", w, 8);
  }
  return 0;
}
