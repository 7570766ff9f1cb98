// Micro-probe: issue rate of v_mfma_f32_16x16x32_f16 in the accumulate patterns the scan uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MF(a,b,c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a,b,c,0,0,0)
template <int MODE>
__global__ __launch_bounds__(256) void k(const half8* in, floatx4* out, unsigned long long* t, int iters){
  half8 a0 = in[threadIdx.x], a1 = in[threadIdx.x+256], b0 = in[threadIdx.x+512], b1 = in[threadIdx.x+768];
  if (MODE & 8) { asm volatile("" : "+a"(a0)); asm volatile("" : "+a"(a1)); }
  floatx4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if ((MODE & 7) == 0) { // one chain
      c0 = MF(a0,b0,c0); c0 = MF(a1,b1,c0); c0 = MF(a0,b1,c0); c0 = MF(a1,b0,c0); c0 = MF(a0,b0,c0); c0 = MF(a1,b1,c0);
    } else if ((MODE & 7) == 1) { // ahi, amid, amid
      c0 = MF(a0,b0,c0); c1 = MF(a0,b1,c1); c1 = MF(a1,b0,c1); c0 = MF(a0,b0,c0); c1 = MF(a0,b1,c1); c1 = MF(a1,b0,c1);
    } else if ((MODE & 7) == 2) { // two tiles, tile-major
      c0 = MF(a0,b0,c0); c1 = MF(a0,b1,c1); c1 = MF(a1,b0,c1); c2 = MF(a0,b0,c2); c3 = MF(a0,b1,c3); c3 = MF(a1,b0,c3);
    } else if ((MODE & 7) == 4) { // ahi, amidA, amidB (three accumulators)
      c0 = MF(a0,b0,c0); c1 = MF(a0,b1,c1); c2 = MF(a1,b0,c2); c0 = MF(a0,b0,c0); c1 = MF(a0,b1,c1); c2 = MF(a1,b0,c2);
    } else if ((MODE & 7) == 5) { // amid, ahi, amid (dependent pair separated)
      c1 = MF(a0,b1,c1); c0 = MF(a0,b0,c0); c1 = MF(a1,b0,c1); c1 = MF(a0,b1,c1); c0 = MF(a0,b0,c0); c1 = MF(a1,b0,c1);
    } else if ((MODE & 7) == 6) { // two tiles interleaved: ahi0 ahi1 amid0 amid1 amid0 amid1
      c0 = MF(a0,b0,c0); c2 = MF(a0,b1,c2); c1 = MF(a0,b1,c1); c3 = MF(a0,b0,c3); c1 = MF(a1,b0,c1); c3 = MF(a1,b1,c3);
    } else if ((MODE & 7) == 3) { // four independent
      c0 = MF(a0,b0,c0); c1 = MF(a0,b1,c1); c2 = MF(a1,b0,c2); c3 = MF(a1,b1,c3); c0 = MF(a0,b1,c0); c1 = MF(a1,b0,c1);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c0 + c1 + c2 + c3;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
template <int MODE> void run(const char* name, half8* in, floatx4* out, unsigned long long* t, int waves){
  int iters = 2000; unsigned long long h;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64*waves), 0, 0, in, out, t, iters);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64*waves), 0, 0, in, out, t, iters);
  (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-34s waves=%d cycles/MFMA = %.2f\n", name, waves, (double)h / (iters * 6.0));
}
int main(){
  half8* in; floatx4* out; unsigned long long* t;
  (void)hipMalloc(&in, 1024*16); (void)hipMemset(in, 0x3c, 1024*16); (void)hipMalloc(&out, 512*16); (void)hipMalloc(&t, 8);
  for (int waves : {4, 8}) {
    run<0>("one chain, A in VGPR", in, out, t, waves);
    run<1>("ahi/amid/amid, A in VGPR", in, out, t, waves);
    run<2>("two tiles tile-major, A in VGPR", in, out, t, waves);
    run<3>("four independent, A in VGPR", in, out, t, waves);
    run<12>("ahi/amidA/amidB, A in AGPR", in, out, t, waves);
    run<13>("amid/ahi/amid, A in AGPR", in, out, t, waves);
    run<14>("two tiles interleaved, A in AGPR", in, out, t, waves);
    run<4>("ahi/amidA/amidB, A in VGPR", in, out, t, waves);
    run<8>("one chain, A in AGPR", in, out, t, waves);
    run<9>("ahi/amid/amid, A in AGPR", in, out, t, waves);
    run<10>("two tiles tile-major, A in AGPR", in, out, t, waves);
    run<11>("four independent, A in AGPR", in, out, t, waves);
  }
  return 0;
}
