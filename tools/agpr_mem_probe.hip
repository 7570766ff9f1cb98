// Micro-probe: vector-memory and LDS instructions with AGPR data operands under an EXEC mask set inside the asm statement
// (the row buffers of scan_duo.hip): global_load_dwordx4 -> a, ds_write_b128 <- a, ds_read_b128 -> a, global_store_dwordx4 <- a.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *in, float *out, float *out2)
{
    __shared__ __attribute__((aligned(16))) float sm[64 * 4 * 2];
    const int lane = threadIdx.x;
    floatx4 r = {-1.f, -2.f, -3.f, -4.f};
    asm volatile("" : "+a"(r));
    const unsigned long long m = __builtin_amdgcn_ballot_w64((lane & 1) == 0); // even lanes load
    const unsigned go = lane * 16;
    unsigned long long sv;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %2\n\tglobal_load_dwordx4 %1, %3, %4\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv), "+a"(r) : "s"(m), "v"(go), "s"(in) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float *)sm + lane * 16;
    asm volatile("ds_write_b128 %0, %1" ::"v"(la), "a"(r) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    floatx4 q = {-9.f, -9.f, -9.f, -9.f};
    asm volatile("" : "+a"(q));
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(lane < 32);
    const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) float *)sm + ((lane + 1) & 63) * 16;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %2\n\tds_read_b128 %1, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv), "+a"(q) : "s"(m2), "v"(lb) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(go), "a"(q), "s"(out) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(go), "a"(r), "s"(out2) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
int main()
{
    float h[256], o[256], o2[256];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *din, *dout, *dout2;
    (void)hipMalloc(&din, 1024); (void)hipMalloc(&dout, 1024); (void)hipMalloc(&dout2, 1024);
    (void)hipMemcpy(din, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dout2);
    (void)hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2, dout2, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        // r: even lanes loaded in[4l..], odd lanes keep -1..-4
        for (int j = 0; j < 4; ++j) {
            const float er = (l & 1) == 0 ? (float)(4 * l + j) : -(float)(j + 1);
            if (o2[4 * l + j] != er) { if (bad < 8) printf("r mismatch lane %d j %d: %f vs %f\n", l, j, o2[4 * l + j], er); ++bad; }
            const int s = (l + 1) & 63;
            const float eq = l < 32 ? ((s & 1) == 0 ? (float)(4 * s + j) : -(float)(j + 1)) : -9.f;
            if (o[4 * l + j] != eq) { if (bad < 8) printf("q mismatch lane %d j %d: %f vs %f\n", l, j, o[4 * l + j], eq); ++bad; }
        }
    }
    printf("agpr memory operands under EXEC masks: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return 0;
}
