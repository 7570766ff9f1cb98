// Round-trip latency of a flag hand-off between two workgroups (gfx950): workgroup A writes k, B answers k, N rounds.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pingpong tools/pingpong_probe.hip && /tmp/pingpong
// Variants: scope of the atomics (agent / system) and placement (same XCD: workgroup ids 0 and 8; different XCDs: 0 and 1;
// workgroup ids go round the 8 XCDs).  Also a data hand-off: 2 KB written with plain stores + release fence, flag, acquire.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE> __global__ void k_ping(unsigned long long *f, int partner, int n, long long *cycles, unsigned *xcc)
{
    const int me = blockIdx.x;
    if (me != 0 && me != partner) return;
    if (threadIdx.x != 0) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[me == 0 ? 0 : 1] = id & 15;
    unsigned long long *mine = f + (me == 0 ? 0 : 32), *theirs = f + (me == 0 ? 32 : 0);
    const long long t0 = wall_clock64();
    for (int k = 1; k <= n; ++k) {
        if (me == 0) {
            __hip_atomic_store(mine, (unsigned long long)k, __ATOMIC_RELAXED, SCOPE);
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE) < (unsigned long long)k) {}
        } else {
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE) < (unsigned long long)k) {}
            __hip_atomic_store(mine, (unsigned long long)k, __ATOMIC_RELAXED, SCOPE);
        }
    }
    if (me == 0) *cycles = wall_clock64() - t0;
}

// one-way chain with payload: stage s waits for flag[s-1] == k, reads 2 KB, writes 2 KB, vmcnt(0), sets flag[s]
template <int SCOPE> __global__ void k_chain(unsigned long long *f, unsigned long long *data, int stride, int nstage, int n,
                                             long long *cycles)
{
    const int st = blockIdx.x / stride;
    if (blockIdx.x % stride != 0 || st >= nstage) return;
    const int lane = threadIdx.x; // 64 threads
    unsigned long long *in = data + (size_t)((st + nstage - 1) % nstage) * 512, *out = data + (size_t)st * 512;
    const long long t0 = wall_clock64();
    for (int k = 1; k <= n; ++k) {
        // token k travels stage 0 -> 1 -> ... -> nstage-1 -> 0 (stage 0 starts round k after the last stage finished k-1)
        const unsigned long long want = st == 0 ? (unsigned long long)(k - 1) : (unsigned long long)k;
        while (__hip_atomic_load(f + 32 * ((st + nstage - 1) % nstage), __ATOMIC_RELAXED, SCOPE) < want) {}
        unsigned long long acc = 0;
        for (int j = 0; j < 4; ++j) acc += __hip_atomic_load(in + j * 64 + lane, __ATOMIC_RELAXED, SCOPE);
        for (int j = 0; j < 4; ++j) __hip_atomic_store(out + j * 64 + lane, acc + k, __ATOMIC_RELAXED, SCOPE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(f + 32 * st, (unsigned long long)k, __ATOMIC_RELAXED, SCOPE);
    }
    if (st == 0 && lane == 0) *cycles = wall_clock64() - t0;
}

int main()
{
    unsigned long long *f, *data;
    long long *cyc;
    unsigned *xcc;
    hipMalloc(&f, 4096 * 8);
    hipMalloc(&data, 64 * 512 * 8);
    hipMalloc(&cyc, 8);
    hipMalloc(&xcc, 8);
    const int n = 2000;
    for (int scope = 0; scope < 2; ++scope)
        for (int partner : {8, 1, 4}) {
            hipMemset(f, 0, 4096 * 8);
            if (scope == 0) hipLaunchKernelGGL(k_ping<__HIP_MEMORY_SCOPE_AGENT>, dim3(16), dim3(64), 0, 0, f, partner, n, cyc, xcc);
            else hipLaunchKernelGGL(k_ping<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(16), dim3(64), 0, 0, f, partner, n, cyc, xcc);
            long long c;
            unsigned x[2];
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
            printf("ping-pong %s scope, workgroups 0 and %d (XCC %u and %u): %.0f ns per round trip\n", scope ? "system" : "agent ",
                   partner, x[0], x[1], c * 10.0 / n); // wall_clock64: 100 MHz
        }
    for (int scope = 0; scope < 2; ++scope)
        for (int stride : {8, 1}) {
            const int nstage = 8;
            hipMemset(f, 0, 4096 * 8);
            hipMemset(data, 0, 64 * 512 * 8);
            if (scope == 0)
                hipLaunchKernelGGL(k_chain<__HIP_MEMORY_SCOPE_AGENT>, dim3(nstage * stride), dim3(64), 0, 0, f, data, stride, nstage, n, cyc);
            else
                hipLaunchKernelGGL(k_chain<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(nstage * stride), dim3(64), 0, 0, f, data, stride, nstage, n, cyc);
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("chain of %d stages with 2 KB payload, %s scope, %s: %.0f ns per stage\n", nstage, scope ? "system" : "agent ",
                   stride == 8 ? "one XCD" : "eight XCDs", c * 10.0 / n / nstage);
        }
    return 0;
}
