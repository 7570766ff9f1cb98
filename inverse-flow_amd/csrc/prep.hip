// Small per-call preparation kernels: the algebraic pre-fold of the in-pixel channel solve
// into the taps (DESIGN.md "fold"), the effective (masked) weight, log|det|, kernel flips.
//
// The exact solver couples the C channels of one pixel sequentially
// (inf/utils/solve_mc.py:96-109: `for c` ... `if c - k_c < 0: break`).  With L the unit (or
// general) lower-triangular diagonal-tap matrix,  L z_p = x_p - sum_t What_t z_{p-t}  becomes
//   z_p = L^-1 x_p - sum_t (L^-1 What_t) z_{p-t}
// so every channel of every pixel of an anti-diagonal is independent and each step of the scan
// is one dense contraction.  L^-1 and the products are formed in fp64 and rounded once.
#include "ifl_common.h"

namespace ifl {

// stored index of the weight element that multiplies the source pixel at logical offset
// (dh,dw) (target p reads p-(dh,dw)):  TL tap (KH-1-dh, KW-1-dw), reflected for the order.
__device__ __forceinline__ size_t w_index(int co, int ci, int dh, int dw, int C, int KH, int KW, int flipH,
                                          int flipW)
{
    int kh = KH - 1 - dh, kw = KW - 1 - dw;
    if (flipH) kh = KH - 1 - kh;
    if (flipW) kw = KW - 1 - kw;
    return (((size_t)co * C + ci) * KH + kh) * KW + kw;
}

// effective diagonal-tap entry L[i][k]
__device__ __forceinline__ double l_entry(const float *w, int i, int k, const Geom &g)
{
    if (k > i) return 0.0;
    if (k == i) return g.general_diag ? (double)w[w_index(i, i, 0, 0, g.C, g.KH, g.KW, g.flipH, g.flipW)] : 1.0;
    return (double)w[w_index(i, k, 0, 0, g.C, g.KH, g.KW, g.flipH, g.flipW)];
}

// One thread per column j of L^-1: forward substitution, reading back its own column.
__global__ void k_linv(const float *__restrict__ w, double *__restrict__ linv, Geom g)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int C = g.C;
    if (j >= C) return;
    for (int i = 0; i < C; ++i) {
        double s = (i == j) ? 1.0 : 0.0;
        if (i < j) {
            linv[(size_t)i * C + j] = 0.0;
            continue;
        }
        for (int k = j; k < i; ++k) s -= l_entry(w, i, k, g) * linv[(size_t)k * C + j];
        linv[(size_t)i * C + j] = s / l_entry(w, i, i, g);
    }
}

// Same recurrence with L and the columns of L^-1 kept in LDS (C <= 96): one block; all threads
// stage L, then thread j owns column j and only ever reads back its own column.
__global__ __launch_bounds__(256) void k_linv_lds(const float *__restrict__ w, double *__restrict__ linv, Geom g)
{
    extern __shared__ double sl[]; // [C*C] L^-1 as [i][j], then [C*C] floats of L
    const int C = g.C;
    float *sL = (float *)(sl + (size_t)C * C);
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) sL[idx] = (float)l_entry(w, idx / C, idx % C, g);
    __syncthreads();
    const int j = threadIdx.x;
    if (j >= C) return;
    for (int i = 0; i < C; ++i) {
        double v = 0.0;
        if (i >= j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; ++k) s -= (double)sL[i * C + k] * sl[k * C + j];
            v = s / (double)sL[i * C + i];
        }
        sl[i * C + j] = v;
        linv[(size_t)i * C + j] = v;
    }
}

// L^-1 for wide layers (C a multiple of 16, e.g. the C = 256 layers of the ImageNet-32 Glow), X kept in global memory:
// the blocked scheme of k_foldpack (scan_mfma.hip) -- diagonal 16x16 blocks by substitution, one thread per column;
// off-diagonal blocks one block-distance at a time, X_ij = -X_ii (sum_k L_ik X_kj) on the fp64 matrix cores, one
// wave per block (layouts: tools/mfma_f64_layout_probe.hip).  One workgroup of 1024 threads; a round's results are
// published to the other waves by a fence + barrier.  The column-serial kernel above took 4.3 ms at C = 256.
__global__ __launch_bounds__(1024) void k_linv_blocked(const float *__restrict__ w, double *__restrict__ linv, Geom g)
{
    typedef double doublex4 __attribute__((ext_vector_type(4)));
    const int C = g.C, NBK = C / 16;
    const int tid = threadIdx.x, wv = tid / 64, lf = tid % 64, li = lf % 16, lk = lf / 16, NWV = blockDim.x / 64;
    for (int idx = tid; idx < C * C; idx += blockDim.x) linv[idx] = 0.0;
    __threadfence();
    __syncthreads();
    for (int j = tid; j < C; j += blockDim.x) { // diagonal blocks: column j inside its block
        const int r0 = (j / 16) * 16, jj = j % 16;
        double col[16];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            double a0 = (ii == jj) ? 1.0 : 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                if (kk < ii && kk >= jj) a0 -= l_entry(w, r0 + ii, r0 + kk, g) * col[kk];
            col[ii] = ii >= jj ? a0 / l_entry(w, r0 + ii, r0 + ii, g) : 0.0;
            linv[(size_t)(r0 + ii) * C + j] = col[ii];
        }
    }
    __threadfence();
    __syncthreads();
    for (int dist = 1; dist < NBK; ++dist) {
        const int npairs = NBK - dist;
        for (int pr = wv; pr < npairs; pr += NWV) { // (wave-uniform trip count per wave)
            const int bi = pr + dist, bj = pr;
            doublex4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb = bj; kb < bi; ++kb)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const int k = 16 * kb + 4 * kc + lk;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(l_entry(w, 16 * bi + li, k, g), linv[(size_t)k * C + 16 * bj + li], acc, 0, 0, 0);
                }
            doublex4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) // X_ii is lower triangular: its upper entries are stored zeros
                res = __builtin_amdgcn_mfma_f64_16x16x4f64(linv[(size_t)(16 * bi + li) * C + 16 * bi + 4 * kc + lk], acc[kc], res, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) linv[(size_t)(16 * bi + 4 * v + lk) * C + 16 * bj + li] = -res[v];
        }
        __threadfence(); // this round's blocks are read by other waves in the next one
        __syncthreads();
    }
}

int launch_linv(const float *w, double *linv, const Geom &g, hipStream_t s)
{
    if (g.C > 96 && g.C % 16 == 0) {
        hipLaunchKernelGGL(k_linv_blocked, dim3(1), dim3(1024), 0, s, w, linv, g);
    } else if (g.C <= 96) {
        const size_t lds = (size_t)g.C * g.C * (sizeof(double) + sizeof(float));
        static LdsOptIn opt_in;
        if (int rc = lds_opt_in(opt_in, (const void *)k_linv_lds, 96 * 96 * 12)) return rc;
        hipLaunchKernelGGL(k_linv_lds, dim3(1), dim3(256), lds, s, w, linv, g);
    } else {
        const int T = 64;
        hipLaunchKernelGGL(k_linv, dim3((g.C + T - 1) / T), dim3(T), 0, s, w, linv, g);
    }
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

// wf[t][kc][c], one thread per element.
__global__ void k_fold(const float *__restrict__ w, const double *__restrict__ linv, float *__restrict__ wf,
                       Geom g, int transposed)
{
    const int C = g.C;
    const size_t total = (size_t)g.KH * g.KW * C * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int kc = (int)((i / C) % C);
        const int t = (int)(i / ((size_t)C * C));
        const int dh = t / g.KW, dw = t % g.KW;
        double acc;
        if (t == 0) {
            acc = transposed ? linv[(size_t)kc * C + c] : linv[(size_t)c * C + kc];
        } else {
            acc = 0.0;
            if (!transposed) {
                // (L^-1 What_t)[c][kc] = sum_{m<=c} Linv[c][m] w[m][kc][t]
                for (int m = 0; m <= c; ++m)
                    acc += linv[(size_t)c * C + m] *
                           (double)w[w_index(m, kc, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)];
            } else {
                // (L^-T What_t^T)[c][kc] = sum_{m>=c} Linv[m][c] w[kc][m][t]
                for (int m = c; m < C; ++m)
                    acc += linv[(size_t)m * C + c] *
                           (double)w[w_index(kc, m, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)];
            }
        }
        wf[i] = (float)acc;
    }
}

// Small layers (C <= 32: the 4- and 8-channel layers of the MNIST Glow run 64 of these preparations a step): L^-1 and the
// folded taps in ONE launch -- k_linv_lds's recurrence, then k_fold's sums with L^-1 read from LDS; the same arithmetic in the
// same order, the same bits.
__global__ __launch_bounds__(256) void k_linv_fold_small(const float *__restrict__ w, double *__restrict__ linv, float *__restrict__ wf,
                                                         Geom g, int transposed)
{
    __shared__ double sl[32 * 32];
    __shared__ float sL[32 * 32];
    const int C = g.C;
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) sL[idx] = (float)l_entry(w, idx / C, idx % C, g);
    __syncthreads();
    const int j = threadIdx.x;
    if (j < C) {
        for (int i = 0; i < C; ++i) {
            double v = 0.0;
            if (i >= j) {
                double a = (i == j) ? 1.0 : 0.0;
                for (int k = j; k < i; ++k) a -= (double)sL[i * C + k] * sl[k * C + j];
                v = a / (double)sL[i * C + i];
            }
            sl[i * C + j] = v;
            linv[(size_t)i * C + j] = v;
        }
    }
    __syncthreads();
    const int total = g.KH * g.KW * C * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int c = i % C, kc = (i / C) % C, t = i / (C * C);
        const int dh = t / g.KW, dw = t % g.KW;
        double acc;
        if (t == 0) {
            acc = transposed ? sl[kc * C + c] : sl[c * C + kc];
        } else {
            acc = 0.0;
            if (!transposed) {
                for (int m = 0; m <= c; ++m) acc += sl[c * C + m] * (double)w[w_index(m, kc, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)];
            } else {
                for (int m = c; m < C; ++m) acc += sl[m * C + c] * (double)w[w_index(kc, m, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)];
            }
        }
        wf[i] = (float)acc;
    }
}

int launch_linv_fold(const float *w, double *linv, float *wf, const Geom &g, int transposed, hipStream_t s)
{
    if (g.C <= 32) {
        hipLaunchKernelGGL(k_linv_fold_small, dim3(1), dim3(256), 0, s, w, linv, wf, g, transposed);
        IFL_HIP(hipGetLastError());
        return IFL_OK;
    }
    if (int rc = launch_linv(w, linv, g, s)) return rc;
    return launch_fold(w, linv, wf, g, transposed, s);
}

int launch_fold(const float *w, const double *linv, float *wf, const Geom &g, int transposed, hipStream_t s)
{
    const size_t total = (size_t)g.KH * g.KW * g.C * g.C;
    const int T = 256;
    size_t blocks = (total + T - 1) / T;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fold, dim3((unsigned)blocks), dim3(T), 0, s, w, linv, wf, g, transposed);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

__global__ void k_effw(const float *__restrict__ w, float *__restrict__ weff, Geom g)
{
    const int C = g.C;
    const size_t total = (size_t)C * C * g.KH * g.KW;
    // stored position of the diagonal tap
    const int dkh = g.flipH ? 0 : g.KH - 1, dkw = g.flipW ? 0 : g.KW - 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kw = (int)(i % g.KW);
        const int kh = (int)((i / g.KW) % g.KH);
        const int ci = (int)((i / ((size_t)g.KW * g.KH)) % C);
        const int co = (int)(i / ((size_t)g.KW * g.KH * C));
        float v = w[i];
        if (kh == dkh && kw == dkw) {
            if (ci > co) v = 0.f;
            else if (ci == co && !g.general_diag) v = 1.f;
        }
        weff[i] = v;
    }
}

int launch_effw(const float *w, float *weff, const Geom &g, hipStream_t s)
{
    const size_t total = (size_t)g.C * g.C * g.KH * g.KW;
    const int T = 256;
    size_t blocks = (total + T - 1) / T;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_effw, dim3((unsigned)blocks), dim3(T), 0, s, w, weff, g);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

__global__ void k_logdet(const float *__restrict__ w, float *__restrict__ out, Geom g)
{
    // single block; fixed-order tree reduction over channels -> deterministic
    __shared__ double red[256];
    double s = 0.0;
    if (g.general_diag)
        for (int c = threadIdx.x; c < g.C; c += blockDim.x)
            s += log(fabs((double)w[w_index(c, c, 0, 0, g.C, g.KH, g.KW, g.flipH, g.flipW)]));
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float v = (float)(red[0] * (double)g.H * (double)g.W);
    for (int b = threadIdx.x; b < g.B; b += blockDim.x) out[b] = v;
}

int launch_logdet(const float *w, float *out, const Geom &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_logdet, dim3(1), dim3(256), 0, s, w, out, g);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

__global__ void k_flip_kernel(const float *__restrict__ w, float *__restrict__ wt, int Co, int Ci, int KH, int KW)
{
    const size_t total = (size_t)Co * Ci * KH * KW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kw = (int)(i % KW);
        const int kh = (int)((i / KW) % KH);
        const int co = (int)((i / ((size_t)KW * KH)) % Co);
        const int ci = (int)(i / ((size_t)KW * KH * Co));
        wt[i] = w[(((size_t)co * Ci + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    }
}

int launch_flip_kernel(const float *w, float *wt, int Co, int Ci, int KH, int KW, hipStream_t s)
{
    const size_t total = (size_t)Co * Ci * KH * KW;
    const int T = 256;
    size_t blocks = (total + T - 1) / T;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_flip_kernel, dim3((unsigned)blocks), dim3(T), 0, s, w, wt, Co, Ci, KH, KW);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

__global__ void k_recon_mix(const float *__restrict__ dx, const float *__restrict__ x, const float *__restrict__ az,
                            float *__restrict__ t, float coef, float *__restrict__ loss, float loss_scale, size_t n)
{
    __shared__ float red[256];
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float r = x[i] - az[i];
        r = (r == r) ? r : 0.f; // NaN -> 0 (inf/layers/selfnorm.py:212)
        t[i] = dx[i] + coef * r;
        s += r * r;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) atomicAdd(loss, red[0] * loss_scale);
}

int launch_recon_mix(const float *dx, const float *x, const float *az, float *t, float coef, float *loss,
                     float loss_scale, size_t n, hipStream_t s)
{
    const int T = 256;
    size_t blocks = (n + T - 1) / T;
    if (blocks > 2048) blocks = 2048;
    if (loss) IFL_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
    hipLaunchKernelGGL(k_recon_mix, dim3((unsigned)blocks), dim3(T), 0, s, dx, x, az, t, coef, loss, loss_scale, n);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
