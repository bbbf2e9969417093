// Backward pass of the fused SingleConv block  y = LeakyReLU(conv3x3x3(GroupNorm(cat(skip, up(x)))))  and of the
// pooling / upsampling around it (first correct version: exact-fp32 matrix cores for the weight gradient,
// the forward conv kernels reused for the data gradient).  References: the reference trains through torch autograd
// over Trainer/models/unet3d/buildingblocks.py:31-60 (SingleConv 'gcl'), :185-186 (MaxPool3d), :265-276, 361-363
// (nearest upsample + concat).
//
//   dP  = dY * (Y > 0 ? 1 : slope)                                   lrelu_bwd
//   dW[co][ci][tap] = sum_v dP[v][co] * Xn[v + tap][ci]              conv_wgrad   (Xn = GN-applied input, 0 outside)
//   dXn = conv3x3x3(dP, W^T mirrored)                                forward conv kernel on transposed weights (host)
//   GroupNorm:  dbeta_c = sum_v dXn,  dgamma_c = sum_v dXn*xhat,
//               dx = rstd_g * (gamma_c*dXn - m1_g - xhat*m2_g),   m1 = mean_g(gamma*dXn), m2 = mean_g(gamma*dXn*xhat)
//   nearest upsample: the gradient of a low-res voxel is the sum over the box of its replicas.
#include "bfm_common.h"

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(n); i += (int64_t)gridDim.x * blockDim.x)

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ----------------------------------------------------------------------------- LeakyReLU
__global__ void lrelu_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int64_t n4, float slope,
                                 float* __restrict__ dP) {
    const float4* a = reinterpret_cast<const float4*>(dY);
    const float4* y = reinterpret_cast<const float4*>(Y);
    float4* o = reinterpret_cast<float4*>(dP);
    GRID_STRIDE(i, n4) {
        const float4 g = a[i], v = y[i];
        o[i] = make_float4(v.x > 0.f ? g.x : g.x * slope, v.y > 0.f ? g.y : g.y * slope, v.z > 0.f ? g.z : g.z * slope,
                           v.w > 0.f ? g.w : g.w * slope);
    }
}

// ----------------------------------------------------------------------------- weight gradient
struct WgParams {
    const float* dP;
    int Cout;
    const float *A, *B;
    int CA, CB;
    UpView up;
    const float *scale, *shift;
    int D, H, W;
    float* part;                      // [S][Cout][Cin][27]
    int S;
    int rows_per_split;               // (z,y) rows per split
};

// One wave per (row split, (kd,kh), 32x32 block of (co, ci)); the three kw taps share the dP fragment.
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain): A[i=co][k] = dP[voxel k][co], B[k][j=ci] = Xn[voxel k + tap][ci],
// K = two x-neighbouring voxels per step -- both fragments are plain coalesced channel-last reads.
__global__ void __launch_bounds__(64) conv_wgrad_kernel(const WgParams p) {
    const int lane = threadIdx.x;
    const int l32 = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x;
    const int kdh = blockIdx.y;
    const int kd = kdh / 3, kh = kdh - kd * 3;
    const int Cin = p.CA + p.CB;
    const int nci = Cin >> 5;
    const int cob = blockIdx.z / nci, cib = blockIdx.z - cob * nci;
    const int co = cob * 32 + l32;                         // A-fragment row of this lane
    const int ci = cib * 32 + l32;                         // B-fragment column of this lane
    const bool fromB = ci >= p.CA;
    const float sc = p.scale[ci], sh = p.shift[ci];

    floatx16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;

    const int nrows = p.D * p.H;
    const int r0 = split * p.rows_per_split;
    const int r1 = min(nrows, r0 + p.rows_per_split);
    for (int r = r0; r < r1; ++r) {
        const int z = r / p.H, y = r - z * p.H;
        const int zz = z + kd - 1, yy = y + kh - 1;
        const bool row_ok = zz >= 0 && zz < p.D && yy >= 0 && yy < p.H;    // wave-uniform
        if (!row_ok) continue;                                            // the whole input row is padding: adds 0
        const float* dprow = p.dP + ((int64_t)(z * p.H + y) * p.W) * p.Cout + co;
        const float* xrow;
        int xstride;
        if (!fromB) {
            xrow = p.A + ((int64_t)(zz * p.H + yy) * p.W) * p.CA + ci;
            xstride = p.CA;
        } else {
            xrow = p.B + ((int64_t)(p.up.mapD[zz] * p.up.h + p.up.mapH[yy]) * p.up.w) * p.CB + (ci - p.CA);
            xstride = p.CB;
        }
        for (int x = 0; x < p.W; x += 2) {
            const int xv = x + lh;                                        // this lane's voxel of the K pair
            const float a = xv < p.W ? dprow[(int64_t)xv * p.Cout] : 0.f;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = xv + kw - 1;
                float b = 0.f;
                if (xv < p.W && xx >= 0 && xx < p.W) {
                    const int xs = fromB ? p.up.mapW[xx] : xx;
                    b = fmaf(xrow[(int64_t)xs * xstride], sc, sh);
                }
                acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kw], 0, 0, 0);
            }
        }
    }
    float* out = p.part + (int64_t)split * p.Cout * Cin * 27;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;              // co within the block
            out[((int64_t)(cob * 32 + row) * Cin + ci) * 27 + kdh * 3 + kw] = acc[kw][i];
        }
}

// narrow inputs (the stem, Cin = 1): one block per (co, ci), threads stride over voxels, 27 taps each
__global__ void __launch_bounds__(256) conv_wgrad_narrow_kernel(const WgParams p) {
    const int Cin = p.CA + p.CB;
    const int co = blockIdx.x / Cin, ci = blockIdx.x - co * Cin;
    const float sc = p.scale[ci], sh = p.shift[ci];
    double acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.0;
    const int64_t nvox = (int64_t)p.D * p.H * p.W;
    for (int64_t v = threadIdx.x; v < nvox; v += 256) {
        const int x = (int)(v % p.W);
        const int64_t t2 = v / p.W;
        const int y = (int)(t2 % p.H), z = (int)(t2 / p.H);
        const float g = p.dP[v * p.Cout + co];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int zz = z + t / 9 - 1, yy = y + (t / 3) % 3 - 1, xx = x + t % 3 - 1;
            if (zz < 0 || zz >= p.D || yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) continue;
            const float xv = ci < p.CA ? p.A[((int64_t)(zz * p.H + yy) * p.W + xx) * p.CA + ci]
                                       : p.B[((int64_t)(p.up.mapD[zz] * p.up.h + p.up.mapH[yy]) * p.up.w + p.up.mapW[xx]) * p.CB + (ci - p.CA)];
            acc[t] += (double)g * (double)fmaf(xv, sc, sh);
        }
    }
    __shared__ double red[256];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        red[threadIdx.x] = acc[t];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) p.part[((int64_t)co * Cin + ci) * 27 + t] = (float)red[0];
        __syncthreads();
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int S, int64_t n, float* __restrict__ dW) {
    GRID_STRIDE(i, n) {
        float s = part[i];
        for (int k = 1; k < S; ++k) s += part[i + (int64_t)k * n];          // split order: deterministic
        dW[i] = s;
    }
}

// ----------------------------------------------------------------------------- GroupNorm backward
struct GnbParams {
    const float* dXn;                 // [D][H][W][Cin]
    const float *A, *B;
    int CA, CB;
    UpView up;
    int D, H, W, G;
    const float *mean, *rstd, *gamma;
};

// per-block partial sums over a run of voxels: part[blk][c][0] = sum dXn, [1] = sum dXn * xhat   (fp64)
__global__ void __launch_bounds__(256) gn_bwd_partial_kernel(const GnbParams p, int64_t vox_per_block,
                                                             double* __restrict__ part) {
    extern __shared__ double sm[];                        // [RP][C][2]
    const int C = p.CA + p.CB;
    const int t = threadIdx.x;
    const int CP = C < 256 ? C : 256;
    const int RP = 256 / CP;
    const int64_t nvox = (int64_t)p.D * p.H * p.W;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    const int cpg = C / p.G;
    for (int cb = 0; cb < C; cb += CP) {
        const int c = cb + t % CP, part_i = t / CP;
        double s1 = 0.0, s2 = 0.0;
        if (part_i < RP && c < C) {
            const float mu = p.mean[c / cpg], rs = p.rstd[c / cpg];
            for (int64_t v = v0 + part_i; v < v1; v += RP) {
                float xv;
                if (c < p.CA) {
                    xv = p.A[v * p.CA + c];
                } else {
                    const int x = (int)(v % p.W);
                    const int64_t t2 = v / p.W;
                    const int y = (int)(t2 % p.H), z = (int)(t2 / p.H);
                    xv = p.B[((int64_t)(p.up.mapD[z] * p.up.h + p.up.mapH[y]) * p.up.w + p.up.mapW[x]) * p.CB + (c - p.CA)];
                }
                const float g = p.dXn[v * C + c];
                s1 += (double)g;
                s2 += (double)g * (double)((xv - mu) * rs);
            }
        }
        sm[(t * 2)] = s1; sm[t * 2 + 1] = s2;
        __syncthreads();
        if (part_i == 0 && c < C) {
            for (int q = 1; q < RP; ++q) { s1 += sm[(q * CP + t % CP) * 2]; s2 += sm[(q * CP + t % CP) * 2 + 1]; }
            part[((int64_t)blockIdx.x * C + c) * 2] = s1;
            part[((int64_t)blockIdx.x * C + c) * 2 + 1] = s2;
        }
        __syncthreads();
    }
}

// one block: channel totals (rows in order) -> dgamma, dbeta; group means m1, m2
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const double* __restrict__ part, int nb, int C, int G,
                                                              double count_per_channel, const float* __restrict__ gamma,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ m1, float* __restrict__ m2) {
    extern __shared__ double sm[];                        // [C][2]: gamma*s1, gamma*s2
    const int t = threadIdx.x;
    for (int c = t; c < C; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int b = 0; b < nb; ++b) { s1 += part[((int64_t)b * C + c) * 2]; s2 += part[((int64_t)b * C + c) * 2 + 1]; }
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        sm[c * 2] = (double)gamma[c] * s1;
        sm[c * 2 + 1] = (double)gamma[c] * s2;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = t; g < G; g += 256) {
        double a = 0.0, b = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += sm[c * 2]; b += sm[c * 2 + 1]; }
        const double n = count_per_channel * (double)cpg;
        m1[g] = (float)(a / n);
        m2[g] = (float)(b / n);
    }
}

// dA[v][c] for the skip channels (c < CA)
__global__ void gn_bwd_apply_a_kernel(const GnbParams p, const float* __restrict__ m1, const float* __restrict__ m2,
                                      float* __restrict__ dA) {
    const int C = p.CA + p.CB;
    const int cpg = C / p.G;
    const int64_t n = (int64_t)p.D * p.H * p.W * p.CA;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % p.CA);
        const int64_t v = i / p.CA;
        const int g = c / cpg;
        const float xh = (p.A[i] - p.mean[g]) * p.rstd[g];
        dA[i] = p.rstd[g] * ((p.gamma[c] * p.dXn[v * C + c] - m1[g]) - xh * m2[g]);
    }
}

// dB[u][c] for the upsampled channels: sum over the box of replicas of low-res voxel u
// startD/H/W[u] = first full-res index mapped to u (exclusive prefix sum of the replication counts)
__global__ void gn_bwd_apply_b_kernel(const GnbParams p, const float* __restrict__ m1, const float* __restrict__ m2,
                                      const int32_t* __restrict__ startD, const int32_t* __restrict__ startH,
                                      const int32_t* __restrict__ startW, const int32_t* __restrict__ repD,
                                      const int32_t* __restrict__ repH, const int32_t* __restrict__ repW,
                                      float* __restrict__ dB) {
    const int C = p.CA + p.CB;
    const int cpg = C / p.G;
    const int64_t n = (int64_t)p.up.d * p.up.h * p.up.w * p.CB;
    GRID_STRIDE(i, n) {
        const int cb = (int)(i % p.CB);
        int64_t u = i / p.CB;
        const int ux = (int)(u % p.up.w); u /= p.up.w;
        const int uy = (int)(u % p.up.h);
        const int uz = (int)(u / p.up.h);
        const int c = p.CA + cb;
        const int g = c / cpg;
        float s = 0.f;
        for (int z = startD[uz]; z < startD[uz] + repD[uz]; ++z)
            for (int y = startH[uy]; y < startH[uy] + repH[uy]; ++y)
                for (int x = startW[ux]; x < startW[ux] + repW[ux]; ++x)
                    s += p.dXn[((int64_t)(z * p.H + y) * p.W + x) * C + c];
        const float cnt = (float)(repD[uz] * repH[uy] * repW[ux]);
        const float xh = (p.B[i] - p.mean[g]) * p.rstd[g];
        dB[i] = p.rstd[g] * ((p.gamma[c] * s - cnt * m1[g]) - cnt * (xh * m2[g]));
    }
}

// ----------------------------------------------------------------------------- MaxPool3d(2) backward
// dIn (zero-filled by this kernel) gets dOut at the first maximum of each 2x2x2 window in (dz,dy,dx) scan order
__global__ void maxpool2_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dOut, int C, int D, int H,
                                    int W, float* __restrict__ dIn) {
    const int d = D / 2, h = H / 2, w = W / 2;
    const int64_t n = (int64_t)D * H * W * C;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        int64_t v = i / C;
        const int x = (int)(v % W); v /= W;
        const int y = (int)(v % H);
        const int z = (int)(v / H);
        const int uz = z >> 1, uy = y >> 1, ux = x >> 1;
        float r = 0.f;
        if (uz < d && uy < h && ux < w) {
            // is (z,y,x) the first maximum of its window?
            const float me = in[i];
            bool first = true;
            for (int q = 0; q < 8 && first; ++q) {
                const int zz = 2 * uz + (q >> 2), yy = 2 * uy + ((q >> 1) & 1), xx = 2 * ux + (q & 1);
                const float o = in[((int64_t)(zz * H + yy) * W + xx) * C + c];
                const bool before = (zz < z) || (zz == z && (yy < y || (yy == y && xx < x)));
                if (o > me || (before && o == me)) first = false;
            }
            if (first) r = dOut[((int64_t)(uz * h + uy) * w + ux) * C + c];
        }
        dIn[i] = r;
    }
}

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256)); }

}  // namespace

extern "C" int bfm_lrelu_bwd(const float* dY, const float* Y, int64_t n, float slope, float* dP, bfm_stream_t stream) {
    if (!dY || !Y || !dP || n <= 0 || n % 4) return BFM_E_ARG;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, bfm_s(stream), dY, Y, n / 4, slope, dP);
    return bfm_launch_status();
}

extern "C" size_t bfm_conv3x3x3_wgrad_workspace(int Cin, int Cout, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (Cin % 32 || Cout % 32) return (size_t)Cout * Cin * 27 * sizeof(float);
    const int nrows = D * H;
    const int blocks = 9 * (Cout / 32) * (Cin / 32);
    int S = bfm_cdiv(2048, blocks);
    if (S > nrows) S = nrows;
    if (S < 1) S = 1;
    return (size_t)S * Cout * Cin * 27 * sizeof(float);
}

extern "C" int bfm_conv3x3x3_wgrad(const float* dP, int Cout, const float* A, int CA, const float* B, int CB, int D,
                                   int H, int W, const bfm_upsample_t* up, const float* scale, const float* shift,
                                   float* dW, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!dP || !A || !scale || !shift || !dW || !workspace || Cout <= 0 || CA <= 0 || D <= 0 || H <= 0 || W <= 0)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || !up->mapD || !up->mapH || !up->mapW))) return BFM_E_ARG;
    const int Cin = CA + CB;
    if (workspace_bytes < bfm_conv3x3x3_wgrad_workspace(Cin, Cout, D, H, W)) return BFM_E_WORKSPACE;
    WgParams p{};
    p.dP = dP; p.Cout = Cout; p.A = A; p.B = B; p.CA = CA; p.CB = CB; p.up = make_upview(up);
    p.scale = scale; p.shift = shift; p.D = D; p.H = H; p.W = W;
    p.part = static_cast<float*>(workspace);
    hipStream_t st = bfm_s(stream);
    const int64_t n = (int64_t)Cout * Cin * 27;
    if (Cin % 32 || Cout % 32 || CA % 32) {                      // narrow layers (the stem): double accumulation, one split
        p.S = 1;
        hipLaunchKernelGGL(conv_wgrad_narrow_kernel, dim3(Cout * Cin), dim3(256), 0, st, p);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(n)), dim3(256), 0, st, p.part, 1, n, dW);
        return bfm_launch_status();
    }
    const int nrows = D * H;
    const int blocks = 9 * (Cout / 32) * (Cin / 32);
    int S = bfm_cdiv(2048, blocks);
    if (S > nrows) S = nrows;
    if (S < 1) S = 1;
    p.rows_per_split = bfm_cdiv(nrows, S);
    p.S = bfm_cdiv(nrows, p.rows_per_split);
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(p.S, 9, (Cout / 32) * (Cin / 32)), dim3(64), 0, st, p);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(n)), dim3(256), 0, st, p.part, p.S, n, dW);
    return bfm_launch_status();
}

extern "C" size_t bfm_gn_bwd_workspace(int C, int D, int H, int W) {
    const int64_t nvox = (int64_t)D * H * W;
    int64_t nb = bfm_cdiv64(nvox, 2048);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (size_t)nb * C * 2 * sizeof(double) + 256;
}

// dXn [D][H][W][CA+CB] -> dA [D][H][W][CA], dB [d][h][w][CB] (when CB > 0), dgamma/dbeta [CA+CB].
// start*/rep* : per low-res index, the first full-res index and the number of full-res indices mapped to it.
extern "C" int bfm_gn_bwd(const float* dXn, const float* A, int CA, const float* B, int CB, int D, int H, int W,
                          const bfm_upsample_t* up, const int32_t* startD, const int32_t* startH, const int32_t* startW,
                          const float* mean, const float* rstd, const float* gamma, int G, float* dA, float* dB,
                          float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!dXn || !A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !mean || !rstd || !gamma || G <= 0 || !dA || !dgamma ||
        !dbeta || !workspace)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || !up->mapD || !up->repD || !startD || !startH || !startW || !dB)))
        return BFM_E_ARG;
    const int C = CA + CB;
    if (C % G) return BFM_E_SHAPE;
    if (workspace_bytes < bfm_gn_bwd_workspace(C, D, H, W)) return BFM_E_WORKSPACE;
    if ((size_t)C * 16 > 64 * 1024) return BFM_E_SHAPE;
    GnbParams p{};
    p.dXn = dXn; p.A = A; p.B = B; p.CA = CA; p.CB = CB; p.up = make_upview(up);
    p.D = D; p.H = H; p.W = W; p.G = G; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
    hipStream_t st = bfm_s(stream);
    const int64_t nvox = (int64_t)D * H * W;
    int64_t nb = bfm_cdiv64(nvox, 2048);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    const int64_t vpb = bfm_cdiv64(nvox, nb);
    nb = bfm_cdiv64(nvox, vpb);
    char* ws = static_cast<char*>(workspace);
    double* part = reinterpret_cast<double*>(ws);
    float* m1 = reinterpret_cast<float*>(ws + (size_t)nb * C * 16);
    float* m2 = m1 + 32;
    if (G > 32) return BFM_E_SHAPE;
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3((unsigned)nb), dim3(256), 256 * 16, st, p, vpb, part);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(1), dim3(256), (size_t)C * 16, st, part, (int)nb, C, G, (double)nvox,
                       gamma, dgamma, dbeta, m1, m2);
    hipLaunchKernelGGL(gn_bwd_apply_a_kernel, dim3(grid_for(nvox * CA)), dim3(256), 0, st, p, m1, m2, dA);
    if (CB > 0) {
        const int64_t nlo = (int64_t)up->d * up->h * up->w * CB;
        hipLaunchKernelGGL(gn_bwd_apply_b_kernel, dim3(grid_for(nlo)), dim3(256), 0, st, p, m1, m2, startD, startH, startW,
                           up->repD, up->repH, up->repW, dB);
    }
    return bfm_launch_status();
}

extern "C" int bfm_maxpool2_bwd(const float* in, const float* dOut, int C, int D, int H, int W, float* dIn,
                                bfm_stream_t stream) {
    if (!in || !dOut || !dIn || C <= 0 || D < 2 || H < 2 || W < 2) return BFM_E_ARG;
    const int64_t n = (int64_t)D * H * W * C;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), in, dOut, C, D, H, W, dIn);
    return bfm_launch_status();
}
