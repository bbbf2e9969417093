// Direct 3x3x3 convolution, exact fp32 FMA, any channel count.
//
// SingleConv 'gcl' body after the statistics: folded GroupNorm affine on load,
// zero padding after the affine, conv (no bias), LeakyReLU
// (Trainer/models/unet3d/buildingblocks.py:31-60).  Used for the Cin=1 stem
// (HBM/L2-bound: K = 27 is not a GEMM) and for widths the MFMA family does not
// take; also serves as the exact-fp32 device reference for the MFMA kernel.
//
// One thread = one output voxel x 8 output channels; blockIdx.y = channel
// group, so the weight addresses are wave-uniform (scalar loads).
#include "bfm_common.h"

namespace {

constexpr int TPB = 256;
constexpr int COT = 8;

__global__ void pack_direct(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ out) {
    // out[tap][ci][co] = w[co][ci][tap]
    int64_t n = (int64_t)27 * Cin * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int co = (int)(i % Cout);
        int64_t r = i / Cout;
        int ci = (int)(r % Cin);
        int tap = (int)(r / Cin);
        out[i] = w[((int64_t)co * Cin + ci) * 27 + tap];
    }
}

__global__ void __launch_bounds__(TPB) conv_direct(const float* __restrict__ A, int CA, const float* __restrict__ B,
                                                   int CB, int D, int H, int W, UpView up,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ wp, int Cout, float slope,
                                                   float* __restrict__ out) {
    const int64_t nvox = (int64_t)D * H * W;
    const int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x;
    const int co0 = blockIdx.y * COT;
    const int Cin = CA + CB;
    const bool live = v < nvox;
    const int64_t vv = live ? v : 0;
    const int x = (int)(vv % W);
    const int64_t t = vv / W;
    const int y = (int)(t % H);
    const int z = (int)(t / H);

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.f;

    for (int kd = 0; kd < 3; ++kd) {
        for (int kh = 0; kh < 3; ++kh) {
            for (int kw = 0; kw < 3; ++kw) {
                const int zz = z + kd - 1, yy = y + kh - 1, xx = x + kw - 1;
                const bool inb = live && zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
                const int tap = (kd * 3 + kh) * 3 + kw;
                const float* wt = wp + (size_t)tap * Cin * Cout + co0;
                int64_t ia = 0, ib = 0;
                if (inb) {
                    ia = (((int64_t)zz * H + yy) * W + xx) * CA;
                    if (CB > 0) ib = (((int64_t)up.mapD[zz] * up.h + up.mapH[yy]) * up.w + up.mapW[xx]) * CB;
                }
                for (int ci = 0; ci < Cin; ++ci) {
                    float xv = 0.f;
                    if (inb) {
                        float raw = ci < CA ? A[ia + ci] : B[ib + (ci - CA)];
                        xv = fmaf(raw, scale[ci], shift[ci]);
                    }
                    const float* wr = wt + (size_t)ci * Cout;
#pragma unroll
                    for (int j = 0; j < COT; ++j) {
                        float wv = (co0 + j < Cout) ? wr[j] : 0.f;
                        acc[j] = fmaf(xv, wv, acc[j]);
                    }
                }
            }
        }
    }
    if (!live) return;
    float* o = out + v * Cout + co0;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        if (co0 + j < Cout) {
            float r = acc[j];
            o[j] = r >= 0.f ? r : r * slope;
        }
    }
}

// Cin == 1 stem (enc0.1: 1 -> 32 @ up to 160^3) as a K = 32 GEMM on the matrix core: the 27 taps (+5 zero
// columns) are the K dimension, 32 consecutive voxels of one x-run the rows, Cout the columns.  The layer is
// bound by its 128 B/voxel output; with v_mfma_f32_32x32x16_f16 (hi/lo split like conv_mfma, fp32-grade) the
// arithmetic is ~6 MFMAs per 32 voxels and every store instruction writes two full 128-byte output lines.
typedef _Float16 half8s __attribute__((ext_vector_type(8)));
typedef float floatx16s __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x2s __attribute__((ext_vector_type(2)));

template <int NB>
__global__ void __launch_bounds__(256, 3) conv_stem_mfma(const float* __restrict__ A, int D, int H, int W,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ bound,
                                                         const float* __restrict__ wp /*[27][Cout]*/, float slope,
                                                         float* __restrict__ out, int64_t nblocks32,
                                                         double* __restrict__ rsum, double* __restrict__ rsq,
                                                         float* __restrict__ rmn, float* __restrict__ rmx) {
    constexpr int Cout = NB * 32;
    const int lane = threadIdx.x & 63;
    const int l32 = lane & 31, kh = lane >> 5;
    // operand scales (powers of two): activations from the GroupNorm bound, weights from max|w|
    float bmax = bound[0];
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) { int ex; (void)frexpf(bmax, &ex); aexp = 14 - ex; aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp); }
    float wmax = 0.f;
    for (int i = lane; i < 27 * Cout; i += 64) wmax = fmaxf(wmax, fabsf(wp[i]));
    wmax = wave_reduce_max(wmax);
    int wexp = 0;
    if (wmax > 0.f && wmax < INFINITY) { int ex; (void)frexpf(wmax, &ex); wexp = 14 - ex; wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp); }
    const float sa = ldexpf(1.f, aexp), sw = ldexpf(1.f, wexp), dq = ldexpf(1.f, -(aexp + wexp));
    const float sc = scale[0] * sa, sh = shift[0] * sa;
    const float dqs = dq * slope;

    // B fragments: lane holds B[k = 16*ks + 8*kh + j][col = nb*32 + l32], k = tap index (>= 27 -> 0)
    half8s bhi[NB][2], blo[NB][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * kh + j;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float w = k < 27 ? wp[k * Cout + nb * 32 + l32] * sw : 0.f;
                const _Float16 h = (_Float16)w;
                bhi[nb][ks][j] = h;
                blo[nb][ks][j] = (_Float16)(w - (float)h);
            }
        }

    const int xb = (W + 31) >> 5;                               // 32-voxel row blocks per x-run
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;

    // gather of one row block: raw input values (the GroupNorm affine is applied later, zero padding stays exact).
    // The kernel is bound by its vector instructions (profiles/r03_hbm_kernels.txt), so everything that does not depend
    // on the block is computed once per lane: tap q's element offset from the block's own voxel (toff) and the set of
    // volume faces the tap must stay inside of (need: bit 0/1 z-1/z+1, 2/3 y-1/y+1, 4/5 x-1/x+1, 6 always -- taps >= 27
    // never load, 7 always -- the lane's own x must be inside the row); per block a tap costs one add, one test, one select.
    int toff[16];
    unsigned need[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = 16 * (q >> 3) + (q & 7) + (kh ? 8 : 0);
        const int dz = k / 9 - 1, dy = (k % 9) / 3 - 1, dx = k % 3 - 1;
        toff[q] = k < 27 ? (dz * H + dy) * W + dx : 0;
        unsigned n = 0x80u;
        if (dz < 0) n |= 1u; if (dz > 0) n |= 2u;
        if (dy < 0) n |= 4u; if (dy > 0) n |= 8u;
        if (dx < 0) n |= 16u; if (dx > 0) n |= 32u;
        if (k >= 27) n |= 64u;
        need[q] = n;
    }
    auto gather = [&](int bx, int y, int z, float (&raw)[16], unsigned& mask) __attribute__((always_inline)) {
        const int x = bx * 32 + l32;
        const int base = (z * H + y) * W + x;                      // the launcher keeps D*H*W below 2^31
        // faces this voxel's neighbours would cross (bit set = that neighbour does not exist)
        const unsigned bad = (z == 0 ? 1u : 0u) | (z == D - 1 ? 2u : 0u) | (y == 0 ? 4u : 0u) | (y == H - 1 ? 8u : 0u) |
                             (x == 0 ? 16u : 0u) | (x >= W - 1 ? 32u : 0u) | 64u | (x >= W ? 0x80u : 0u);
        mask = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool inb = (need[q] & bad) == 0;
            raw[q] = A[inb ? base + toff[q] : 0];                  // a tap outside reads element 0 and is masked below
            mask |= inb ? (1u << q) : 0u;
        }
    };

    // optional moment rows of the output: one row per wave ([gridDim.x*4][Cout]), see conv3d_mfma.hip.  Per block the
    // <= 16 values a lane stores per column go through fp32 partials (as conv_mfma's epilogue does with <= 32), then fp64.
    double ms[NB], mq[NB];
    float mmn[NB], mmx[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { ms[nb] = 0.0; mq[nb] = 0.0; mmn[nb] = INFINITY; mmx[nb] = -INFINITY; }

    // block -> (x block, y, z): divided once, then stepped by the (constant) stride in mixed radix
    const int s_bx = (int)(nw % xb), s_y = (int)((nw / xb) % H), s_z = (int)((nw / xb) / H);
    int nbx = (int)(gw % xb), ny = (int)((gw / xb) % H), nz = (int)((gw / xb) / H);
    auto advance = [&]() __attribute__((always_inline)) {
        nbx += s_bx;
        int c = nbx >= xb ? 1 : 0;
        nbx -= c ? xb : 0;
        ny += s_y + c;
        c = ny >= H ? 1 : 0;
        ny -= c ? H : 0;
        nz += s_z + c;
    };
    float raw[16];
    unsigned mask = 0;
    if (gw < nblocks32) gather(nbx, ny, nz, raw, mask);
    for (int64_t blk = gw; blk < nblocks32; blk += nw) {
        const int bx = nbx;
        const int64_t t = (int64_t)nz * H + ny;
        advance();
        float nraw[16];
        unsigned nmask = 0;
        if (blk + nw < nblocks32) gather(nbx, ny, nz, nraw, nmask);    // in flight while this block is multiplied / stored
        // x = hi + lo by truncation, two values per instruction (v_cvt_pkrtz: x - hi is exact in fp32), as conv3d_wino.hip
        half8s ahi[2], alo[2];
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (mask >> q) & 1u ? fmaf(raw[q], sc, sh) : 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v0 = v[8 * ks + 2 * j], v1 = v[8 * ks + 2 * j + 1];
                const fp16x2s h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
                const fp16x2s l = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h[0], v1 - (float)h[1]);
                hw[j] = __builtin_bit_cast(unsigned, h);
                lw[j] = __builtin_bit_cast(unsigned, l);
            }
            ahi[ks] = __builtin_bit_cast(half8s, make_uint4(hw[0], hw[1], hw[2], hw[3]));
            alo[ks] = __builtin_bit_cast(half8s, make_uint4(lw[0], lw[1], lw[2], lw[3]));
        }
        float* orow = out + (t * W + bx * 32) * Cout;               // t = z*H + y
        const bool whole = bx * 32 + 32 <= W;                       // wave-uniform: no per-row test on a full block
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            floatx16s acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], bhi[nb][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], blo[nb][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bhi[nb][ks], acc, 0, 0, 0);
            }
            float fs = 0.f, fq = 0.f;
            auto emit = [&](bool check) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = (i & 3) + 8 * (i >> 2) + 4 * kh;     // C/D row of this register
                    if (!check || bx * 32 + rr < W) {
                        // LeakyReLU and the dequantisation in one multiply: dq is a power of two, so
                        // acc * (dq * slope) rounds as (acc * dq) * slope does
                        const float r = acc[i] * (acc[i] >= 0.f ? dq : dqs);
                        orow[rr * Cout + nb * 32 + l32] = r;
                        if (rsum) {
                            fs += r; fq = fmaf(r, r, fq);
                            mmn[nb] = fminf(mmn[nb], r); mmx[nb] = fmaxf(mmx[nb], r);
                        }
                    }
                }
            };
            if (whole) emit(false); else emit(true);
            if (rsum) { ms[nb] += (double)fs; mq[nb] += (double)fq; }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) raw[q] = nraw[q];
        mask = nmask;
    }
    if (rsum) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            ms[nb] += __shfl_xor(ms[nb], 32);
            mq[nb] += __shfl_xor(mq[nb], 32);
            mmn[nb] = fminf(mmn[nb], __shfl_xor(mmn[nb], 32));
            mmx[nb] = fmaxf(mmx[nb], __shfl_xor(mmx[nb], 32));
            if (kh == 0) {
                const size_t o = (size_t)gw * Cout + nb * 32 + l32;
                rsum[o] = ms[nb]; rsq[o] = mq[nb]; rmn[o] = mmn[nb]; rmx[o] = mmx[nb];
            }
        }
    }
}

}  // namespace

extern "C" size_t bfm_pack_conv_weights_direct_bytes(int Cin, int Cout) {
    return (size_t)27 * Cin * Cout * sizeof(float);
}

extern "C" int bfm_pack_conv_weights_direct(const float* w, int Cin, int Cout, float* wpacked, bfm_stream_t stream) {
    if (!w || !wpacked || Cin <= 0 || Cout <= 0) return BFM_E_ARG;
    int64_t n = (int64_t)27 * Cin * Cout;
    int nb = (int)std::min<int64_t>(2048, bfm_cdiv64(n, 256));
    hipLaunchKernelGGL(pack_direct, dim3(nb), dim3(256), 0, bfm_s(stream), w, Cin, Cout, wpacked);
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_direct(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                                    const bfm_upsample_t* up, const float* scale, const float* shift,
                                    const float* wpacked, int Cout, float slope, float* out, bfm_stream_t stream) {
    if (!A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift || !wpacked || Cout <= 0 || !out)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || !up->mapD || !up->mapH || !up->mapW))) return BFM_E_ARG;
    int64_t nvox = (int64_t)D * H * W;
    dim3 grid((unsigned)bfm_cdiv64(nvox, TPB), (unsigned)bfm_cdiv(Cout, COT));
    hipLaunchKernelGGL(conv_direct, grid, dim3(TPB), 0, bfm_s(stream), A, CA, B, CB, D, H, W, make_upview(up), scale,
                       shift, wpacked, Cout, slope, out);
    return bfm_launch_status();
}

namespace {
int64_t stem_grid(int D, int H, int W) {
    const int64_t nblk = (int64_t)D * H * ((W + 31) / 32);
    int64_t nb = bfm_cdiv64(nblk, 4);
    if (nb > 256 * 16) nb = 256 * 16;
    return nb;
}
}  // namespace

// rows of the output-moment table bfm_conv3x3x3_stem_ex writes (one per wave of the launch)
extern "C" int bfm_conv3x3x3_stem_rows(int D, int H, int W) {
    if (D <= 0 || H <= 0 || W <= 0) return 0;
    return (int)(stem_grid(D, H, W) * 4);
}

extern "C" int bfm_conv3x3x3_stem_ex(const float* A, int D, int H, int W, const float* scale, const float* shift,
                                     const float* bound, const float* wpacked_direct, int Cout, float slope,
                                     float* out, void* moment_rows, bfm_stream_t stream) {
    if (!A || D <= 0 || H <= 0 || W <= 0 || !scale || !shift || !bound || !wpacked_direct || !out) return BFM_E_ARG;
    if (Cout != 32 && Cout != 64) return BFM_E_SHAPE;
    if ((int64_t)D * H * W >= ((int64_t)1 << 31)) return BFM_E_SHAPE;          // the kernel indexes the input with 32 bits
    const int64_t nblk = (int64_t)D * H * ((W + 31) / 32);
    const int64_t nb = stem_grid(D, H, W);
    double *rsum = nullptr, *rsq = nullptr;
    float *rmn = nullptr, *rmx = nullptr;
    if (moment_rows) {
        if (reinterpret_cast<uintptr_t>(moment_rows) & 7) return BFM_E_ARG;
        char* rb = static_cast<char*>(moment_rows);
        const size_t n = (size_t)nb * 4 * Cout;
        rsum = reinterpret_cast<double*>(rb);
        rsq = reinterpret_cast<double*>(rb + n * 8);
        rmn = reinterpret_cast<float*>(rb + n * 16);
        rmx = reinterpret_cast<float*>(rb + n * 20);
    }
    if (Cout == 32)
        hipLaunchKernelGGL(conv_stem_mfma<1>, dim3((unsigned)nb), dim3(256), 0, bfm_s(stream), A, D, H, W, scale, shift,
                           bound, wpacked_direct, slope, out, nblk, rsum, rsq, rmn, rmx);
    else
        hipLaunchKernelGGL(conv_stem_mfma<2>, dim3((unsigned)nb), dim3(256), 0, bfm_s(stream), A, D, H, W, scale, shift,
                           bound, wpacked_direct, slope, out, nblk, rsum, rsq, rmn, rmx);
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_stem(const float* A, int D, int H, int W, const float* scale, const float* shift,
                                  const float* bound, const float* wpacked_direct, int Cout, float slope, float* out,
                                  bfm_stream_t stream) {
    return bfm_conv3x3x3_stem_ex(A, D, H, W, scale, shift, bound, wpacked_direct, Cout, slope, out, nullptr, stream);
}
