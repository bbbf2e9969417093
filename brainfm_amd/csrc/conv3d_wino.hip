// 3x3x3 convolution with a Winograd F(2,3) transform along x: 1.5x fewer matrix-core FLOPs for the same result.
//
// The conv_mfma family is bound by what the matrix pipe sustains under its power limit, so the only
// way up is to issue fewer MFMAs.  Along x, two neighbouring outputs (x0, x0+1) of a 3-tap correlation need 4
// products instead of 6:
//      d = in[x0-1 .. x0+2]            V = (d0-d2, d1+d2, d2-d1, d1-d3)
//      g = w[.., kw=0..2]              U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//      m_p = sum over (kd,kh,ci) of V_p * U_p          y0 = m0+m1+m2,  y1 = m1-m2-m3
// so the GEMM becomes 4 "positions" x (M/2 output pairs) x K = 9 (kd,kh) taps x Cin: 4*9/2 = 18 tap-rows per
// output voxel instead of 27.  The transforms are fp32 additions (input: at staging time, after the GroupNorm affine
// and zero padding; output: in the epilogue); the products keep conv_mfma's split-fp16 three-pass scheme, so the
// result is fp32-grade (transform rounding ~1e-7 relative on top of the 2^-22 product error).
//
// Workgroup = 4 waves = the 4 positions of one box of 256 output voxels (128 pairs) x 64 couts; each wave holds
// 4x2 blocks of v_mfma_f32_32x32x16_f16 accumulators (128 pair-rows x 64 cols).  The transformed, split halo'd box
// lives in LDS ([pos][k-half][hi|lo][row][pair][8 ch]); each wave's weights (its position's U) are private and
// stream L2 -> VGPR two taps ahead.  Epilogue: the four m_p meet in LDS, one thread per (pair, cout) forms y0/y1,
// dequantises, optionally adds what `out` holds (accumulate mode, the skip half of an up-folded decoder conv),
// applies LeakyReLU and stores 128-byte row segments.  Single-source inputs only (CB == 0), no split-K.
#include "bfm_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;
constexpr int NTHR = 256;
constexpr int MLD = 33;            // epilogue LDS row stride in floats (odd: conflict-free)

struct WinoParams {
    const float* A;
    int CA, D, H, W;
    const float *scale, *shift, *bound;
    int G;
    const uint4* wp;
    int wexp, Cout;
    float slope;
    float* out;
    int accum;
    int TD, TH, TW, HT, PW;          // box, halo'd rows per slice, pairs per row
    int pw_shift, thp_shift;         // log2(PW), log2(TH*PW)
    int nTy, nTx, nMt, NT, KCN;
    int npos_lds, plane_stride;      // (TD+2)*HT*PW positions; bytes per plane
};

__device__ __forceinline__ int row_perm(int l) {        // as conv_mfma: each 16-lane b128 group reads 16 consecutive positions
    if (l < 4) return l;
    if (l < 12) return l + 12;
    if (l < 16) return l - 8;
    if (l < 20) return l + 8;
    if (l < 28) return l - 12;
    return l;
}

__device__ __forceinline__ void pair_coords(const WinoParams& p, int q, int& d, int& h, int& j) {
    d = q >> p.thp_shift;
    const int rem = q & ((1 << p.thp_shift) - 1);
    h = rem >> p.pw_shift;
    j = rem & ((1 << p.pw_shift) - 1);
}

template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino(const WinoParams p) {
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int NF = 2 * NPL;
    constexpr int BATCH = (NPASS == 3) ? 1 : 3;      // staging items loaded together (register budget: 128 accumulators)
    constexpr int NSET = 3;                           // weight register sets: NSET-1 taps ahead
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int pos = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = Winograd position 0..3
    const int l32 = lane & 31, khalf = lane >> 5;

    int bid = blockIdx.x;
    {
        const int nblk = p.nMt * p.NT;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / p.NT, nt = bid - mt * p.NT;
    const int tx = mt % p.nTx;
    const int ty = (mt / p.nTx) % p.nTy;
    const int tz = mt / (p.nTx * p.nTy);
    const int z0 = tz * p.TD, y0 = ty * p.TH, x0 = tx * p.TW;

    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, p.bound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        aexp = 13 - ex;                                            // |V| <= 2 * bound
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    // A base offsets of the four 32-row blocks: pair (d,h,j), tap (kd,kh)=(0,0) reads halo row (d, h), pair j
    int a_off[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        int d, h, j;
        pair_coords(p, mb * 32 + row_perm(l32), d, h, j);
        a_off[mb] = ((pos * 2 + khalf) * NPL) * p.plane_stride + ((d * p.HT + h) * p.PW + j) * 16;
    }

    // staging items: e = tid + it*NTHR -> (halo row, pair, channel quad); off0 = element offset of voxel
    // (gz, gy, x0 + 2j - 1) channel 0 (may point outside the row: the mask says which of the 4 x positions exist)
    constexpr int MAX_IT = 5;
    const int n_el = p.npos_lds * 4;
    const int q4 = tid & 3;
    int off0[MAX_IT];
    int msk[MAX_IT];                                               // bit i: x position i inside the volume; -1: no item
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int e = tid + it * NTHR;
        off0[it] = 0;
        msk[it] = -1;
        if (e < n_el) {
            const int ps = e >> 2;
            const int j = ps & ((1 << p.pw_shift) - 1);
            const int r = ps >> p.pw_shift;
            const int hz = r / p.HT, hy = r - hz * p.HT;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + 2 * j - 1;
            int m = 0;
            if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (gx + i >= 0 && gx + i < p.W) m |= 1 << i;
            }
            msk[it] = m;
            off0[it] = ((gz * p.H + gy) * p.W + gx) * p.CA;
        }
    }
    const int st_plane = ((q4 >> 1) * NPL) * p.plane_stride + (q4 & 1) * 8;

    floatx16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    // this wave's weight stream: S = KCN*9 steps (chunk-major, (kd,kh)-minor), NF fragments of 64 x uint4 per step;
    // ring of three register sets, two steps ahead (9 % 3 == 0: the set index is the tap index mod 3)
    const int S = p.KCN * 9;
    const uint4* wbase = p.wp + (size_t)(nt * 4 + pos) * S * (NF * 64) + lane;
    uint4 wq[NSET][NF];
    auto fetch = [&](int s, uint4 (&dst)[NF]) __attribute__((always_inline)) {
        const int sc = s < S ? s : S - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f] = wbase[(size_t)sc * (NF * 64) + f * 64];
    };
    fetch(0, wq[0]);
    if constexpr (NSET == 3) fetch(1, wq[1]);

    auto do_chunk = [&](int kc, auto par_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        const int c0 = kc * KC;
        const float* src = p.A + c0 + q4 * 4;
        const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + c0 + q4 * 4);
        const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + c0 + q4 * 4);
        const float sc[4] = {sc4.x * a_scale, sc4.y * a_scale, sc4.z * a_scale, sc4.w * a_scale};
        const float sh[4] = {sh4.x * a_scale, sh4.y * a_scale, sh4.z * a_scale, sh4.w * a_scale};
        __syncthreads();                                 // previous chunk's readers are done
        // batches of BATCH items: all global loads of a batch are in flight before the first is consumed
#pragma unroll
        for (int b0 = 0; b0 < MAX_IT; b0 += BATCH) {
            float4 v[BATCH][4];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int it = b0 + u;
                if (it >= MAX_IT) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (msk[it] >= 0 && (msk[it] & (1 << i)))
                        v[u][i] = *reinterpret_cast<const float4*>(src + off0[it] + i * p.CA);
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int it = b0 + u;
                if (it >= MAX_IT) break;
                if (msk[it] < 0) continue;
                float dd[4][4];                          // [x position][channel]: affine, zero padding after it
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = msk[it] & (1 << i);
                    const float y[4] = {v[u][i].x, v[u][i].y, v[u][i].z, v[u][i].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) dd[i][c] = ok ? fmaf(y[c], sc[c], sh[c]) : 0.f;
                }
                const int e = tid + it * NTHR;
                unsigned char* dst = lds + st_plane + (e >> 2) * 16;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    half4 hi, lo;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float t = ps == 0 ? dd[0][c] - dd[2][c]
                                      : ps == 1 ? dd[1][c] + dd[2][c]
                                      : ps == 2 ? dd[2][c] - dd[1][c]
                                                : dd[1][c] - dd[3][c];
                        const _Float16 hh = (_Float16)t;
                        hi[c] = hh;
                        lo[c] = (_Float16)(t - (float)hh);
                    }
                    unsigned char* dp = dst + (ps * 2 * NPL) * p.plane_stride;
                    *reinterpret_cast<half4*>(dp) = hi;
                    if constexpr (NPASS == 3) *reinterpret_cast<half4*>(dp + p.plane_stride) = lo;
                }
            }
            __builtin_amdgcn_sched_barrier(0);           // keep the next batch's loads from being hoisted (registers)
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int s = kc * 9 + t;
            const int kd = t / 3, kh = t - kd * 3;
            const int toff = (kd * p.HT + kh) * p.PW * 16;
            // 9 taps per chunk: with 3 sets the set index is t % 3; with 2 sets the parity of the global step
            // alternates per chunk, so the chunk loop body is instantiated for both parities (PAR)
            const int cur = NSET == 3 ? t % 3 : (PAR + t) & 1;
            uint4 bw[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) bw[f] = wq[cur][f];
            fetch(s + NSET - 1, wq[(cur + NSET - 1) % NSET]);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                half8 a[NPL];
#pragma unroll
                for (int hl = 0; hl < NPL; ++hl)
                    a[hl] = *reinterpret_cast<const half8*>(lds + a_off[mb] + hl * p.plane_stride + toff);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const half8 bhi = __builtin_bit_cast(half8, bw[nb * NPL]);
                    if constexpr (NPASS == 3) {
                        const half8 blo = __builtin_bit_cast(half8, bw[nb * NPL + 1]);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bhi, acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], blo, acc[mb][nb], 0, 0, 0);
                    }
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bhi, acc[mb][nb], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (NSET == 3) {
        for (int kc = 0; kc < p.KCN; ++kc) do_chunk(kc, std::integral_constant<int, 0>{});
    } else {
        for (int kc = 0; kc < p.KCN; kc += 2) {            // 9 steps per chunk: the parity flips every chunk
            do_chunk(kc, std::integral_constant<int, 0>{});
            if (kc + 1 < p.KCN) do_chunk(kc + 1, std::integral_constant<int, 1>{});
        }
    }

    // ================= epilogue: output transform through LDS =================
    float* m = reinterpret_cast<float*>(lds);                      // [4 positions][128 pairs][MLD]
    const int col = tid & 31, row0 = tid >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        __syncthreads();                                           // A planes (or the previous round) fully consumed
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = (i >> 2) * 8 + khalf * 4 + (i & 3);
                m[(pos * 128 + mb * 32 + row_perm(rr)) * MLD + l32] = acc[mb][nb][i];
            }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int q = row0 + 8 * it;
            int d, h, j;
            pair_coords(p, q, d, h, j);
            const int gz = z0 + d, gy = y0 + h, gx = x0 + 2 * j;
            if (gz >= p.D || gy >= p.H || gx >= p.W) continue;
            const float m0 = m[(0 * 128 + q) * MLD + col], m1 = m[(1 * 128 + q) * MLD + col];
            const float m2 = m[(2 * 128 + q) * MLD + col], m3 = m[(3 * 128 + q) * MLD + col];
            float* o = p.out + (((int64_t)gz * p.H + gy) * p.W + gx) * p.Cout + nt * 64 + nb * 32 + col;
            float y0v = ((m0 + m1) + m2) * dq;
            float y1v = ((m1 - m2) - m3) * dq;
            if (p.accum) y0v = y0v + o[0];
            y0v = y0v >= 0.f ? y0v : y0v * p.slope;
            o[0] = y0v;
            if (gx + 1 < p.W) {
                if (p.accum) y1v = y1v + o[p.Cout];
                y1v = y1v >= 0.f ? y1v : y1v * p.slope;
                o[p.Cout] = y1v;
            }
        }
    }
}

// packed[ntile64][pos 4][kc][(kd,kh) 9][nb 2][hl][lane] (uint4 = 8 halfs): lane l holds
// B[k = 8*(l>>5)+j][n = l&31] = U_pos[co = ntile*64 + nb*32 + (l&31)][ci = kc*16 + 8*(l>>5) + j][kd][kh] * 2^wexp,
// U = G g along kw: (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
__global__ void pack_wino(const float* __restrict__ w, int Cin, int Cout, int wexp, int npl, uint4* __restrict__ out) {
    const int KCN = Cin / KC;
    const int nf = 2 * npl;
    const int64_t n = (int64_t)(Cout / 64) * 4 * KCN * 9 * nf * 64;
    const float s = ldexpf(1.0f, wexp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int f = (int)(r % nf); r /= nf;
        const int t = (int)(r % 9); r /= 9;
        const int kc = (int)(r % KCN); r /= KCN;
        const int ps = (int)(r & 3); r >>= 2;
        const int ntile = (int)r;
        const int nb = f / npl, hl = f - nb * npl;
        const int co = ntile * 64 + nb * 32 + (lane & 31);
        const int ci0 = kc * KC + 8 * (lane >> 5);
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* g = w + ((size_t)co * Cin + ci0 + j) * 27 + t * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
            const float u = ps == 0 ? g0 : ps == 1 ? ((g0 + g1) + g2) * 0.5f : ps == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
            const float x = u * s;
            const _Float16 hh = (_Float16)x;
            v[j] = hl == 0 ? hh : (_Float16)(x - (float)hh);
        }
        out[i] = __builtin_bit_cast(uint4, v);
    }
}

int ilog2i(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }

bool choose_box(int D, int H, int W, int npl, int& TD, int& TH, int& TW) {
    static const int opts[][3] = {{4, 4, 16}, {4, 8, 8}, {8, 4, 8}, {8, 8, 4}, {2, 4, 32}, {4, 2, 32}, {2, 8, 16},
                                  {8, 2, 16}, {16, 4, 4}, {4, 16, 4}};
    int64_t best = -1;
    for (auto& o : opts) {
        const int pw = o[2] / 2;
        const int64_t npos = (int64_t)(o[0] + 2) * (o[1] + 2) * pw;
        if (npos * 4 > 5 * NTHR) continue;
        const int64_t plane = ((npos * 16 + 255) / 256) * 256 + 16;
        if (8 * npl * plane > 80 * 1024) continue;                 // two workgroups per CU
        int64_t cost = (int64_t)bfm_cdiv(D, o[0]) * bfm_cdiv(H, o[1]) * bfm_cdiv(W, o[2]);
        cost = cost * 64 - o[2];
        if (best < 0 || cost < best) { best = cost; TD = o[0]; TH = o[1]; TW = o[2]; }
    }
    return best >= 0;
}

}  // namespace

extern "C" size_t bfm_pack_conv_weights_wino_bytes(int Cin, int Cout, int passes) {
    if (Cin <= 0 || Cout <= 0 || Cin % KC || Cout % 64) return 0;
    const int npl = passes == 3 ? 2 : 1;
    return (size_t)(Cout / 64) * 4 * (Cin / KC) * 9 * 2 * npl * 64 * sizeof(uint4);
}

extern "C" int bfm_pack_conv_weights_wino(const float* w_oidhw, int Cin, int Cout, float wmax_abs_host, int passes,
                                          void* wpacked, int* wexp_host, bfm_stream_t stream) {
    if (!w_oidhw || !wpacked || !wexp_host || Cin <= 0 || Cout <= 0) return BFM_E_ARG;
    if (Cin % KC || Cout % 64 || (passes != 1 && passes != 3)) return BFM_E_SHAPE;
    int wexp = 0;
    if (wmax_abs_host > 0.f && wmax_abs_host < INFINITY) {
        int ex;
        (void)frexpf(2.f * wmax_abs_host, &ex);                  // |U| <= 1.5 max|g|
        wexp = 14 - ex;
        wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp);
    }
    *wexp_host = wexp;
    const int npl = passes == 3 ? 2 : 1;
    const int64_t n = (int64_t)(Cout / 64) * 4 * (Cin / KC) * 9 * 2 * npl * 64;
    int nb = (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256));
    hipLaunchKernelGGL(pack_wino, dim3(nb), dim3(256), 0, bfm_s(stream), w_oidhw, Cin, Cout, wexp, npl,
                       static_cast<uint4*>(wpacked));
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_wino(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                  const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                  int passes, int accumulate, float* out, bfm_stream_t stream) {
    if (!A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift || !bound || G <= 0 || !wpacked || !out)
        return BFM_E_ARG;
    if (CA % KC || Cout % 64 || Cout <= 0) return BFM_E_SHAPE;
    if (passes != 1 && passes != 3) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(scale) & 15) ||
        (reinterpret_cast<uintptr_t>(shift) & 15) || (reinterpret_cast<uintptr_t>(wpacked) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return BFM_E_ARG;
    if ((int64_t)D * H * W * CA > 0x7fffffffLL) return BFM_E_SHAPE;       // 32-bit staging offsets
    const int npl = passes == 3 ? 2 : 1;
    WinoParams p{};
    p.A = A; p.CA = CA; p.D = D; p.H = H; p.W = W;
    p.scale = scale; p.shift = shift; p.bound = bound; p.G = G;
    p.wp = static_cast<const uint4*>(wpacked);
    p.wexp = wexp; p.Cout = Cout; p.slope = slope; p.out = out; p.accum = accumulate ? 1 : 0;
    if (!choose_box(D, H, W, npl, p.TD, p.TH, p.TW)) return BFM_E_SHAPE;
    p.HT = p.TH + 2; p.PW = p.TW / 2;
    p.pw_shift = ilog2i(p.PW); p.thp_shift = ilog2i(p.TH * p.PW);
    const int nTz = bfm_cdiv(D, p.TD);
    p.nTy = bfm_cdiv(H, p.TH); p.nTx = bfm_cdiv(W, p.TW);
    p.nMt = nTz * p.nTy * p.nTx;
    p.NT = Cout / 64;
    p.KCN = CA / KC;
    p.npos_lds = (p.TD + 2) * p.HT * p.PW;
    p.plane_stride = ((p.npos_lds * 16 + 255) / 256) * 256 + 16;
    size_t smem = (size_t)8 * npl * p.plane_stride;
    const size_t epi = (size_t)4 * 128 * MLD * sizeof(float);
    if (smem < epi) smem = epi;
    if (smem > 80 * 1024) return BFM_E_SHAPE;
    if ((int64_t)p.nMt * p.NT > 0x7fffffff) return BFM_E_SHAPE;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024);
        attr_done = true;
    }
    dim3 grid((unsigned)(p.nMt * p.NT));
    if (passes == 3) hipLaunchKernelGGL(conv_wino<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    else hipLaunchKernelGGL(conv_wino<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    return bfm_launch_status();
}
