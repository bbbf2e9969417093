// 3x3x3 convolution as an implicit GEMM on CDNA4 matrix cores (gfx950).
//
//   out[v][co] = LeakyReLU( sum_{tap,ci} (x[v+tap][ci]*scale[ci]+shift[ci]) * w[co][ci][tap] )
//
// i.e. the body of SingleConv 'gcl' (Trainer/models/unet3d/buildingblocks.py:31-60)
// with the GroupNorm affine folded into the operand load, zero padding applied
// after the affine, and -- for decoder inputs -- the nearest-upsample + concat
// of Decoder._joining (buildingblocks.py:265-276) folded into the load as a
// second source pointer (never materialised).
//
// GEMM view: M = output voxels (a TDxTHxTW box per workgroup), N = Cout,
// K = 27 taps x Cin.  Per 16-channel K-chunk the workgroup stages the box's
// halo ((TD+2)(TH+2)(TW+2) voxels x 16 ch) into LDS once -- affine applied,
// split into fp16 hi + lo planes -- and re-reads it for all 27 taps
// (the im2col happens at ds_read time).  Weights are pre-packed in exact MFMA
// fragment order and stream L2 -> VGPR with one 16-byte load per lane.
//
// Numerics: fp32 operands are split x = hi + lo (both fp16 after a power-of-two
// pre-scale chosen from the GroupNorm bound / max|w| so that nothing can
// overflow fp16) and the product is hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3 MFMAs at 16x the fp32
// matrix rate each => 5.3x faster than v_mfma_f32_32x32x2_f32 at ~2^-22
// relative product error (passes=3).  passes=1 keeps only hi*hi.
//
// Wave tile 64x64 (2x2 MFMA blocks), workgroup = WM x WN waves.
#include "bfm_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;            // channels per K-chunk
#ifdef BFM_STAMPS
__device__ long long g_stamps[4096];
#define STAMP(i) do { if (stamp_on) g_stamps[stamp_base + (i)] = clock64(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
constexpr int FRAG_U4 = 64;       // one fragment = 64 lanes x uint4

struct ConvParams {
    const float *A, *B;
    int CA, CB, D, H, W;
    UpView up;
    const float *scale, *shift, *bound;
    int G;
    const uint4* wp;
    int wexp;
    int Cout;
    float slope;
    float* out;
    int TD, TH, TW, HT, WT;        // box and halo'd box
    int nTy, nTx, nMt, NT;          // tiles
    int KCN, kc_per_split, splitk;
    int nvox_lds, plane_stride;     // LDS geometry (plane_stride in bytes)
    int tw_shift, thw_shift;        // >=0 when TW / TH*TW are powers of two
    int64_t split_stride;           // elements between split-K slabs
    int accum;                      // 1: add what `out` already holds (the up-folded low-res half) before LeakyReLU
    // S > 1: a batch of S same-shape samples (tiles of one volume) in one launch.  A, B, out are [S][...] contiguous
    // (element strides sA, sB, sO), scale / shift [S][CA+CB], bound [S][G] (GroupNorm statistics stay per sample,
    // buildingblocks.py:48-60); nMt = S * nMtS M tiles, sample fastest in the block order so that the S workgroups
    // that need the same weight fragments run side by side and share them in L2.  Every workgroup does exactly what it
    // does in the per-sample launch (same box, same K order, same split-K): results are bit-identical.
    int S, nMtS;
    int saff;                        // elements between two samples' scale / shift rows (CA + CB unless the caller's tables are wider)
    int64_t sA, sB, sO;
    // optional moment rows of the OUTPUT (one row per M tile, [nMt][Cout]): {sum, sumsq} fp64, {min, max} fp32.  The
    // consumer's GroupNorm reduces these instead of re-reading the activation (gn_stats.hip: bfm_gn_stats_rows).
    double *rsum, *rsq;
    float *rmn, *rmx;
    int abl;                         // diagnostics builds (-DBFM_MFMA_ABLATE): 32 = no K loop, 16 = no epilogue
};


// Cross-wave fold of per-wave column moments and the row store.  Called by every thread of the workgroup after
// the K loop; `lds` is free by then (a barrier separates the last MFMA reads from these writes).
// s/q/mn/mx: this lane's totals for column `col` (0..63 of the wave's 64-column tile), valid when `writer`.
template <int WM, int WN>
__device__ __forceinline__ void store_moment_row(const ConvParams& p, unsigned char* lds, int tid, int wm, int wn,
                                                 bool writer, const int (&col)[4], int ncol, const double (&s)[4],
                                                 const double (&q)[4], const float (&mn)[4], const float (&mx)[4],
                                                 int mt, int nt) {
    double* ls = reinterpret_cast<double*>(lds);                 // [WM*WN][64]
    double* lq = ls + WM * WN * 64;
    float* lmn = reinterpret_cast<float*>(lq + WM * WN * 64);
    float* lmx = lmn + WM * WN * 64;
    __syncthreads();
    if (writer) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < ncol) {
                const int i = (wm * WN + wn) * 64 + col[j];
                ls[i] = s[j]; lq[i] = q[j]; lmn[i] = mn[j]; lmx[i] = mx[j];
            }
        }
    }
    __syncthreads();
    if (tid < 64 * WN) {
        const int w2 = tid >> 6, c = tid & 63;
        double S = 0.0, Q = 0.0;
        float MN = INFINITY, MX = -INFINITY;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int i = (m * WN + w2) * 64 + c;
            S += ls[i]; Q += lq[i];
            MN = fminf(MN, lmn[i]); MX = fmaxf(MX, lmx[i]);
        }
        const size_t o = (size_t)mt * p.Cout + (nt * WN + w2) * 64 + c;
        p.rsum[o] = S; p.rsq[o] = Q; p.rmn[o] = MN; p.rmx[o] = MX;
    }
}

// lane (= MFMA row) -> position inside the 32-row block such that every
// ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} reads 16
// consecutive box positions (one w-run when TW == 16): conflict-free.
__device__ __forceinline__ int row_perm(int l) {
    if (l < 4) return l;
    if (l < 12) return l + 12;
    if (l < 16) return l - 8;
    if (l < 20) return l + 8;
    if (l < 28) return l - 12;
    return l;
}

__device__ __forceinline__ void box_coords(const ConvParams& p, int q, int& d, int& h, int& w) {
    if (p.tw_shift >= 0 && p.thw_shift >= 0) {
        d = q >> p.thw_shift;
        int rem = q & ((1 << p.thw_shift) - 1);
        h = rem >> p.tw_shift;
        w = rem & ((1 << p.tw_shift) - 1);
    } else {
        int thw = p.TH * p.TW;
        d = q / thw;
        int rem = q - d * thw;
        h = rem / p.TW;
        w = rem - h * p.TW;
    }
}

// The element offset of every row's output voxel relative to the box's first voxel (-1: a padding row or a voxel outside the
// tensor), one entry per row in LDS at offset 0, written by one thread per row.  Round 5: the accumulate-mode preload and the
// epilogue worked the row -> (d,h,w) -> address chain out per STORED ROW AND THREAD (32 rows x two integer divisions each);
// on the split-K layers of the deep levels, whose workgroups run four K chunks, that epilogue was 10-20 % of the launch
// (tests/diag/diag_mfma_overheads.py).  The caller puts a barrier before (LDS free) and after (table visible).
template <int ROWS, int NTHR>
__device__ __forceinline__ const int* build_row_table(const ConvParams& p, unsigned char* lds, int tid, int z0, int y0, int x0,
                                                      int boxN) {
    int* tab = reinterpret_cast<int*>(lds);
    for (int q = tid; q < ROWS; q += NTHR) {
        int rel = -1;
        if (q < boxN) {
            int d, h, w;
            box_coords(p, q, d, h, w);
            if (z0 + d < p.D && y0 + h < p.H && x0 + w < p.W) rel = ((d * p.H + h) * p.W + w) * p.Cout;
        }
        tab[q] = rel;
    }
    return tab;
}

template <int WM, int WN, int NPASS, int NSLOT>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_mfma(const ConvParams p) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int NW = WM * WN;                    // waves per workgroup
    constexpr int NPL = (NPASS == 3) ? 2 : 1;      // planes per k-half (hi[, lo])
    constexpr int ROWFR = 12 * WN;                 // 1-KiB weight fragments per (kd,kh) row: 3 taps x 2 nb x 2 hl x WN
    constexpr int KPR = ROWFR / NW;                // LDS-DMA instructions per wave per row
    constexpr int DPF = NSLOT - 1;                 // rows the weight stream runs ahead
    static_assert(ROWFR % NW == 0, "row fragments must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l32 = lane & 31, khalf = lane >> 5;

    // ---- block -> (M tile, N tile): XCD-aware bijective remap so that the 8
    // round-robin XCDs each own a contiguous run of tiles (neighbouring boxes
    // share halo lines and all N tiles of a box share its input in one L2).
    int bid = blockIdx.x;
    {
        const int nblk = p.nMt * p.NT;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int mt, nt, smp = 0;
    if (p.S > 1) {                                       // (M tile of a sample, N tile, sample): sample fastest
        smp = bid % p.S;
        const int t = bid / p.S;
        nt = t % p.NT;
        mt = smp * p.nMtS + t / p.NT;
    } else {
        mt = bid / p.NT;
        nt = bid - mt * p.NT;
    }
    const int mtl = mt - smp * p.nMtS;                   // tile index inside its sample
    const float* const pA = p.A + smp * p.sA;
    const float* const pB = p.B + smp * p.sB;
    const float* const pscale = p.scale + smp * p.saff;
    const float* const pshift = p.shift + smp * p.saff;
    const float* const pbound = p.bound + smp * p.G;
    float* const pout = p.out + smp * p.sO;
    const int split = blockIdx.y;
#ifdef BFM_STAMPS
    const bool stamp_on = tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3) && blockIdx.y == gridDim.y / 2;
    int stamp_base = blockIdx.x == 0 ? 0 : 1024;
    long long t_wait = 0, t_bar = 0;
#endif
    const int tx = mtl % p.nTx;
    const int ty = (mtl / p.nTx) % p.nTy;
    const int tz = mtl / (p.nTx * p.nTy);
    const int z0 = tz * p.TD, y0 = ty * p.TH, x0 = tx * p.TW;
    const int boxN = p.TD * p.TH * p.TW;

    // ---- operand scale from the GroupNorm bound (power of two, exact)
    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, pbound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);       // bmax = m * 2^ex, m in [0.5,1)
        aexp = 14 - ex;
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    // ---- per-lane A fragment base offsets (bytes) for the two 32-row blocks
    int a_off[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        int q = wm * 64 + mb * 32 + row_perm(l32);
        int d = 0, h = 0, w = 0;
        if (q < boxN) box_coords(p, q, d, h, w);
        int vox = (d * p.HT + h) * p.WT + w;
        a_off[mb] = (khalf * NPL) * p.plane_stride + vox * 16;
    }

    // ---- staging bookkeeping: element e = tid + it*NTHR -> (halo voxel, channel quad).
    // off[it] = element offset of the voxel's channel 0 in the current source tensor
    // (-1: outside the volume -> staged as zero, -2: no such element), recomputed only when the
    // source switches from the skip tensor to the upsampled low-res tensor.
    constexpr int MAX_IT = 12;
    const int n_el = p.nvox_lds * 4;
    const int q4 = tid & 3;                              // NTHR % 4 == 0 -> fixed per thread
    int off[MAX_IT];
    auto compute_offsets = [&](bool fromB) {
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            const int e = tid + it * NTHR;
            off[it] = -2;
            if (e < n_el) {
                const int vox = e >> 2;
                const int hz = vox / (p.HT * p.WT);
                const int rem = vox - hz * (p.HT * p.WT);
                const int hy = rem / p.WT;
                const int hx = rem - hy * p.WT;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                off[it] = -1;
                if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    if (fromB) off[it] = ((p.up.mapD[gz] * p.up.h + p.up.mapH[gy]) * p.up.w + p.up.mapW[gx]) * p.CB;
                    else off[it] = ((gz * p.H + gy) * p.W + gx) * p.CA;
                }
            }
        }
    };
    bool off_fromB = (split * p.kc_per_split * KC) >= p.CA;
    compute_offsets(off_fromB);
    const int st_plane = ((q4 >> 1) * NPL) * p.plane_stride + (q4 & 1) * 8;

    floatx16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    const int ntw = nt * WN + wn;                        // 64-column tile of this wave
    if (p.accum && p.splitk == 1) {
        // accumulate mode: start from what `out` holds (the up-folded half), in the accumulator's scaled domain; the
        // loads are issued here so that their latency hides under the K loop instead of the epilogue
        const float inv_dq = ldexpf(1.0f, aexp + p.wexp);
        const int* rowtab = build_row_table<WM * 64, NTHR>(p, lds, tid, z0, y0, x0, boxN);    // LDS is not in use yet
        __syncthreads();
        const float* ib = pout + (((int64_t)z0 * p.H + y0) * p.W + x0) * p.Cout + ntw * 64 + l32;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int ig = 0; ig < 4; ++ig) {
                const int4 r4 = *reinterpret_cast<const int4*>(rowtab + wm * 64 + mb * 32 + row_perm(ig * 8 + khalf * 4));
                const int rel[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (rel[j] < 0) continue;
                    const float* orow = ib + (unsigned)rel[j];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[mb][nb][ig * 4 + j] = orow[nb * 32] * inv_dq;
                }
            }
        }
        // (the first chunk's barrier comes before anything is staged over the table)
    }
    const int kc_begin = split * p.kc_per_split;
    int kc_end = min(p.KCN, kc_begin + p.kc_per_split);
#ifdef BFM_MFMA_ABLATE
    if (p.abl & 32) kc_end = kc_begin;
#endif

    // ---- weight stream: row R = (chunk, kd, kh); fragment f of a row -> LDS slot (R % NSLOT)
    const int b_base = 2 * NPL * p.plane_stride;
    const int total_rows = (kc_end - kc_begin) * 9;
    auto issue_row = [&](int R) {
        const int Rc = R < total_rows ? R : total_rows - 1;       // past the end: harmless reload, never read
        const int kcR = kc_begin + Rc / 9, rr = Rc % 9;
        const int slot = R % NSLOT;
#pragma unroll
        for (int i = 0; i < KPR; ++i) {
            const int f = wave + i * NW;
            const int j = f / 12, g = f - j * 12;
            const uint4* src = p.wp + ((size_t)((nt * WN + j) * p.KCN + kcR) * 27 + rr * 3) * (4 * FRAG_U4) +
                               g * FRAG_U4 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + b_base +
                                                                                      (slot * ROWFR + f) * 1024),
                                             16, 0, 0);
        }
    };
    if (total_rows > 0) {
#pragma unroll
        for (int R = 0; R < DPF; ++R) issue_row(R);
    }

    for (int kc = kc_begin; kc < kc_end; ++kc) {
        // ================= stage chunk kc =================
        const int c0 = kc * KC;
        const bool fromB = c0 >= p.CA;
        if (fromB != off_fromB) { off_fromB = fromB; compute_offsets(fromB); }
        const float* src = (fromB ? pB + (c0 - p.CA) : pA + c0) + q4 * 4;
        const float4 sc4 = *reinterpret_cast<const float4*>(pscale + c0 + q4 * 4);
        const float4 sh4 = *reinterpret_cast<const float4*>(pshift + c0 + q4 * 4);
        const float sc[4] = {sc4.x * a_scale, sc4.y * a_scale, sc4.z * a_scale, sc4.w * a_scale};
        const float sh[4] = {sh4.x * a_scale, sh4.y * a_scale, sh4.z * a_scale, sh4.w * a_scale};

        STAMP((kc - kc_begin) * 8 + 0);
        __syncthreads();                                 // previous chunk's readers are done
        STAMP((kc - kc_begin) * 8 + 1);
#pragma unroll
        for (int it0 = 0; it0 < MAX_IT; it0 += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (off[it0 + u] >= 0) v[u] = *reinterpret_cast<const float4*>(src + off[it0 + u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + (it0 + u) * NTHR;
                if (off[it0 + u] != -2) {
                    const bool ok = off[it0 + u] >= 0;
                    float y[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    half4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = ok ? fmaf(y[i], sc[i], sh[i]) : 0.f;      // zero padding AFTER the affine
                        _Float16 hh = (_Float16)t;
                        hi[i] = hh;
                        lo[i] = (_Float16)(t - (float)hh);
                    }
                    unsigned char* dst = lds + st_plane + (e >> 2) * 16;
                    *reinterpret_cast<half4*>(dst) = hi;
                    if constexpr (NPASS == 3) *reinterpret_cast<half4*>(dst + p.plane_stride) = lo;
                }
            }
        }
        STAMP((kc - kc_begin) * 8 + 2);
        __syncthreads();
        STAMP((kc - kc_begin) * 8 + 3);

        // ================= 27 taps of MFMA on the staged chunk =================
        // Weights stream through an LDS ring, one (kd,kh) row = 3 taps per slot, filled by LDS-DMA
        // (global_load_lds_dwordx4: one packed 1-KiB fragment per wave-instruction) DPF rows ahead.
        // Per row: own DMAs of this row landed (counted vmcnt) -> barrier (everyone's landed, and the
        // slot consumed one row ago is free) -> refill that slot -> 36 MFMAs.
        const int rbase = (kc - kc_begin) * 9;
        for (int r = 0; r < 9; ++r) {
            const int R = rbase + r;
#ifdef BFM_STAMPS
            long long ta = 0, tb = 0, tc = 0;
            if (stamp_on) ta = clock64();
#endif
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPR * (DPF - 1)) : "memory");
#ifdef BFM_STAMPS
            if (stamp_on) tb = clock64();
#endif
            __builtin_amdgcn_s_barrier();
#ifdef BFM_STAMPS
            if (stamp_on) { tc = clock64(); t_wait += tb - ta; t_bar += tc - tb; }
#endif
            issue_row(R + DPF);
            const unsigned char* bs = lds + b_base + ((R % NSLOT) * ROWFR + wn * 12) * 1024 + lane * 16;
            const int kd = r / 3, kh = r - kd * 3;
            const int roff = (kd * p.HT + kh) * p.WT * 16;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                half8 a[2][NPL], b[2][NPL];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl)
                        b[nb][hl] = *reinterpret_cast<const half8*>(bs + (kw * 4 + nb * 2 + hl) * 1024);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl)
                        a[mb][hl] = *reinterpret_cast<const half8*>(lds + a_off[mb] + hl * p.plane_stride + roff +
                                                                    kw * 16);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        if constexpr (NPASS == 3) {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb][1], b[nb][0], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb][0], b[nb][1], acc[mb][nb], 0, 0, 0);
                        }
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb][0], b[nb][0], acc[mb][nb], 0, 0, 0);
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the run-ahead DMAs before the LDS is released
#ifdef BFM_STAMPS
    if (stamp_on) { g_stamps[stamp_base + 1000] = t_wait; g_stamps[stamp_base + 1001] = t_bar; g_stamps[stamp_base + 1002] = clock64();
                    g_stamps[stamp_base + 1003] = kc_end - kc_begin; }
#endif

    // ================= epilogue =================
#ifdef BFM_MFMA_ABLATE
    if (p.abl & 16) {
        float sm = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sm += acc[mb][0][i] + acc[mb][1][i];
        if (sm == 12345.678f) pout[tid] = sm;
        return;
    }
#endif
    const bool final_out = p.splitk == 1;
    const bool want_rows = final_out && p.rsum != nullptr;
    float* obase = pout + (final_out ? 0 : (int64_t)split * p.split_stride);
    // per-lane partial moments of <= 32 stored values per column in fp32 (then fp64 across lanes, waves and tiles)
    float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
    float mmn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, mmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    __syncthreads();                                           // every wave is done with the planes and the ring
    const int* rowtab = build_row_table<WM * 64, NTHR>(p, lds, tid, z0, y0, x0, boxN);
    __syncthreads();
    float* const ob = obase + (((int64_t)z0 * p.H + y0) * p.W + x0) * p.Cout + ntw * 64 + l32;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int ig = 0; ig < 4; ++ig) {
            // rows (ig, khalf, j = 0..3) are four consecutive entries (row_perm keeps aligned groups of four together)
            const int4 r4 = *reinterpret_cast<const int4*>(rowtab + wm * 64 + mb * 32 + row_perm(ig * 8 + khalf * 4));
            const int rel[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (rel[j] < 0) continue;
                float* orow = ob + (unsigned)rel[j];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    float r = acc[mb][nb][ig * 4 + j] * dq;
                    if (final_out) r = r >= 0.f ? r : r * p.slope;     // accumulate mode preloaded `out` into acc
                    orow[nb * 32] = r;
                    if (want_rows) {
                        fs[nb] += r; fq[nb] = fmaf(r, r, fq[nb]);
                        mmn[nb] = fminf(mmn[nb], r); mmx[nb] = fmaxf(mmx[nb], r);
                    }
                }
            }
        }
    }
    if (want_rows) {                                         // uniform per workgroup
        double ms[4], mq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ms[j] = (double)fs[j]; mq[j] = (double)fq[j]; }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {                     // the other k-half holds the other rows of the same column
            ms[nb] += __shfl_xor(ms[nb], 32);
            mq[nb] += __shfl_xor(mq[nb], 32);
            mmn[nb] = fminf(mmn[nb], __shfl_xor(mmn[nb], 32));
            mmx[nb] = fmaxf(mmx[nb], __shfl_xor(mmx[nb], 32));
        }
        const int cols[4] = {l32, 32 + l32, 0, 0};
        store_moment_row<WM, WN>(p, lds, tid, wm, wn, khalf == 0, cols, 2, ms, mq, mmn, mmx, mt, nt);
    }
}

// ---------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_f16 variant of the 4-wave kernel.  Under a sustained matrix load MI355X is
// power-limited and holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per
// FLOP (MI355X_MICROARCH.md, DVFS give-back item 7), so the same MFMA work finishes sooner.
// K = 32 is two taps x 16 channels: lane l holds row/col l&15 and k-group kg = l>>4, where kg&1 is the
// 8-channel half (LDS plane) and kg>>1 selects the first or second tap of the pair.  27 taps = 14
// pairs (the 28th tap has zero weights).  Wave tile 64x64 = 4x4 blocks of 16x16.
template <int WM, int WN, int NPASS, int SP, int NSLOT>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_mfma16(const ConvParams p) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int NW = WM * WN;
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int PAIRFR = 8 * WN;                 // 1-KiB fragments per tap pair: 4 col-blocks x (hi,lo) x WN
    constexpr int SLOTFR = SP * PAIRFR;            // fragments per ring slot
    constexpr int KPS = SLOTFR / NW;               // LDS-DMA instructions per wave per slot
    constexpr int NSL = 14 / SP;                   // slots per K-chunk
    constexpr int DPF = NSLOT - 1;
    static_assert(SLOTFR % NW == 0 && 14 % SP == 0, "slot geometry");
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l16 = lane & 15, kg = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nblk = p.nMt * p.NT;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int mt, nt, smp = 0;
    if (p.S > 1) {                                       // (M tile of a sample, N tile, sample): sample fastest
        smp = bid % p.S;
        const int t = bid / p.S;
        nt = t % p.NT;
        mt = smp * p.nMtS + t / p.NT;
    } else {
        mt = bid / p.NT;
        nt = bid - mt * p.NT;
    }
    const int mtl = mt - smp * p.nMtS;                   // tile index inside its sample
    const float* const pA = p.A + smp * p.sA;
    const float* const pB = p.B + smp * p.sB;
    const float* const pscale = p.scale + smp * p.saff;
    const float* const pshift = p.shift + smp * p.saff;
    const float* const pbound = p.bound + smp * p.G;
    float* const pout = p.out + smp * p.sO;
    const int split = blockIdx.y;
    const int tx = mtl % p.nTx;
    const int ty = (mtl / p.nTx) % p.nTy;
    const int tz = mtl / (p.nTx * p.nTy);
    const int z0 = tz * p.TD, y0 = ty * p.TH, x0 = tx * p.TW;
    const int boxN = p.TD * p.TH * p.TW;

    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, pbound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        aexp = 14 - ex;
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    // per-lane A base offsets of the four 16-row blocks (rows = consecutive box positions: one w-run at TW=16)
    int a_off[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        int q = wm * 64 + rb * 16 + l16;
        int d = 0, h = 0, w = 0;
        if (q < boxN) box_coords(p, q, d, h, w);
        int vox = (d * p.HT + h) * p.WT + w;
        a_off[rb] = ((kg & 1) * NPL) * p.plane_stride + vox * 16;
    }

    constexpr int MAX_IT = 12;
    const int n_el = p.nvox_lds * 4;
    const int q4 = tid & 3;
    int off[MAX_IT];
    auto compute_offsets = [&](bool fromB) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            const int e = tid + it * NTHR;
            off[it] = -2;
            if (e < n_el) {
                const int vox = e >> 2;
                const int hz = vox / (p.HT * p.WT);
                const int rem = vox - hz * (p.HT * p.WT);
                const int hy = rem / p.WT;
                const int hx = rem - hy * p.WT;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                off[it] = -1;
                if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    if (fromB) off[it] = ((p.up.mapD[gz] * p.up.h + p.up.mapH[gy]) * p.up.w + p.up.mapW[gx]) * p.CB;
                    else off[it] = ((gz * p.H + gy) * p.W + gx) * p.CA;
                }
            }
        }
    };
    bool off_fromB = (split * p.kc_per_split * KC) >= p.CA;
    compute_offsets(off_fromB);
    const int st_plane = ((q4 >> 1) * NPL) * p.plane_stride + (q4 & 1) * 8;

    floatx4 acc[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[rb][cb][i] = 0.f;

    const int ntw = nt * WN + wn;
    if (p.accum && p.splitk == 1) {                      // accumulate mode: see conv_mfma
        const float inv_dq = ldexpf(1.0f, aexp + p.wexp);
        const int* rowtab = build_row_table<WM * 64, NTHR>(p, lds, tid, z0, y0, x0, boxN);    // LDS is not in use yet
        __syncthreads();
        const float* ib = pout + (((int64_t)z0 * p.H + y0) * p.W + x0) * p.Cout + ntw * 64 + l16;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int4 r4 = *reinterpret_cast<const int4*>(rowtab + wm * 64 + rb * 16 + kg * 4);
            const int rel[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (rel[i] < 0) continue;
                const float* orow = ib + (unsigned)rel[i];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[rb][cb][i] = orow[cb * 16] * inv_dq;
            }
        }
    }
    const int kc_begin = split * p.kc_per_split;
    int kc_end = min(p.KCN, kc_begin + p.kc_per_split);
#ifdef BFM_MFMA_ABLATE
    if (p.abl & 32) kc_end = kc_begin;
#endif
    const int b_base = 2 * NPL * p.plane_stride;
    const int total_slots = (kc_end - kc_begin) * NSL;
    auto issue_slot = [&](int S) __attribute__((always_inline)) {
        const int Sc = S < total_slots ? S : total_slots - 1;
        const int kcS = kc_begin + Sc / NSL, sl = Sc % NSL;
        const int slot = S % NSLOT;
#pragma unroll
        for (int i = 0; i < KPS; ++i) {
            const int f = wave + i * NW;
            const int j = f / (SP * 8), rem = f - j * (SP * 8);
            const int pr = rem >> 3, g = rem & 7;
            const uint4* src = p.wp + ((size_t)((nt * WN + j) * p.KCN + kcS) * 14 + sl * SP + pr) * (8 * FRAG_U4) +
                               g * FRAG_U4 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + b_base +
                                                                                      (slot * SLOTFR + f) * 1024),
                                             16, 0, 0);
        }
    };
    if (total_slots > 0) {
#pragma unroll
        for (int S = 0; S < DPF; ++S) issue_slot(S);
    }

    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int c0 = kc * KC;
        const bool fromB = c0 >= p.CA;
        if (fromB != off_fromB) { off_fromB = fromB; compute_offsets(fromB); }
        const float* src = (fromB ? pB + (c0 - p.CA) : pA + c0) + q4 * 4;
        const float4 sc4 = *reinterpret_cast<const float4*>(pscale + c0 + q4 * 4);
        const float4 sh4 = *reinterpret_cast<const float4*>(pshift + c0 + q4 * 4);
        const float sc[4] = {sc4.x * a_scale, sc4.y * a_scale, sc4.z * a_scale, sc4.w * a_scale};
        const float sh[4] = {sh4.x * a_scale, sh4.y * a_scale, sh4.z * a_scale, sh4.w * a_scale};

        __syncthreads();
#pragma unroll
        for (int it0 = 0; it0 < MAX_IT; it0 += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (off[it0 + u] >= 0) v[u] = *reinterpret_cast<const float4*>(src + off[it0 + u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + (it0 + u) * NTHR;
                if (off[it0 + u] != -2) {
                    const bool ok = off[it0 + u] >= 0;
                    float y[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    half4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = ok ? fmaf(y[i], sc[i], sh[i]) : 0.f;
                        _Float16 hh = (_Float16)t;
                        hi[i] = hh;
                        lo[i] = (_Float16)(t - (float)hh);
                    }
                    unsigned char* dst = lds + st_plane + (e >> 2) * 16;
                    *reinterpret_cast<half4*>(dst) = hi;
                    if constexpr (NPASS == 3) *reinterpret_cast<half4*>(dst + p.plane_stride) = lo;
                }
            }
        }
        __syncthreads();

        const int sbase = (kc - kc_begin) * NSL;
        for (int sl = 0; sl < NSL; ++sl) {
            const int S = sbase + sl;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPS * (DPF - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
            issue_slot(S + DPF);
#pragma unroll
            for (int pr = 0; pr < SP; ++pr) {
                const int P = sl * SP + pr;
                const int t0 = 2 * P, t1 = min(2 * P + 1, 26);
                const int kd0 = t0 / 9, r0 = t0 - kd0 * 9, kh0 = r0 / 3, kw0 = r0 - kh0 * 3;
                const int kd1 = t1 / 9, r1 = t1 - kd1 * 9, kh1 = r1 / 3, kw1 = r1 - kh1 * 3;
                const int toff0 = ((kd0 * p.HT + kh0) * p.WT + kw0) * 16;
                const int toff1 = ((kd1 * p.HT + kh1) * p.WT + kw1) * 16;
                const int toff = (kg >> 1) ? toff1 : toff0;
                const unsigned char* bs = lds + b_base + ((S % NSLOT) * SLOTFR + (wn * SP + pr) * 8) * 1024 + lane * 16;
                half8 a[4][NPL], b[4][NPL];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl)
                        b[cb][hl] = *reinterpret_cast<const half8*>(bs + (cb * 2 + hl) * 1024);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl)
                        a[rb][hl] = *reinterpret_cast<const half8*>(lds + a_off[rb] + hl * p.plane_stride + toff);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        if constexpr (NPASS == 3) {
                            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rb][1], b[cb][0], acc[rb][cb], 0, 0, 0);
                            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rb][0], b[cb][1], acc[rb][cb], 0, 0, 0);
                        }
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rb][0], b[cb][0], acc[rb][cb], 0, 0, 0);
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#ifdef BFM_MFMA_ABLATE
    if (p.abl & 16) {
        float sm = 0.f;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) sm += acc[rb][cb][0] + acc[rb][cb][1] + acc[rb][cb][2] + acc[rb][cb][3];
        if (sm == 12345.678f) pout[tid] = sm;
        return;
    }
#endif
    const bool final_out = p.splitk == 1;
    const bool want_rows = final_out && p.rsum != nullptr;
    float* obase = pout + (final_out ? 0 : (int64_t)split * p.split_stride);
    float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};      // <= 16 values per column per lane in fp32
    float mmn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, mmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    __syncthreads();                                           // every wave is done with the planes and the ring
    const int* rowtab = build_row_table<WM * 64, NTHR>(p, lds, tid, z0, y0, x0, boxN);
    __syncthreads();
    float* const ob = obase + (((int64_t)z0 * p.H + y0) * p.W + x0) * p.Cout + ntw * 64 + l16;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        // C/D: row = (lane>>4)*4 + reg, col = lane&15: this lane's four rows of the block are consecutive table entries
        const int4 r4 = *reinterpret_cast<const int4*>(rowtab + wm * 64 + rb * 16 + kg * 4);
        const int rel[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (rel[i] < 0) continue;
            float* orow = ob + (unsigned)rel[i];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                float r = acc[rb][cb][i] * dq;
                if (final_out) r = r >= 0.f ? r : r * p.slope;     // accumulate mode preloaded `out` into acc
                orow[cb * 16] = r;
                if (want_rows) {
                    fs[cb] += r; fq[cb] = fmaf(r, r, fq[cb]);
                    mmn[cb] = fminf(mmn[cb], r); mmx[cb] = fmaxf(mmx[cb], r);
                }
            }
        }
    }
    if (want_rows) {
        double ms[4], mq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ms[j] = (double)fs[j]; mq[j] = (double)fq[j]; }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {                     // the four k-groups hold different rows of the same column
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                ms[cb] += __shfl_xor(ms[cb], m);
                mq[cb] += __shfl_xor(mq[cb], m);
                mmn[cb] = fminf(mmn[cb], __shfl_xor(mmn[cb], m));
                mmx[cb] = fmaxf(mmx[cb], __shfl_xor(mmx[cb], m));
            }
        }
        const int cols[4] = {l16, 16 + l16, 32 + l16, 48 + l16};
        store_moment_row<WM, WN>(p, lds, tid, wm, wn, kg == 0, cols, 4, ms, mq, mmn, mmx, mt, nt);
    }
}

// packed16[ntile64][kc][pair 14][cb 4][hl][lane] (uint4 = 8 halfs): lane l, kg = l>>4 holds
// B[k = 8*kg + j][n = l&15] = w[co = ntile*64 + cb*16 + (l&15)][ci = kc*16 + 8*(kg&1) + j][tap = 2*pair + (kg>>1)] * 2^wexp
__global__ void pack_mfma16(const float* __restrict__ w, int Cin, int Cout, int wexp, uint4* __restrict__ out) {
    const int KCN = Cin / KC;
    const int64_t n = (int64_t)(Cout / 64) * KCN * 14 * 4 * 2 * 64;
    const float s = ldexpf(1.0f, wexp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int lane = (int)(i & 63);
        int64_t r = i >> 6;
        int hl = (int)(r & 1); r >>= 1;
        int cb = (int)(r & 3); r >>= 2;
        int pair = (int)(r % 14); r /= 14;
        int kc = (int)(r % KCN);
        int ntile = (int)(r / KCN);
        int co = ntile * 64 + cb * 16 + (lane & 15);
        int kgq = lane >> 4;
        int tap = 2 * pair + (kgq >> 1);
        int ci0 = kc * KC + (kgq & 1) * 8;
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = tap < 27 ? w[((int64_t)co * Cin + ci0 + j) * 27 + tap] * s : 0.f;
            _Float16 hh = (_Float16)x;
            v[j] = hl ? (_Float16)(x - (float)hh) : hh;
        }
        out[i] = *reinterpret_cast<uint4*>(&v);
    }
}

// The same packing with the 32 (co) x 16 (ci) x 27 block of one (N tile, K chunk, column half) staged through LDS: 32 rows of
// 432 contiguous floats read once with 16-byte loads, 56 fragments of 1 KB written out (round 4: the gather above read
// every element twice with a 108-byte stride, 175-580 us per deep layer, and training re-packs every iteration).
constexpr int PK16_ROW = KC * 27 + 1;
__global__ void __launch_bounds__(256) pack_mfma16_tiled(const float* __restrict__ w, int Cin, int Cout, int wexp,
                                                         uint4* __restrict__ out) {
    extern __shared__ float pk16_lds[];                     // [32][PK16_ROW]
    const int KCN = Cin / KC;
    const int half = blockIdx.x & 1;                        // column blocks cb = 2 half, 2 half + 1
    const int kc = (blockIdx.x >> 1) % KCN, ntile = (blockIdx.x >> 1) / KCN;
    const float s = ldexpf(1.0f, wexp);
    const float* src0 = w + ((int64_t)(ntile * 64 + half * 32) * Cin + kc * KC) * 27;
    bfm_stage_rows<32, KC * 27, PK16_ROW, 256>(src0, (int64_t)Cin * 27, pk16_lds, s,
                                               ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && (Cin & 3) == 0);
    __syncthreads();
    uint4* dst = out + (int64_t)(blockIdx.x >> 1) * (14 * 4 * 2 * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kgq = lane >> 4;
    for (int q = wave; q < 14 * 2; q += 4) {                // (pair, cb within the half): hi and lo planes
        const int cbl = q & 1, pair = q >> 1;
        const int tap = 2 * pair + (kgq >> 1);              // 27 = the padding tap of the last pair: zeros
        const float* src = pk16_lds + (cbl * 16 + (lane & 15)) * PK16_ROW + ((kgq & 1) * 8) * 27 + min(tap, 26);
        half8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = tap < 27 ? src[j * 27] : 0.f;
            const _Float16 hh = (_Float16)x;
            vh[j] = hh;
            vl[j] = (_Float16)(x - (float)hh);
        }
        dst[((pair * 4 + half * 2 + cbl) * 2 + 0) * 64 + lane] = *reinterpret_cast<uint4*>(&vh);
        dst[((pair * 4 + half * 2 + cbl) * 2 + 1) * 64 + lane] = *reinterpret_cast<uint4*>(&vl);
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised, persistent variant: 8 waves per workgroup, one workgroup per CU, each workgroup
// walks a strided list of output tiles.  Waves 0-3 (one per SIMD) only read LDS and issue MFMAs;
// waves 4-7 (their SIMD partners) are loaders running ONE K-chunk ahead -- across tile boundaries
// too, so a tile's first chunk is staged while the previous tile is still being multiplied and
// stored: they write the next chunk's halo tile into the other half of a double-buffered A image
// (global fp32 -> GroupNorm affine -> fp16 hi/lo planes) and stream the weight rows by LDS-DMA two
// rows ahead.  VALU/VMEM work of the loaders runs beside the matrix pipe instead of in front of it.
// One s_barrier per (kd,kh) row is the only synchronisation:
//   loaders, before barrier R : counted vmcnt (DMA of row R landed) + lgkmcnt(0) (their ds_writes done)
//   after barrier R           : consumers read B slot R%3 (and, at r==0, the A buffer written during the
//                               previous chunk); loaders refill B slot (R+2)%3 -- consumed during row R-1 --
//                               and stage 1-2 elements per thread of the next chunk into the A buffer the
//                               consumers finished with one chunk ago.
__device__ __forceinline__ void ws_tile_coords(const ConvParams& p, int lin, int& mt, int& nt) {
    const int nblk = p.nMt * p.NT;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = lin & 7, idx = lin >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    mt = bid / p.NT;
    nt = bid - mt * p.NT;
}

template <int WM, int WN, int NPASS>
__global__ void __launch_bounds__(512, 2) conv_mfma_ws(const ConvParams p) {
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int ROWFR = 12 * WN;
    constexpr int KPR = ROWFR / 4;                 // DMA instructions per loader wave per row
    constexpr int PTHR = 256;                      // loader threads
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int split = blockIdx.y;
    const int boxN = p.TD * p.TH * p.TW;
    const int ntiles = p.nMt * p.NT;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;

    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, p.bound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        aexp = 14 - ex;
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    const int a_bytes = 2 * NPL * p.plane_stride;         // one A buffer
    const int b_base = 2 * a_bytes;
    const int kc_begin = split * p.kc_per_split;
    const int kc_end = min(p.KCN, kc_begin + p.kc_per_split);
    const int nchunks = kc_end - kc_begin;
    const int total_chunks = my_tiles * nchunks;           // flattened (tile, chunk) sequence of this workgroup
    const int total_rows = total_chunks * 9;
    if (total_chunks <= 0) return;

    if (producer) {
        // =========================== loader waves ===========================
        const int ptid = tid - 256;
        const int pw = wave - 4;
        constexpr int MAX_IT = 12;
        const int n_el = p.nvox_lds * 4;
        const int q4 = ptid & 3;
        int off[MAX_IT];
        int z0 = 0, y0 = 0, x0 = 0;
        auto compute_offsets = [&](bool fromB) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < MAX_IT; ++it) {
                const int e = ptid + it * PTHR;
                off[it] = -2;
                if (e < n_el) {
                    const int vox = e >> 2;
                    const int hz = vox / (p.HT * p.WT);
                    const int rem = vox - hz * (p.HT * p.WT);
                    const int hy = rem / p.WT;
                    const int hx = rem - hy * p.WT;
                    const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                    off[it] = -1;
                    if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                        if (fromB) off[it] = ((p.up.mapD[gz] * p.up.h + p.up.mapH[gy]) * p.up.w + p.up.mapW[gx]) * p.CB;
                        else off[it] = ((gz * p.H + gy) * p.W + gx) * p.CA;
                    }
                }
            }
        };
        bool off_fromB = false;
        const int st_plane = ((q4 >> 1) * NPL) * p.plane_stride + (q4 & 1) * 8;

        float4 v[12];                                        // at most two of them live at a time (static indices)
        float sc[4], sh[4];
        const float* src = nullptr;
        // chunk g of the flattened sequence: (tile g / nchunks, K-chunk g % nchunks)
        auto begin_chunk = [&](int g) __attribute__((always_inline)) {
            const int c = g % nchunks;
            const int c0 = (kc_begin + c) * KC;
            const bool fromB = c0 >= p.CA;
            if (c == 0) {
                int mt, nt;
                ws_tile_coords(p, (int)blockIdx.x + (g / nchunks) * G, mt, nt);
                const int tx = mt % p.nTx;
                const int ty = (mt / p.nTx) % p.nTy;
                const int tz = mt / (p.nTx * p.nTy);
                z0 = tz * p.TD; y0 = ty * p.TH; x0 = tx * p.TW;
                off_fromB = fromB;
                compute_offsets(fromB);
            } else if (fromB != off_fromB) {
                off_fromB = fromB;
                compute_offsets(fromB);
            }
            src = (fromB ? p.B + (c0 - p.CA) : p.A + c0) + q4 * 4;
            const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + c0 + q4 * 4);
            const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + c0 + q4 * 4);
            sc[0] = sc4.x * a_scale; sc[1] = sc4.y * a_scale; sc[2] = sc4.z * a_scale; sc[3] = sc4.w * a_scale;
            sh[0] = sh4.x * a_scale; sh[1] = sh4.y * a_scale; sh[2] = sh4.z * a_scale; sh[3] = sh4.w * a_scale;
        };
        // element `it` (compile-time): one unconditional 16-B load (clamped offset) so that every wave issues
        // the same number of VMEM instructions -- the counted vmcnt below depends on it
        auto load_el = [&](auto itc) __attribute__((always_inline)) {
            constexpr int it = decltype(itc)::value;
            const int o = off[it] >= 0 ? off[it] : 0;
            v[it] = *reinterpret_cast<const float4*>(src + o);
        };
        auto write_el = [&](auto itc, int buf) __attribute__((always_inline)) {
            constexpr int it = decltype(itc)::value;
            const int o = off[it];
            const int e = ptid + it * PTHR;
            if (o != -2) {
                const bool ok = o >= 0;
                float y[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
                half4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = ok ? fmaf(y[i], sc[i], sh[i]) : 0.f;
                    _Float16 hh = (_Float16)t;
                    hi[i] = hh;
                    lo[i] = (_Float16)(t - (float)hh);
                }
                unsigned char* dst = lds + buf * a_bytes + st_plane + (e >> 2) * 16;
                *reinterpret_cast<half4*>(dst) = hi;
                if constexpr (NPASS == 3) *reinterpret_cast<half4*>(dst + p.plane_stride) = lo;
            }
        };
        auto issue_row = [&](int R) __attribute__((always_inline)) {
            const int Rs = R < total_rows ? R : total_rows - 1;    // past the end: harmless reload into a free slot
            const int g = Rs / 9, rr = Rs - g * 9;
            int mt, nt;
            ws_tile_coords(p, (int)blockIdx.x + (g / nchunks) * G, mt, nt);
            const int kcR = kc_begin + g % nchunks;
            const int slot = R % 3;
#pragma unroll
            for (int i = 0; i < KPR; ++i) {
                const int f = pw + i * 4;
                const int j = f / 12, gg = f - j * 12;
                const uint4* wsrc = p.wp + ((size_t)((nt * WN + j) * p.KCN + kcR) * 27 + rr * 3) * (4 * FRAG_U4) +
                                    gg * FRAG_U4 + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
                                                 (__attribute__((address_space(3))) void*)(lds + b_base +
                                                                                          (slot * ROWFR + f) * 1024),
                                                 16, 0, 0);
            }
        };
        auto for_el = [&](auto first, auto count, auto&& fn) __attribute__((always_inline)) {
            constexpr int f0 = decltype(first)::value, n = decltype(count)::value;
            if constexpr (n >= 1) fn(std::integral_constant<int, f0>{});
            if constexpr (n >= 2) fn(std::integral_constant<int, f0 + 1>{});
        };
#define BFM_IC(n) std::integral_constant<int, n>{}

        // prologue: chunk 0 of the first tile into A buffer 0; weight rows 0 and 1 into slots 0 and 1
        begin_chunk(0);
        for_el(BFM_IC(0), BFM_IC(2), [&](auto i) { load_el(i); });  for_el(BFM_IC(0), BFM_IC(2), [&](auto i) { write_el(i, 0); });
        for_el(BFM_IC(2), BFM_IC(2), [&](auto i) { load_el(i); });  for_el(BFM_IC(2), BFM_IC(2), [&](auto i) { write_el(i, 0); });
        for_el(BFM_IC(4), BFM_IC(2), [&](auto i) { load_el(i); });  for_el(BFM_IC(4), BFM_IC(2), [&](auto i) { write_el(i, 0); });
        for_el(BFM_IC(6), BFM_IC(2), [&](auto i) { load_el(i); });  for_el(BFM_IC(6), BFM_IC(2), [&](auto i) { write_el(i, 0); });
        for_el(BFM_IC(8), BFM_IC(2), [&](auto i) { load_el(i); });  for_el(BFM_IC(8), BFM_IC(2), [&](auto i) { write_el(i, 0); });
        for_el(BFM_IC(10), BFM_IC(2), [&](auto i) { load_el(i); }); for_el(BFM_IC(10), BFM_IC(2), [&](auto i) { write_el(i, 0); });
        issue_row(0);
        issue_row(1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

        // Row schedule of the next chunk's staging (12 element slots per thread over rows 0..8):
        //   loads : rows 0-3 -> 2 elements each (0..7), rows 4-7 -> 1 element each (8..11)
        //   writes: one row after the load.
        // Queue order inside a row: [writes of the previous row's loads] [this row's loads] [DMA of row R+2].
        // Before barrier R the DMA of row R (issued two rows earlier, ahead of the previous row's loads and
        // DMA) must have landed: vmcnt(loads of the previous row + KPR).
        for (int g = 0; g < total_chunks; ++g) {
            const bool more = g + 1 < total_chunks;
            const int nbuf = (g + 1) & 1;
            if (more) begin_chunk(g + 1);
            auto row = [&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                constexpr int NL_PREV = r == 0 ? 0 : (r <= 4 ? 2 : 1);        // loads issued in row r-1
                constexpr int NL = r <= 3 ? 2 : (r <= 7 ? 1 : 0);             // loads issued in row r
                constexpr int F_PREV = r == 0 ? 0 : (r <= 4 ? 2 * (r - 1) : 8 + (r - 5));
                constexpr int F = r <= 3 ? 2 * r : 8 + (r - 4);
                const int R = g * 9 + r;
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL_PREV + KPR) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPR) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (more) {
                    for_el(BFM_IC(F_PREV), BFM_IC(NL_PREV), [&](auto i) { write_el(i, nbuf); });
                    for_el(BFM_IC(F), BFM_IC(NL), [&](auto i) { load_el(i); });
                }
                issue_row(R + 2);                                // always issued: the counted vmcnt relies on it
            };
            row(BFM_IC(0)); row(BFM_IC(1)); row(BFM_IC(2)); row(BFM_IC(3)); row(BFM_IC(4));
            row(BFM_IC(5)); row(BFM_IC(6)); row(BFM_IC(7)); row(BFM_IC(8));
        }
#undef BFM_IC
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        return;
    }

    // =========================== MFMA waves ===========================
    const int wm = wave / WN, wn = wave % WN;
    const int l32 = lane & 31, khalf = lane >> 5;
    int a_off[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        int q = wm * 64 + mb * 32 + row_perm(l32);
        int d = 0, h = 0, w = 0;
        if (q < boxN) box_coords(p, q, d, h, w);
        int vox = (d * p.HT + h) * p.WT + w;
        a_off[mb] = (khalf * NPL) * p.plane_stride + vox * 16;
    }
    floatx16 acc[2][2];
    const bool final_out = p.splitk == 1;
    float* obase = p.out + (final_out ? 0 : (int64_t)split * p.split_stride);

    for (int g = 0; g < total_chunks; ++g) {
        const int c = g % nchunks;
        if (c == 0) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
        }
        const unsigned char* abuf = lds + (g & 1) * a_bytes;
        for (int r = 0; r < 9; ++r) {
            const int R = g * 9 + r;
            __builtin_amdgcn_s_barrier();
            const unsigned char* bs = lds + b_base + ((R % 3) * ROWFR + wn * 12) * 1024 + lane * 16;
            const int kd = r / 3, kh = r - kd * 3;
            const int roff = (kd * p.HT + kh) * p.WT * 16;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                half8 a[2][NPL], b[2][NPL];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl)
                        b[nb][hl] = *reinterpret_cast<const half8*>(bs + (kw * 4 + nb * 2 + hl) * 1024);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl)
                        a[mb][hl] = *reinterpret_cast<const half8*>(abuf + a_off[mb] + hl * p.plane_stride + roff +
                                                                    kw * 16);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        if constexpr (NPASS == 3) {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb][1], b[nb][0], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb][0], b[nb][1], acc[mb][nb], 0, 0, 0);
                        }
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb][0], b[nb][0], acc[mb][nb], 0, 0, 0);
                    }
            }
        }
        if (c == nchunks - 1) {
            // ---- epilogue of this tile (the loaders are already staging the next tile)
            int mt, nt;
            ws_tile_coords(p, (int)blockIdx.x + (g / nchunks) * G, mt, nt);
            const int tx = mt % p.nTx;
            const int ty = (mt / p.nTx) % p.nTy;
            const int tz = mt / (p.nTx * p.nTy);
            const int z0 = tz * p.TD, y0 = ty * p.TH, x0 = tx * p.TW;
            const int ntw = nt * WN + wn;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = (i >> 2) * 8 + khalf * 4 + (i & 3);
                    const int q = wm * 64 + mb * 32 + row_perm(rr);
                    if (q >= boxN) continue;
                    int d, h, w;
                    box_coords(p, q, d, h, w);
                    const int gz = z0 + d, gy = y0 + h, gx = x0 + w;
                    if (gz >= p.D || gy >= p.H || gx >= p.W) continue;
                    float* orow = obase + (((int64_t)gz * p.H + gy) * p.W + gx) * p.Cout + ntw * 64 + l32;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        float r = acc[mb][nb][i] * dq;
                        if (final_out) {
                            if (p.accum) r = r + orow[nb * 32];
                            r = r >= 0.f ? r : r * p.slope;
                        }
                        orow[nb * 32] = r;
                    }
                }
            }
        }
    }
}

// sum split-K slabs in slab order, then LeakyReLU
__global__ void splitk_reduce(const float* __restrict__ ws, int splitk, int64_t n4, int64_t stride4, float slope,
                              int accum, float* __restrict__ out) {
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 s = w4[i];
        for (int k = 1; k < splitk; ++k) {
            float4 t = w4[i + k * stride4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (accum) {
            float4 t = o4[i];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        s.x = s.x >= 0.f ? s.x : s.x * slope; s.y = s.y >= 0.f ? s.y : s.y * slope;
        s.z = s.z >= 0.f ? s.z : s.z * slope; s.w = s.w >= 0.f ? s.w : s.w * slope;
        o4[i] = s;
    }
}

// The same, emitting the output's moment rows (what the conv epilogues write when there is no split): workgroup
// (rb, cz, s) sums rows [rb * vpb, (rb + 1) * vpb) of sample s on its CGB column quads; thread (vl, cgl) takes every NV-th
// voxel, the NV partials of a column fold in lane order (fp32 per thread over <= vpb / NV values, fp64 across threads, as
// the epilogues do).  Row (s * nrows + rb) of the tables.  Deterministic: fixed order, no atomics.
// column quads per workgroup of splitk_reduce_rows: the largest of 64 / 32 / 16 that divides Cout / 4 (a multiple of 16, as
// Cout % 64 == 0), so that the grid's y extent covers every quad exactly and 256 / CGB voxel lanes is a whole number for
// every such Cout (192, 320, 384 ... included: ADVICE r5); unchanged for the widths 64 * 2^k
__host__ __device__ inline int splitk_cgb(int CG) { return (CG & 63) == 0 ? 64 : ((CG & 31) == 0 ? 32 : 16); }

__global__ void __launch_bounds__(256) splitk_reduce_rows(const float* __restrict__ ws, int splitk, int64_t stride4,
                                                          int nvox, int Cout, int vpb, int nrows, float slope, int accum,
                                                          float* __restrict__ out, double* __restrict__ rsum,
                                                          double* __restrict__ rsq, float* __restrict__ rmn,
                                                          float* __restrict__ rmx) {
    __shared__ float4 lsh[4][256];
    const int CG = Cout >> 2;
    const int CGB = splitk_cgb(CG);
    const int NV = 256 / CGB;
    const int tid = threadIdx.x;
    const int cgl = tid % CGB, vl = tid / CGB;
    const int c4 = blockIdx.y * CGB + cgl;
    const int smp = blockIdx.z, rb = blockIdx.x;
    const int v0 = rb * vpb, v1 = min(nvox, v0 + vpb);
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    float4* o4 = reinterpret_cast<float4*>(out);
    float4 fs = make_float4(0.f, 0.f, 0.f, 0.f), fq = fs;
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int v = v0 + vl; v < v1; v += NV) {
        const int64_t i = ((int64_t)smp * nvox + v) * CG + c4;
        float4 a = w4[i];
        for (int k = 1; k < splitk; ++k) {
            const float4 t = w4[i + k * stride4];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        if (accum) {
            const float4 t = o4[i];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        a.x = a.x >= 0.f ? a.x : a.x * slope; a.y = a.y >= 0.f ? a.y : a.y * slope;
        a.z = a.z >= 0.f ? a.z : a.z * slope; a.w = a.w >= 0.f ? a.w : a.w * slope;
        o4[i] = a;
        fs.x += a.x; fs.y += a.y; fs.z += a.z; fs.w += a.w;
        fq.x = fmaf(a.x, a.x, fq.x); fq.y = fmaf(a.y, a.y, fq.y); fq.z = fmaf(a.z, a.z, fq.z); fq.w = fmaf(a.w, a.w, fq.w);
        mn.x = fminf(mn.x, a.x); mn.y = fminf(mn.y, a.y); mn.z = fminf(mn.z, a.z); mn.w = fminf(mn.w, a.w);
        mx.x = fmaxf(mx.x, a.x); mx.y = fmaxf(mx.y, a.y); mx.z = fmaxf(mx.z, a.z); mx.w = fmaxf(mx.w, a.w);
    }
    lsh[0][tid] = fs; lsh[1][tid] = fq; lsh[2][tid] = mn; lsh[3][tid] = mx;
    __syncthreads();
    if (vl == 0) {
        double S[4] = {0.0, 0.0, 0.0, 0.0}, Q[4] = {0.0, 0.0, 0.0, 0.0};
        float MN[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, MX[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int j = 0; j < NV; ++j) {
            const float4 a = lsh[0][j * CGB + cgl], b = lsh[1][j * CGB + cgl], c = lsh[2][j * CGB + cgl], d = lsh[3][j * CGB + cgl];
            S[0] += (double)a.x; S[1] += (double)a.y; S[2] += (double)a.z; S[3] += (double)a.w;
            Q[0] += (double)b.x; Q[1] += (double)b.y; Q[2] += (double)b.z; Q[3] += (double)b.w;
            MN[0] = fminf(MN[0], c.x); MN[1] = fminf(MN[1], c.y); MN[2] = fminf(MN[2], c.z); MN[3] = fminf(MN[3], c.w);
            MX[0] = fmaxf(MX[0], d.x); MX[1] = fmaxf(MX[1], d.y); MX[2] = fmaxf(MX[2], d.z); MX[3] = fmaxf(MX[3], d.w);
        }
        const size_t o = ((size_t)smp * nrows + rb) * Cout + (size_t)c4 * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { rsum[o + k] = S[k]; rsq[o + k] = Q[k]; rmn[o + k] = MN[k]; rmx[o + k] = MX[k]; }
    }
}

// rows of voxels one workgroup of splitk_reduce_rows folds into one moment row (a multiple of its voxel lanes; at most 128
// rows per sample, so that the batched finalize reads the table directly), and the row count
static int splitk_rows_vpb(int64_t nvox, int Cout) {
    const int CG = Cout / 4, CGB = splitk_cgb(CG), NV = 256 / CGB;
    int64_t vpb = std::max<int64_t>((int64_t)NV * 4, bfm_cdiv64(nvox, 128));
    vpb = bfm_cdiv64(vpb, NV) * NV;
    return (int)std::min<int64_t>(vpb, 0x7fffffff);
}

// packed[ntile64][kc][tap][nb][hl][lane] (uint4 = 8 halfs): lane l holds
// B[k = 8*(l>>5)+j][n = l&31] = w[co = ntile*64 + nb*32 + (l&31)][ci = kc*16 + 8*(l>>5) + j][tap] * 2^wexp
// The same packing with the 64 (co) x 16 (ci) x 27 block of one (N tile, K chunk) staged through LDS: its 64 rows of
// 432 contiguous floats are read coalesced once, and the 108 fragments (110 KB, contiguous in `out`) are written
// coalesced.  pack_mfma below gathers every element with a 108-byte stride and reads it twice (hi and lo lanes): 1.5 ms
// for the 2048 x 1024 layer, and training re-packs every layer every iteration.
constexpr int PK_ROW = KC * 27 + 1;                         // odd row pitch: lanes of a fragment differ in co
__global__ void __launch_bounds__(256) pack_mfma_tiled(const float* __restrict__ w, int Cin, int Cout, int wexp,
                                                       uint4* __restrict__ out) {
    extern __shared__ float pk_lds[];                       // [32][PK_ROW]: one 32-channel column block (55 KB: 2 blocks / CU)
    const int KCN = Cin / KC;
    const int nb = blockIdx.x & 1;
    const int kc = (blockIdx.x >> 1) % KCN, ntile = (blockIdx.x >> 1) / KCN;
    const float s = ldexpf(1.0f, wexp);
    const float* src0 = w + ((int64_t)(ntile * 64 + nb * 32) * Cin + kc * KC) * 27;
    bfm_stage_rows<32, KC * 27, PK_ROW, 256>(src0, (int64_t)Cin * 27, pk_lds, s,
                                             ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && (Cin & 3) == 0);
    __syncthreads();
    uint4* dst = out + (int64_t)(blockIdx.x >> 1) * (27 * 2 * 2 * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int tap = wave; tap < 27; tap += 4) {              // hi and lo planes of this column block's tap
        const float* src = pk_lds + (lane & 31) * PK_ROW + ((lane >> 5) * 8) * 27 + tap;
        half8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = src[j * 27];
            const _Float16 hh = (_Float16)x;
            vh[j] = hh;
            vl[j] = (_Float16)(x - (float)hh);
        }
        dst[((tap * 2 + nb) * 2 + 0) * 64 + lane] = *reinterpret_cast<uint4*>(&vh);
        dst[((tap * 2 + nb) * 2 + 1) * 64 + lane] = *reinterpret_cast<uint4*>(&vl);
    }
}

__global__ void pack_mfma(const float* __restrict__ w, int Cin, int Cout, int wexp, uint4* __restrict__ out) {
    const int KCN = Cin / KC;
    const int64_t n = (int64_t)(Cout / 64) * KCN * 27 * 2 * 2 * 64;
    const float s = ldexpf(1.0f, wexp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int lane = (int)(i & 63);
        int64_t r = i >> 6;
        int hl = (int)(r & 1); r >>= 1;
        int nb = (int)(r & 1); r >>= 1;
        int tap = (int)(r % 27); r /= 27;
        int kc = (int)(r % KCN);
        int ntile = (int)(r / KCN);
        int co = ntile * 64 + nb * 32 + (lane & 31);
        int ci0 = kc * KC + (lane >> 5) * 8;
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = w[((int64_t)co * Cin + ci0 + j) * 27 + tap] * s;
            _Float16 hh = (_Float16)x;
            v[j] = hl ? (_Float16)(x - (float)hh) : hh;
        }
        out[i] = *reinterpret_cast<uint4*>(&v);
    }
}

constexpr int nslot_for(int WN) { return WN == 1 ? 3 : 2; }
constexpr int ring_bytes(int WN) { return nslot_for(WN) * 12 * WN * 1024; }

struct HostPlan {
    int WM, WN, TD, TH, TW, splitk;
    int ver;        // 0: 4-wave kernel, 2 workgroups/CU; 1: wave-specialised 8-wave kernel, 1 workgroup/CU
};

constexpr int LDS_LIMIT_WS = 160 * 1024;
int ws_smem(int npl, int plane_stride, int WN) { return 2 * (2 * npl * plane_stride) + 3 * 12 * WN * 1024; }

constexpr int LDS_LIMIT = 80 * 1024;      // two workgroups per CU (160 KiB)

int plane_stride_for(int nvox) { return ((nvox * 16 + 63) / 64) * 64 + 32; }
int plane_stride16_for(int nvox) { return ((nvox * 16 + 127) / 128) * 128; }      // k-half planes 256 B-aligned apart
constexpr int ring16_bytes(int WN) { return WN == 1 ? 2 * 2 * 8 * 1024 : 3 * 1 * 16 * 1024; }

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

HostPlan choose_plan(int Cin, int Cout, int D, int H, int W) {
    HostPlan hp{};
    if (Cout % 128 == 0) { hp.WM = 2; hp.WN = 2; } else { hp.WM = 4; hp.WN = 1; }
    const int rows = hp.WM * 64;
    double best = 1e300;
    for (int tw = 1; tw <= std::min(W, 32); ++tw)
        for (int th = 1; th <= std::min(H, rows / tw); ++th) {
            int td = std::min(D, rows / (tw * th));
            if (td < 1) continue;
            int nvox = (td + 2) * (th + 2) * (tw + 2);
            if (4 * plane_stride_for(nvox) + ring_bytes(hp.WN) > LDS_LIMIT) continue;
            if (nvox * 4 > 12 * 64 * hp.WM * hp.WN) continue;          // MAX_IT staging slots
            double tiles = (double)bfm_cdiv(D, td) * bfm_cdiv(H, th) * bfm_cdiv(W, tw);
            double conflict = (tw % 16 == 0) ? 1.0 : 1.25;
            // MFMA work per tile is fixed (rows x 64*WN x K); staging grows with the halo
            double cost = tiles * (rows * 27.0 + nvox * 4.0 * conflict);
            if (cost < best) { best = cost; hp.TD = td; hp.TH = th; hp.TW = tw; }
        }
    const int KCN = Cin / KC;
    int nMt = bfm_cdiv(D, hp.TD) * bfm_cdiv(H, hp.TH) * bfm_cdiv(W, hp.TW);
    int wgs = nMt * (Cout / (64 * hp.WN));
    hp.splitk = 1;
    if (wgs < 256 && KCN >= 8) {
        int want = bfm_cdiv(512, wgs);
        int maxs = KCN / 4;
        int s = std::min(want, maxs);
        if (s > 1) {
            int per = bfm_cdiv(KCN, s);
            hp.splitk = bfm_cdiv(KCN, per);
        }
    }
    // kernel variant (measured on MI355X, profiles/): the wave-specialised persistent kernel wins once a
    // tile has >= 8 K-chunks and needs no split-K; shallow-K layers keep the 2-workgroup/CU kernel.
    hp.ver = (KCN >= 8 && hp.splitk == 1) ? 1 : 0;
    if (const char* e = getenv("BFM_CONV_VER")) hp.ver = atoi(e);
    return hp;
}

void launch16(const ConvParams& p, int passes, bool wm4, dim3 grid, size_t smem, hipStream_t st) {
    if (wm4) {
        if (passes == 3) hipLaunchKernelGGL((conv_mfma16<4, 1, 3, 2, 2>), grid, dim3(256), smem, st, p);
        else hipLaunchKernelGGL((conv_mfma16<4, 1, 1, 2, 2>), grid, dim3(256), smem, st, p);
    } else {
        if (passes == 3) hipLaunchKernelGGL((conv_mfma16<2, 2, 3, 1, 3>), grid, dim3(256), smem, st, p);
        else hipLaunchKernelGGL((conv_mfma16<2, 2, 1, 1, 3>), grid, dim3(256), smem, st, p);
    }
}

template <int WM, int WN>
void launch_ws(const ConvParams& p, int passes, dim3 grid, size_t smem, hipStream_t st) {
    if (passes == 3) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_ws<WM, WN, 3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_WS);
        hipLaunchKernelGGL((conv_mfma_ws<WM, WN, 3>), grid, dim3(512), smem, st, p);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_ws<WM, WN, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_WS);
        hipLaunchKernelGGL((conv_mfma_ws<WM, WN, 1>), grid, dim3(512), smem, st, p);
    }
}

template <int WM, int WN>
void launch(const ConvParams& p, int passes, dim3 grid, size_t smem, hipStream_t st) {
    if (passes == 3) hipLaunchKernelGGL((conv_mfma<WM, WN, 3, nslot_for(WN)>), grid, dim3(64 * WM * WN), smem, st, p);
    else hipLaunchKernelGGL((conv_mfma<WM, WN, 1, nslot_for(WN)>), grid, dim3(64 * WM * WN), smem, st, p);
}

}  // namespace

extern "C" size_t bfm_pack_conv_weights_mfma_bytes(int Cin, int Cout) {
    if (Cin % KC || Cout % 64) return 0;
    return (size_t)(Cout / 64) * (Cin / KC) * 27 * 2 * 2 * 64 * 16;
}

extern "C" int bfm_pack_conv_weights_mfma(const float* w, int Cin, int Cout, float wmax_abs_host, void* wpacked,
                                          int* wexp_host, bfm_stream_t stream) {
    if (!w || !wpacked || !wexp_host || Cin <= 0 || Cout <= 0) return BFM_E_ARG;
    if (Cin % KC || Cout % 64) return BFM_E_SHAPE;
    int wexp = 0;
    if (wmax_abs_host > 0.f && wmax_abs_host < INFINITY) {
        int ex;
        (void)frexpf(wmax_abs_host, &ex);
        wexp = 14 - ex;
        wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp);
    }
    *wexp_host = wexp;
    const int64_t nblk = (int64_t)(Cout / 64) * (Cin / KC) * 2;
    const size_t smem = (size_t)32 * PK_ROW * sizeof(float);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_mfma_tiled), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return BFM_E_LAUNCH;
        attr = true;
    }
    if (nblk <= 0x7fffffff) {
        hipLaunchKernelGGL(pack_mfma_tiled, dim3((unsigned)nblk), dim3(256), smem, bfm_s(stream), w, Cin, Cout, wexp,
                           reinterpret_cast<uint4*>(wpacked));
    } else {
        int64_t n = (nblk / 2) * 27 * 2 * 2 * 64;
        int nb = (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256));
        hipLaunchKernelGGL(pack_mfma, dim3(nb), dim3(256), 0, bfm_s(stream), w, Cin, Cout, wexp,
                           reinterpret_cast<uint4*>(wpacked));
    }
    return bfm_launch_status();
}

extern "C" size_t bfm_pack_conv_weights_mfma16_bytes(int Cin, int Cout) {
    if (Cin % KC || Cout % 64) return 0;
    return (size_t)(Cout / 64) * (Cin / KC) * 14 * 4 * 2 * 64 * 16;
}

extern "C" int bfm_pack_conv_weights_mfma16(const float* w, int Cin, int Cout, float wmax_abs_host, void* wpacked,
                                            int* wexp_host, bfm_stream_t stream) {
    if (!w || !wpacked || !wexp_host || Cin <= 0 || Cout <= 0) return BFM_E_ARG;
    if (Cin % KC || Cout % 64) return BFM_E_SHAPE;
    int wexp = 0;
    if (wmax_abs_host > 0.f && wmax_abs_host < INFINITY) {
        int ex;
        (void)frexpf(wmax_abs_host, &ex);
        wexp = 14 - ex;
        wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp);
    }
    *wexp_host = wexp;
    const int64_t nblk = (int64_t)(Cout / 64) * (Cin / KC) * 2;
    if (nblk <= 0x7fffffff) {
        const size_t smem = (size_t)32 * PK16_ROW * sizeof(float);
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_mfma16_tiled),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return BFM_E_LAUNCH;
            attr = true;
        }
        hipLaunchKernelGGL(pack_mfma16_tiled, dim3((unsigned)nblk), dim3(256), smem, bfm_s(stream), w, Cin, Cout, wexp,
                           reinterpret_cast<uint4*>(wpacked));
        return bfm_launch_status();
    }
    int64_t n = (int64_t)(Cout / 64) * (Cin / KC) * 14 * 4 * 2 * 64;
    int nb = (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256));
    hipLaunchKernelGGL(pack_mfma16, dim3(nb), dim3(256), 0, bfm_s(stream), w, Cin, Cout, wexp,
                       reinterpret_cast<uint4*>(wpacked));
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_mfma_plan(int Cin, int Cout, int D, int H, int W, int* cfg) {
    if (!cfg || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return BFM_E_ARG;
    if (Cin % KC || Cout % 64) return BFM_E_SHAPE;
    HostPlan hp = choose_plan(Cin, Cout, D, H, W);
    if (hp.TD == 0) return BFM_E_SHAPE;
    cfg[0] = hp.WM; cfg[1] = hp.WN; cfg[2] = hp.TD; cfg[3] = hp.TH; cfg[4] = hp.TW; cfg[5] = hp.splitk;
    cfg[6] = hp.ver; cfg[7] = 0;
    return BFM_OK;
}

extern "C" size_t bfm_conv3x3x3_mfma_workspace(int Cin, int Cout, int D, int H, int W, int splitk) {
    (void)Cin;
    if (splitk <= 1) return 0;
    return (size_t)splitk * D * H * W * Cout * sizeof(float);
}

// rows of the output-moment table a launch with this plan writes (0: this plan cannot emit them)
extern "C" int bfm_conv3x3x3_mfma_rows(int Cin, int Cout, int D, int H, int W, const int* cfg) {
    if (!cfg || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (cfg[6] >= 3 || cfg[6] < 0) return 0;                    // the Winograd variants have their own row counts
    int splitk = cfg[5] < 1 ? 1 : cfg[5];
    const int KCN = Cin / KC;
    if (splitk > KCN) splitk = KCN;
    if (splitk > 1 && bfm_cdiv(KCN, bfm_cdiv(KCN, splitk)) > 1) {  // split-K: the slab reduction writes the rows (round 5)
        if (Cout % 64) return 0;
        const int64_t nvox = (int64_t)D * H * W;
        return (int)bfm_cdiv64(nvox, splitk_rows_vpb(nvox, Cout));
    }
    if (cfg[6] == 1) return 0;                                  // the persistent variant does not emit rows
    if (cfg[2] < 1 || cfg[3] < 1 || cfg[4] < 1) return 0;
    return bfm_cdiv(D, cfg[2]) * bfm_cdiv(H, cfg[3]) * bfm_cdiv(W, cfg[4]);
}

extern "C" int bfm_conv3x3x3_mfma_ex(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                                     const bfm_upsample_t* up, const float* scale, const float* shift,
                                     const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                     int passes, const int* cfg, float* out, void* workspace, size_t workspace_bytes,
                                     void* moment_rows, bfm_stream_t stream);

extern "C" int bfm_conv3x3x3_mfma(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                                  const bfm_upsample_t* up, const float* scale, const float* shift,
                                  const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                  int passes, const int* cfg, float* out, void* workspace, size_t workspace_bytes,
                                  bfm_stream_t stream) {
    return bfm_conv3x3x3_mfma_ex(A, CA, B, CB, D, H, W, up, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes,
                                 cfg, out, workspace, workspace_bytes, nullptr, stream);
}

static int conv_mfma_launch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                            const bfm_upsample_t* up, const float* scale, const float* shift, const float* bound, int G,
                            const void* wpacked, int wexp, int Cout, float slope, int passes, const int* cfg, float* out,
                            void* workspace, size_t workspace_bytes, void* moment_rows, int affine_stride, bfm_stream_t stream);

extern "C" int bfm_conv3x3x3_mfma_ex(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                                     const bfm_upsample_t* up, const float* scale, const float* shift,
                                     const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                     int passes, const int* cfg, float* out, void* workspace, size_t workspace_bytes,
                                     void* moment_rows, bfm_stream_t stream) {
    return conv_mfma_launch(A, CA, B, CB, 1, D, H, W, up, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, cfg,
                            out, workspace, workspace_bytes, moment_rows, 0, stream);
}

#ifdef BFM_STAMPS
extern "C" int bfm_debug_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(long long) * 4096);
}
#endif

extern "C" size_t bfm_conv3x3x3_mfma_batch_workspace(int Cin, int Cout, int S, int D, int H, int W, int splitk) {
    return (size_t)(S < 1 ? 1 : S) * bfm_conv3x3x3_mfma_workspace(Cin, Cout, D, H, W, splitk);
}

extern "C" int bfm_conv3x3x3_mfma_batch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                                        const bfm_upsample_t* up, const float* scale, const float* shift,
                                        const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                        int passes, const int* cfg, float* out, void* workspace, size_t workspace_bytes,
                                        void* moment_rows, int affine_stride, bfm_stream_t stream) {
    if (S < 1 || !cfg) return BFM_E_ARG;
    if (affine_stride != 0 && (affine_stride < CA + CB || (affine_stride & 3))) return BFM_E_ARG;
    if (cfg[6] != 0 && cfg[6] != 2) return BFM_E_SHAPE;        // the persistent and Winograd variants take one sample
    return conv_mfma_launch(A, CA, B, CB, S, D, H, W, up, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, cfg,
                            out, workspace, workspace_bytes, moment_rows, affine_stride, stream);
}

static int conv_mfma_launch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                            const bfm_upsample_t* up, const float* scale, const float* shift, const float* bound, int G,
                            const void* wpacked, int wexp, int Cout, float slope, int passes, const int* cfg, float* out,
                            void* workspace, size_t workspace_bytes, void* moment_rows, int affine_stride,
                            bfm_stream_t stream) {
    if (!A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift || !bound || G <= 0 || !wpacked || !out)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || !up->mapD || !up->mapH || !up->mapW || up->d <= 0 || up->h <= 0 ||
                              up->w <= 0)))
        return BFM_E_ARG;
    if (CA % KC || CB % KC || Cout % 64) return BFM_E_SHAPE;
    if (passes != 1 && passes != 3) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (CB > 0 && (reinterpret_cast<uintptr_t>(B) & 15)) ||
        (reinterpret_cast<uintptr_t>(scale) & 15) || (reinterpret_cast<uintptr_t>(shift) & 15) ||
        (reinterpret_cast<uintptr_t>(wpacked) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return BFM_E_ARG;
    const int Cin = CA + CB;
    HostPlan hp;
    if (cfg) {
        hp = HostPlan{cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], cfg[6]};
    } else {
        hp = choose_plan(Cin, Cout, D, H, W);
    }
    const bool wg_ok = (hp.WM == 4 && hp.WN == 1) || (hp.WM == 2 && hp.WN == 2);
    if (!wg_ok || hp.TD < 1 || hp.TH < 1 || hp.TW < 1 || hp.TD * hp.TH * hp.TW > hp.WM * 64) return BFM_E_SHAPE;
    if (Cout % (64 * hp.WN)) return BFM_E_SHAPE;
    if (hp.TD + 2 > 1023 || hp.TH + 2 > 1023 || hp.TW + 2 > 1023) return BFM_E_SHAPE;

    ConvParams p{};
    p.A = A; p.B = B; p.CA = CA; p.CB = CB; p.D = D; p.H = H; p.W = W;
    p.up = make_upview(up);
    p.scale = scale; p.shift = shift; p.bound = bound; p.G = G;
    p.wp = reinterpret_cast<const uint4*>(wpacked);
    p.wexp = wexp; p.Cout = Cout; p.slope = slope;
    p.accum = (cfg && (cfg[7] & 1)) ? 1 : 0;
    p.TD = hp.TD; p.TH = hp.TH; p.TW = hp.TW; p.HT = hp.TH + 2; p.WT = hp.TW + 2;
    const int nTz = bfm_cdiv(D, hp.TD);
    p.nTy = bfm_cdiv(H, hp.TH); p.nTx = bfm_cdiv(W, hp.TW);
    p.nMtS = nTz * p.nTy * p.nTx;
    p.S = S;
    p.saff = affine_stride > 0 ? affine_stride : CA + CB;
    p.nMt = S * p.nMtS;
    p.NT = Cout / (64 * hp.WN);
    p.KCN = Cin / KC;
    p.splitk = hp.splitk < 1 ? 1 : hp.splitk;
    if (p.splitk > p.KCN) p.splitk = p.KCN;
    p.kc_per_split = bfm_cdiv(p.KCN, p.splitk);
    p.splitk = bfm_cdiv(p.KCN, p.kc_per_split);
    p.nvox_lds = (hp.TD + 2) * p.HT * p.WT;
    p.plane_stride = hp.ver == 2 ? plane_stride16_for(p.nvox_lds) : plane_stride_for(p.nvox_lds);
    p.tw_shift = is_pow2(hp.TW) ? ilog2(hp.TW) : -1;
    p.thw_shift = is_pow2(hp.TH * hp.TW) ? ilog2(hp.TH * hp.TW) : -1;
    const int npl = passes == 3 ? 2 : 1;
    const size_t smem = hp.ver == 1 ? (size_t)ws_smem(npl, p.plane_stride, hp.WN)
                        : hp.ver == 2 ? (size_t)2 * npl * p.plane_stride + ring16_bytes(hp.WN)
                                      : (size_t)2 * npl * p.plane_stride + ring_bytes(hp.WN);
    if (smem > (size_t)(hp.ver == 1 ? LDS_LIMIT_WS : LDS_LIMIT)) return BFM_E_SHAPE;
    if (hp.ver < 0 || hp.ver > 2) return BFM_E_ARG;
    if (p.nvox_lds * 4 > 12 * 64 * hp.WM * hp.WN) return BFM_E_SHAPE;
    const int64_t nvox = (int64_t)D * H * W;
    p.sA = nvox * CA;
    p.sB = CB > 0 ? (int64_t)up->d * up->h * up->w * CB : 0;
    p.sO = nvox * Cout;
    p.split_stride = (int64_t)S * nvox * Cout;
    if (p.splitk > 1) {
        if (!workspace || workspace_bytes < (size_t)p.splitk * S * nvox * Cout * sizeof(float)) return BFM_E_WORKSPACE;
        if (reinterpret_cast<uintptr_t>(workspace) & 15) return BFM_E_ARG;
        p.out = static_cast<float*>(workspace);
    } else {
        p.out = out;
    }
    if (moment_rows && p.splitk > 1) {
        if (reinterpret_cast<uintptr_t>(moment_rows) & 7) return BFM_E_ARG;      // written by splitk_reduce_rows below
    } else if (moment_rows) {
        if (hp.ver == 1) return BFM_E_SHAPE;                   // see bfm_conv3x3x3_mfma_rows
        if (reinterpret_cast<uintptr_t>(moment_rows) & 7) return BFM_E_ARG;
        char* rb = static_cast<char*>(moment_rows);
        const size_t n = (size_t)p.nMt * Cout;
        p.rsum = reinterpret_cast<double*>(rb);
        p.rsq = reinterpret_cast<double*>(rb + n * 8);
        p.rmn = reinterpret_cast<float*>(rb + n * 16);
        p.rmx = reinterpret_cast<float*>(rb + n * 20);
    }
#ifdef BFM_MFMA_ABLATE
    if (const char* e = getenv("BFM_MFMA_ABL")) p.abl = atoi(e);
#endif
    if ((int64_t)hp.TD * H * W * Cout > 0x7fffffffLL) return BFM_E_SHAPE;     // 32-bit row offsets inside a box (build_row_table)
    if ((int64_t)p.nMt * p.NT > 0x7fffffff) return BFM_E_SHAPE;
    if (nvox * CA > 0x7fffffffLL || (CB > 0 && (int64_t)up->d * up->h * up->w * CB > 0x7fffffffLL))
        return BFM_E_SHAPE;                                   // the staging path keeps 32-bit element offsets
    unsigned gx = (unsigned)(p.nMt * p.NT);
    if (hp.ver == 1) {
        const int cap = 256;                                   // one 8-wave workgroup per CU
        if ((int)gx > cap) gx = (unsigned)cap;
    }
    dim3 grid(gx, (unsigned)p.splitk);
    hipStream_t st = bfm_s(stream);
    if (hp.ver == 2) {
        launch16(p, passes, hp.WM == 4, grid, smem, st);
    } else if (hp.ver == 1) {
        if (hp.WM == 4) launch_ws<4, 1>(p, passes, grid, smem, st);
        else launch_ws<2, 2>(p, passes, grid, smem, st);
    } else {
        if (hp.WM == 4) launch<4, 1>(p, passes, grid, smem, st);
        else launch<2, 2>(p, passes, grid, smem, st);
    }
    int rc = bfm_launch_status();
    if (rc != BFM_OK) return rc;
    if (p.splitk > 1 && moment_rows) {
        if (nvox > 0x7fffffffLL || S > 65535) return BFM_E_SHAPE;
        const int vpb = splitk_rows_vpb(nvox, Cout);
        const int nrows = (int)bfm_cdiv64(nvox, vpb);
        const int CG = Cout / 4, CGB = splitk_cgb(CG);
        char* rb = static_cast<char*>(moment_rows);
        const size_t n = (size_t)S * nrows * Cout;
        hipLaunchKernelGGL(splitk_reduce_rows, dim3(nrows, CG / CGB, S), dim3(256), 0, st,
                           static_cast<const float*>(workspace), p.splitk, p.split_stride / 4, (int)nvox, Cout, vpb, nrows,
                           slope, p.accum, out, reinterpret_cast<double*>(rb), reinterpret_cast<double*>(rb + n * 8),
                           reinterpret_cast<float*>(rb + n * 16), reinterpret_cast<float*>(rb + n * 20));
        rc = bfm_launch_status();
    } else if (p.splitk > 1) {
        int64_t n4 = (int64_t)S * nvox * Cout / 4;
        int nb = (int)std::min<int64_t>(2048, bfm_cdiv64(n4, 256));
        hipLaunchKernelGGL(splitk_reduce, dim3(nb), dim3(256), 0, st, static_cast<const float*>(workspace), p.splitk,
                           n4, p.split_stride / 4, slope, p.accum, out);
        rc = bfm_launch_status();
    }
    return rc;
}
