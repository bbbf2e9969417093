// 3x3x3 convolution with a Winograd F(2,3) transform along x: 1.5x fewer matrix-core FLOPs for the same result.
//
// The conv_mfma family is bound by what the matrix pipe sustains under its power limit (HISTORY.md section 3.1), so the only
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
#include "wino_shared.h"
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
    double *rsum, *rsq;              // optional output-moment rows [nMt][Cout] (see conv3d_mfma.hip), 4-wave kernel only
    float *rmn, *rmx;
    const float* mask_img;           // optional (D,H,W) image: a box none of whose voxels is non-zero there is not computed
    const unsigned char* uni_flags;  // optional [nMt]: 0, or 1 + the class of a box whose operands equal its class mates'
    const int* uni_first;            // [27] first box of each class (>= nMt: none): the ones conv_wino_rest computes
    float* uni_acc;                  // [27][NT][2][16][NTHR][2]: those boxes' output-transformed sums, for their class mates
    const int* list;                 // sparse forms: the boxes to compute, ascending
    const int* list_n;               // ... their number, on the device
    // fused MaxPool3d(2) (the POOL kernels: box 8 x 8 x 4, every box inside the tensor): the pooled output (D/2,H/2,W/2,Cout)
    // and its own moment rows [nMt][Cout]
    float* pool_out;
    double *prsum, *prsq;
    float *prmn, *prmx;
};

__device__ __forceinline__ int row_perm(int l) {        // as conv_mfma: each 16-lane b128 group reads 16 consecutive positions
    if (l < 4) return l;
    if (l < 12) return l + 12;
    if (l < 16) return l - 8;
    if (l < 20) return l + 8;
    if (l < 28) return l - 12;
    return l;
}

__device__ __forceinline__ int row_unperm(int q) {      // inverse of row_perm
    if (q < 4) return q;
    if (q < 8) return q + 8;
    if (q < 16) return q + 12;
    if (q < 24) return q - 12;
    if (q < 28) return q - 8;
    return q;
}

__device__ __forceinline__ void pair_coords(const WinoParams& p, int q, int& d, int& h, int& j) {
    d = q >> p.thp_shift;
    const int rem = q & ((1 << p.thp_shift) - 1);
    h = rem >> p.pw_shift;
    j = rem & ((1 << p.pw_shift) - 1);
}

// x = hi + lo in fp16 for four values: hi by v_cvt_pkrtz_f16_f32 (truncation, two values per instruction: x - hi is then
// exact in fp32 and at most 2^-10 |x|), lo = fp16(x - hi) by one v_fma_mixlo/hi_f16 per value (fp32 = fp16 * -1 + fp32,
// rounded to nearest; round 5: in place of convert-back, subtract and re-pack -- 6 instead of 12 instructions per four
// values); the neglected lo*lo product stays below 2^-20 relative.
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
template <bool LO>
__device__ __forceinline__ void split_store4(const float (&t)[4], unsigned char* dp, int plane_stride) {
    const fp16x2_t h01 = __builtin_amdgcn_cvt_pkrtz(t[0], t[1]);
    const fp16x2_t h23 = __builtin_amdgcn_cvt_pkrtz(t[2], t[3]);
    uint2 hv;
    hv.x = __builtin_bit_cast(unsigned, h01);
    hv.y = __builtin_bit_cast(unsigned, h23);
    *reinterpret_cast<uint2*>(dp) = hv;
    if constexpr (LO) {
        uint2 lv;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lv.x) : "v"(hv.x), "v"(t[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lv.x) : "v"(hv.x), "v"(t[1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lv.y) : "v"(hv.y), "v"(t[2]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lv.y) : "v"(hv.y), "v"(t[3]));
        *reinterpret_cast<uint2*>(dp + plane_stride) = lv;
    }
}

// MODE 0: every box in full.  1 (conv_wino_masked): boxes without input are not computed.  2 / 3 (conv_wino_rest,
// conv_wino_uniform; launched as a pair): boxes flagged uniform -- the network's input is constant over everything the box
// can see (the background of a head volume), so the layer's input around the box is a function of the distances to the
// tile's faces alone.  Boxes of one CLASS (per axis: first box, last box, or in between -- 27 classes) then multiply the
// same operands in the same order, voxel for voxel, and end the main loop with the same accumulators: conv_wino_rest
// computes the unflagged boxes and the first flagged box of every class, whose output-transformed sums (before accumulate
// mode and the activation) it also leaves in uni_acc; conv_wino_uniform gives every other flagged box its class's sums --
// the bits its own main loop and transform would have produced -- in the same thread order, and runs the rest of the
// epilogue (accumulate mode, activation, store, moment rows) on them: no staging, no weights, no matrix products, no LDS.  Two kernels rather than a branch: a second loop body in one kernel made the
// allocator spill 54 registers and the layer 25 % slower.
// POOL = 1 (encoder layers whose output nn.MaxPool3d(2) reads next, buildingblocks.py:185-186; box 8 x 8 x 4 with every box
// inside the tensor, checked by the host): the epilogue also writes the pooled tensor and its moment rows -- the 2 x 2 x 2
// windows are whole inside a box, so the separate pooling launch and its read of the full-resolution output go away.  A
// thread's 16 pairs are (j, h) = (row0 & 1, (row0 >> 1) + 4 (it & 1)), d = it >> 1: the x pair of a window is the Winograd
// pair, its d pair is in the thread, its h pair in the thread 64 further (row0 ^ 2), met through LDS.  max in the order
// bfm_maxpool2 takes it (dz, dy, dx), NaN kept the same way.
// NaN-sticky like torch's max: a NaN seen at any corner of the window stays (fmaxf alone would drop it on the next corner)
__device__ __forceinline__ float pool_max(float m, float q) { return (m != m) ? m : ((q != q) ? q : fmaxf(m, q)); }

template <int NPASS, int MODE, int POOL = 0>
__device__ __forceinline__ void conv_wino_body(const WinoParams& p) {
    constexpr bool UNI = MODE == 3;                      // this kernel computes the flagged boxes (one row block each)
    constexpr bool BY_FLAG = MODE == 2 || MODE == 3;
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int NF = 2 * NPL;
    constexpr int BATCH = (NPASS == 3) ? 1 : 3;      // staging items loaded together (register budget: 128 accumulators)
    constexpr int NSET = 3;                           // weight register sets: NSET-1 taps ahead
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    // One workgroup per (box, cout tile), XCD-aware bijective remap (neighbouring boxes share halo lines in one L2).  The
    // sparse forms take their boxes from a list built on the device (wino_mask_list_kernel / uniform_lists_kernel): the
    // grid still covers every box -- the host does not know the count, and a persistent loop over the list made the
    // compiler spill 150 registers -- but a workgroup beyond the list reads one cached scalar and ends, where round 2's
    // workgroups each loaded their box of the mask image (a memory round trip per round of 512 resident workgroups:
    // conv_wino_rest took 18 ms of a step where its share of the dense launches is 12.5).
    constexpr bool LIST = MODE != 0;
    const int nblk = LIST ? p.list_n[0] * p.NT : p.nMt * p.NT;
    const int bid = blockIdx.x;
    if (LIST && bid >= nblk) return;                     // workgroup-uniform, before any barrier
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int pos = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = Winograd position 0..3
    const int l32 = lane & 31, khalf = lane >> 5;
    int item;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = LIST ? p.list[item / p.NT] : item / p.NT;
    const int nt = item % p.NT;
    const int tx = mt % p.nTx;
    const int ty = (mt / p.nTx) % p.nTy;
    const int tz = mt / (p.nTx * p.nTy);
    const int z0 = tz * p.TD, y0 = ty * p.TH, x0 = tx * p.TW;
    // conv_wino_rest's list: the unflagged boxes and the first flagged box of every class (rep), which leaves its sums for
    // its class mates; conv_wino_uniform's list: those mates
    int cls = 0;                                         // 1 + class of a flagged box
    if constexpr (BY_FLAG) cls = __builtin_amdgcn_readfirstlane((int)p.uni_flags[mt]);
    const bool rep = MODE == 2 && cls != 0;

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
    fetch(1, wq[1]);

    auto do_chunk = [&](int kc, auto par_tag, auto mbn_tag) __attribute__((always_inline)) {
        (void)par_tag;
        constexpr int MBN = decltype(mbn_tag)::value;      // row blocks multiplied: 4, or 1 for a uniform box
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
                if (msk[it] == 15) {                     // interior item (the common case): no padding selects
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float y[4] = {v[u][i].x, v[u][i].y, v[u][i].z, v[u][i].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) dd[i][c] = fmaf(y[c], sc[c], sh[c]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool ok = msk[it] & (1 << i);
                        const float y[4] = {v[u][i].x, v[u][i].y, v[u][i].z, v[u][i].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) dd[i][c] = ok ? fmaf(y[c], sc[c], sh[c]) : 0.f;
                    }
                }
                const int e = tid + it * NTHR;
                unsigned char* dst = lds + st_plane + (e >> 2) * 16;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    float t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        t[c] = ps == 0 ? dd[0][c] - dd[2][c]
                             : ps == 1 ? dd[1][c] + dd[2][c]
                             : ps == 2 ? dd[2][c] - dd[1][c]
                                       : dd[1][c] - dd[3][c];
                    unsigned char* dp = dst + (ps * 2 * NPL) * p.plane_stride;
                    split_store4<NPASS == 3>(t, dp, p.plane_stride);
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
            const int cur = t % 3;
            uint4 bw[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) bw[f] = wq[cur][f];
            // Issue the loads of tap t+2 HERE and pin them: left alone, the scheduler sinks each load to just before
            // its first use (shorter live ranges) and the L2 latency is paid at every tap (ISA: vmcnt(0) four MFMAs
            // after the load).  An opaque inline-asm load would be faster still but is unsafe: the allocator may move
            // or spill a register whose load is still in flight.
            fetch(s + 2, wq[(cur + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MBN; ++mb) {
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
    if constexpr (!UNI) {
        using MbTag = std::integral_constant<int, 4>;
        if constexpr (NSET == 3) {
            for (int kc = 0; kc < p.KCN; ++kc) do_chunk(kc, std::integral_constant<int, 0>{}, MbTag{});
        } else {
            for (int kc = 0; kc < p.KCN; kc += 2) {        // 9 steps per chunk: the parity flips every chunk
                do_chunk(kc, std::integral_constant<int, 0>{}, MbTag{});
                if (kc + 1 < p.KCN) do_chunk(kc + 1, std::integral_constant<int, 1>{}, MbTag{});
            }
        }
    }
    // conv_wino_rest leaves, for the first flagged box of a class, the output-transformed sums (before accumulate mode and
    // the activation) of every thread's 16 pairs in uni_acc [class][nt][nb][it][thread] (y0, y1); conv_wino_uniform reads
    // them back in the same thread order and goes on from there
    float2* const ybuf = BY_FLAG && cls ? reinterpret_cast<float2*>(p.uni_acc) +
                                              ((size_t)(cls - 1) * p.NT + nt) * (2 * 16 * NTHR) + tid
                                        : nullptr;

    // ================= epilogue: output transform through LDS =================
    float* m = reinterpret_cast<float*>(lds);                      // [4 positions][128 accumulator rows][MLD]
    float* mw = m + (pos * 128 + khalf * 4) * MLD + l32;           // this lane's write base
    const int col = tid & 31, row0 = tid >> 5;
    // Interior boxes (all of a 160^3 tile but its last slabs) take a path without per-item index arithmetic: the
    // thread's 16 pairs are q = row0 + 8 it, and with power-of-two box sides the coordinates of q split into a
    // per-thread part (from row0, computed once) and a wave-uniform part (from it, on the scalar unit); the address
    // is then uniform base + one 32-bit lane offset.  The general path spent ~1600 of the epilogue's 3200 vector
    // instructions per box on q -> (d,h,j) -> address chains, 480 of them quarter-rate 32/64-bit multiplies -- more
    // vector work than the four K-chunks of a 64-channel layer's main loop.
    // (the POOL kernels are launched on tensors their box tiles exactly: no other path is compiled for them)
    const bool interior = POOL == 1 || (z0 + p.TD <= p.D && y0 + p.TH <= p.H && x0 + p.TW <= p.W);      // wave-uniform
    const int th_shift = p.thp_shift - p.pw_shift;
    unsigned off_t = 0;
    int un0 = 0, un3 = 0;
    if (interior) {
        const int jt = p.pw_shift >= 3 ? row0 : (row0 & ((1 << p.pw_shift) - 1));
        const int rt = p.pw_shift >= 3 ? 0 : (row0 >> p.pw_shift);
        const int h_t = rt & ((1 << th_shift) - 1), d_t = rt >> th_shift;
        off_t = (unsigned)(((d_t * p.H + h_t) * p.W + 2 * jt) * p.Cout + col);
        un0 = row_unperm(row0);                                    // accumulator row of pair (q & 31) = row0
        un3 = row_unperm(row0 + 24);                               //                              = row0 + 24
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        __syncthreads();                                           // A planes (or the previous round) fully consumed
        float fs = 0.f, fq = 0.f, fmn = INFINITY, fmx = -INFINITY;   // this thread's column, its 16 pairs (<= 32 values)
        float pm[8];                                               // POOL: this thread's row of its 8 windows (hb, dp)
#pragma unroll
        for (int k = 0; k < 8; ++k) pm[k] = -INFINITY;
        if (interior) {
            float* ob = p.out + nt * 64 + nb * 32;
            float prev0[16], prev1[16];
            if (p.accum) {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int qu = 8 * it;
                    const int j_u = qu & ((1 << p.pw_shift) - 1), r_u = qu >> p.pw_shift;
                    const int h_u = r_u & ((1 << th_shift) - 1), d_u = r_u >> th_shift;
                    const float* o = ob + (((int64_t)(z0 + d_u) * p.H + (y0 + h_u)) * p.W + x0 + 2 * j_u) * p.Cout;
                    prev0[it] = o[off_t];
                    prev1[it] = o[off_t + (unsigned)p.Cout];
                }
            }
            if constexpr (!UNI) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        mw[(mb * 32 + (i >> 2) * 8 + (i & 3)) * MLD] = acc[mb][nb][i];
                __syncthreads();
            }
            const float* mr = m + col;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int qu = 8 * it;
                const int j_u = qu & ((1 << p.pw_shift) - 1), r_u = qu >> p.pw_shift;
                const int h_u = r_u & ((1 << th_shift) - 1), d_u = r_u >> th_shift;
                float* o = ob + (((int64_t)(z0 + d_u) * p.H + (y0 + h_u)) * p.W + x0 + 2 * j_u) * p.Cout;
                float y0v, y1v;
                if constexpr (UNI) {
                    const float2 yy = ybuf[(nb * 16 + it) * NTHR];
                    y0v = yy.x; y1v = yy.y;
                } else {
                    // accumulator row holding pair q: (q & ~31) + row_unperm(q & 31), q & 31 = row0 + 8 (it & 3)
                    const int sub = it & 3;
                    const int qr = (qu & ~31) + (sub == 0 ? un0 : sub == 1 ? row0 + 8 + 12 : sub == 2 ? row0 + 16 - 12 : un3);
                    const float m0 = mr[(0 * 128 + qr) * MLD], m1 = mr[(1 * 128 + qr) * MLD];
                    const float m2 = mr[(2 * 128 + qr) * MLD], m3 = mr[(3 * 128 + qr) * MLD];
                    y0v = ((m0 + m1) + m2) * dq;
                    y1v = ((m1 - m2) - m3) * dq;
                    if (MODE == 2 && rep) ybuf[(nb * 16 + it) * NTHR] = make_float2(y0v, y1v);
                }
                if (p.accum) { y0v = y0v + prev0[it]; y1v = y1v + prev1[it]; }
                y0v = y0v >= 0.f ? y0v : y0v * p.slope;
                y1v = y1v >= 0.f ? y1v : y1v * p.slope;
                o[off_t] = y0v;
                o[off_t + (unsigned)p.Cout] = y1v;
                fs += y0v; fq = fmaf(y0v, y0v, fq); fmn = fminf(fmn, y0v); fmx = fmaxf(fmx, y0v);
                fs += y1v; fq = fmaf(y1v, y1v, fq); fmn = fminf(fmn, y1v); fmx = fmaxf(fmx, y1v);
                if constexpr (POOL == 1) {                          // window (hb, dp) = (it & 1, it >> 2): dz-major, then dx
                    pm[(it & 1) * 4 + (it >> 2)] = pool_max(pool_max(pm[(it & 1) * 4 + (it >> 2)], y0v), y1v);
                }
            }
        } else {
        // accumulate mode: what `out` holds is fetched now, so that its latency hides under the LDS exchange
        float prev0[16], prev1[16];
        if (p.accum) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int q = row0 + 8 * it;
                int d, h, j;
                pair_coords(p, q, d, h, j);
                const int gz = z0 + d, gy = y0 + h, gx = x0 + 2 * j;
                prev0[it] = 0.f;
                prev1[it] = 0.f;
                if (gz < p.D && gy < p.H && gx < p.W) {
                    const float* o = p.out + (((int64_t)gz * p.H + gy) * p.W + gx) * p.Cout + nt * 64 + nb * 32 + col;
                    prev0[it] = o[0];
                    if (gx + 1 < p.W) prev1[it] = o[p.Cout];
                }
            }
        }
        if constexpr (!UNI) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int i = 0; i < 16; ++i)                        // accumulator row order (compile-time offsets);
                    mw[(mb * 32 + (i >> 2) * 8 + (i & 3)) * MLD] = acc[mb][nb][i];   // the reader undoes row_perm
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int q = row0 + 8 * it;
            int d, h, j;
            pair_coords(p, q, d, h, j);
            const int gz = z0 + d, gy = y0 + h, gx = x0 + 2 * j;
            if (gz >= p.D || gy >= p.H || gx >= p.W) continue;
            float* o = p.out + (((int64_t)gz * p.H + gy) * p.W + gx) * p.Cout + nt * 64 + nb * 32 + col;
            float y0v, y1v;
            if constexpr (UNI) {
                const float2 yy = ybuf[(nb * 16 + it) * NTHR];
                y0v = yy.x; y1v = yy.y;
            } else {
                const int qr = (q & ~31) + row_unperm(q & 31);       // accumulator row holding pair q
                const float m0 = m[(0 * 128 + qr) * MLD + col], m1 = m[(1 * 128 + qr) * MLD + col];
                const float m2 = m[(2 * 128 + qr) * MLD + col], m3 = m[(3 * 128 + qr) * MLD + col];
                y0v = ((m0 + m1) + m2) * dq;
                y1v = ((m1 - m2) - m3) * dq;
                if (MODE == 2 && rep) ybuf[(nb * 16 + it) * NTHR] = make_float2(y0v, y1v);
            }
            if (p.accum) y0v = y0v + prev0[it];
            y0v = y0v >= 0.f ? y0v : y0v * p.slope;
            o[0] = y0v;
            fs += y0v; fq = fmaf(y0v, y0v, fq); fmn = fminf(fmn, y0v); fmx = fmaxf(fmx, y0v);
            if (gx + 1 < p.W) {
                if (p.accum) y1v = y1v + prev1[it];
                y1v = y1v >= 0.f ? y1v : y1v * p.slope;
                o[p.Cout] = y1v;
                fs += y1v; fq = fmaf(y1v, y1v, fq); fmn = fminf(fmn, y1v); fmx = fmaxf(fmx, y1v);
            }
        }
        }
        if (p.rsum != nullptr) {
            // moment row of this tile: fold the 8 row groups of every column in fixed order (scratch behind m)
            // (conv_wino_uniform uses no other LDS: its fold sits at offset 0 and the launch asks for 6 KB)
            double* ls = reinterpret_cast<double*>(lds + (UNI ? 0 : 4 * 128 * MLD * sizeof(float)));   // [8][32]
            double* lq = ls + 256;
            float* lmn = reinterpret_cast<float*>(lq + 256);
            float* lmx = lmn + 256;
            ls[row0 * 32 + col] = (double)fs; lq[row0 * 32 + col] = (double)fq;
            lmn[row0 * 32 + col] = fmn; lmx[row0 * 32 + col] = fmx;
            __syncthreads();
            if (tid < 32) {
                double S = 0.0, Q = 0.0;
                float MN = INFINITY, MX = -INFINITY;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    S += ls[r * 32 + tid]; Q += lq[r * 32 + tid];
                    MN = fminf(MN, lmn[r * 32 + tid]); MX = fmaxf(MX, lmx[r * 32 + tid]);
                }
                const size_t o = (size_t)mt * p.Cout + nt * 64 + nb * 32 + tid;
                p.rsum[o] = S; p.rsq[o] = Q; p.rmn[o] = MN; p.rmx[o] = MX;
            }
        }
        if constexpr (POOL == 1) {
            if (interior) {                                            // (the host launches these kernels on such tensors only)
            // window (hb, dp) of the pair of threads (row0, row0 ^ 2) = rows h, h ^ 1: each thread has the maximum over its
            // row's two d slices and x pair (pm, in bfm_maxpool2's order dz, dx); the even-h thread finishes the windows of
            // hb = 0, the odd-h one those of hb = 1 (dy = 0 first)
            const int POOL_OFF = UNI ? 6144 : 0;                       // m is consumed (barrier below); UNI: behind its row fold
            float* xs = reinterpret_cast<float*>(lds + POOL_OFF);      // [8 (hb, dp)][NTHR]
            __syncthreads();                                           // every thread is done with m
#pragma unroll
            for (int k = 0; k < 8; ++k) xs[k * NTHR + tid] = pm[k];
            __syncthreads();
            const int hb = (row0 >> 1) & 1;
            const int te = tid & ~64, to = tid | 64;                   // the pair's even-h / odd-h thread (row0 bit 1 = tid bit 6)
            float ps = 0.f, pq = 0.f, pmn = INFINITY, pmx = -INFINITY;
            const int H2 = p.H >> 1, W2 = p.W >> 1;
            float* pb = p.pool_out + nt * 64 + nb * 32 + col +
                        (((int64_t)(z0 >> 1) * H2 + (y0 >> 1) + (row0 >> 2) + 2 * hb) * W2 + (x0 >> 1) + (row0 & 1)) * p.Cout;
            const unsigned zstep = (unsigned)(H2 * W2 * p.Cout);       // one pooled slice (fits: the tensor has < 2^31 elements)
            const float* xe = xs + hb * 4 * NTHR + te;
            const float* xo = xs + hb * 4 * NTHR + to;
#pragma unroll
            for (int dp = 0; dp < 4; ++dp) {
                const float mv = pool_max(pool_max(-INFINITY, xe[dp * NTHR]), xo[dp * NTHR]);
                pb[dp * zstep] = mv;
                ps += mv; pq = fmaf(mv, mv, pq); pmn = fminf(pmn, mv); pmx = fmaxf(pmx, mv);
            }
            if (p.prsum != nullptr) {
                double* ls2 = reinterpret_cast<double*>(lds + POOL_OFF + 8 * NTHR * sizeof(float));        // behind xs
                double* lq2 = ls2 + 256;
                float* lmn2 = reinterpret_cast<float*>(lq2 + 256);
                float* lmx2 = lmn2 + 256;
                ls2[row0 * 32 + col] = (double)ps; lq2[row0 * 32 + col] = (double)pq;
                lmn2[row0 * 32 + col] = pmn; lmx2[row0 * 32 + col] = pmx;
                __syncthreads();
                if (tid < 32) {
                    double S = 0.0, Q = 0.0;
                    float MN = INFINITY, MX = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        S += ls2[r * 32 + tid]; Q += lq2[r * 32 + tid];
                        MN = fminf(MN, lmn2[r * 32 + tid]); MX = fmaxf(MX, lmx2[r * 32 + tid]);
                    }
                    const size_t o = (size_t)mt * p.Cout + nt * 64 + nb * 32 + tid;
                    p.prsum[o] = S; p.prsq[o] = Q; p.prmn[o] = MN; p.prmx[o] = MX;
                }
            }
            }
        }
    }
}

template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino(const WinoParams p) { conv_wino_body<NPASS, 0>(p); }
// the dense form and the uniform-box pair with the pooling fused (POOL above)
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino_pool(const WinoParams p) { conv_wino_body<NPASS, 0, 1>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino_rest_pool(const WinoParams p) { conv_wino_body<NPASS, 2, 1>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino_uniform_pool(const WinoParams p) { conv_wino_body<NPASS, 3, 1>(p); }

// the same kernel for the tile loop's last convolution: boxes whose image voxels (p.mask_img) are all zero return at once
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino_masked(const WinoParams p) { conv_wino_body<NPASS, 1>(p); }

// the pair for the layers that read the network's first activations: boxes flagged uniform (bfm_uniform_boxes) run a
// quarter of the matrix products in conv_wino_uniform, the others as ever in conv_wino_rest
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino_rest(const WinoParams p) { conv_wino_body<NPASS, 2>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino_uniform(const WinoParams p) { conv_wino_body<NPASS, 3>(p); }

// Work lists for the sparse forms: list[] = the boxes with pred(box), ascending; returns their number (thread 0's value
// is the total).  One workgroup of 1024 threads; `sh` = 17 ints of LDS.
template <class Pred>
__device__ __forceinline__ int wino_compact(int nMt, Pred pred, int* __restrict__ list, int* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sh[16] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nMt; b0 += 1024) {
        const int mt = b0 + tid;
        const bool keep = mt < nMt && pred(mt);
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) sh[wv] = __popcll(bal);
        __syncthreads();
        int off = sh[16];
        for (int w = 0; w < wv; ++w) off += sh[w];
        if (keep) list[off + before] = mt;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += sh[w];
            sh[16] += t;
        }
        __syncthreads();
    }
    return sh[16];
}

// act[box] = 1 when a voxel of the box is non-zero in the (D,H,W) image (NaN counts); one wave per box
__global__ void __launch_bounds__(256) wino_box_active_kernel(const float* __restrict__ img, int D, int H, int W, int TD,
                                                               int TH, int TW, int nTy, int nTx, int nMt,
                                                               unsigned char* __restrict__ act) {
    const int mt = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (mt >= nMt) return;
    const int tx = mt % nTx, ty = (mt / nTx) % nTy, tz = mt / (nTx * nTy);
    const int z0 = tz * TD, y0 = ty * TH, x0 = tx * TW;
    const int thw = TH * TW;
    bool nz = false;
    for (int q = lane; q < TD * thw; q += 64) {
        const int d = q / thw, r = q - d * thw, h = r / TW, w = r - h * TW;
        const int gz = z0 + d, gy = y0 + h, gx = x0 + w;
        if (gz < D && gy < H && gx < W) nz = nz || img[((int64_t)gz * H + gy) * W + gx] != 0.f;
    }
    const bool any = __any(nz);
    if (lane == 0) act[mt] = any ? 1 : 0;
}

__global__ void __launch_bounds__(1024) wino_mask_list_kernel(const unsigned char* __restrict__ act, int nMt,
                                                               int* __restrict__ list, int* __restrict__ n_out) {
    __shared__ int sh[17];
    const int n = wino_compact(nMt, [&](int mt) { return act[mt] != 0; }, list, sh);
    if (threadIdx.x == 0) n_out[0] = n;
}

// flags[box] = 0, or 1 + class when the (D,H,W) image is bitwise constant over the box -- a box of the layer's grid at
// pooling level L, i.e. voxels [z0 << L, (z0 + TD) << L) of the image -- grown by `R` image voxels and clipped to the
// volume.  class = 9 cz + 3 cy + cx with c = 0 for the first box along the axis, 2 for the last (if there is more than
// one), 1 in between: what the box can see of the tile's faces (zero padding at every level on the way) is the same for
// all boxes of a class, because a box side is longer than the reach R at its level.  One workgroup per box.
__global__ void __launch_bounds__(256) uniform_boxes_kernel(const float* __restrict__ img, int D, int H, int W, int TD, int TH,
                                                            int TW, int nTz, int nTy, int nTx, int L, int R,
                                                            unsigned char* __restrict__ flags) {
    const int mt = blockIdx.x;
    const int tx = mt % nTx, ty = (mt / nTx) % nTy, tz = mt / (nTx * nTy);
    const int z0 = max(((tz * TD) << L) - R, 0), y0 = max(((ty * TH) << L) - R, 0), x0 = max(((tx * TW) << L) - R, 0);
    const int z1 = min((((tz + 1) * TD) << L) + R, D), y1 = min((((ty + 1) * TH) << L) + R, H);
    const int x1 = min((((tx + 1) * TW) << L) + R, W);
    const int ed = z1 - z0, eh = y1 - y0, ew = x1 - x0;
    int bad = 0;
    const unsigned ref = __float_as_uint(img[((int64_t)z0 * H + y0) * W + x0]);
    for (int q = threadIdx.x; q < ed * eh * ew; q += 256) {
        const int w = q % ew, r = q / ew, h = r % eh, d = r / eh;
        bad |= __float_as_uint(img[((int64_t)(z0 + d) * H + (y0 + h)) * W + (x0 + w)]) != ref ? 1 : 0;
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        const int cz = tz == 0 ? 0 : (tz == nTz - 1 ? 2 : 1), cy = ty == 0 ? 0 : (ty == nTy - 1 ? 2 : 1);
        const int cx = tx == 0 ? 0 : (tx == nTx - 1 ? 2 : 1);
        // a box that is not the last of its axis must not see the far face either: when the last box is a remainder
        // narrower than the reach, its neighbour's grown region passes the level's extent (d << L image voxels; the
        // floor of the pooling can only move that face inwards) and the box is no class mate of the true middle boxes
        const int dl = (D >> L) << L, hl = (H >> L) << L, wl = (W >> L) << L;
        const bool far = (tz != nTz - 1 && ((((tz + 1) * TD) << L) + R > dl)) ||
                         (ty != nTy - 1 && ((((ty + 1) * TH) << L) + R > hl)) ||
                         (tx != nTx - 1 && ((((tx + 1) * TW) << L) + R > wl));
        flags[mt] = bad || far ? 0 : (unsigned char)(1 + 9 * cz + 3 * cy + cx);
    }
}

// first[c] = index of the first box of class c (n: none), and the two work lists of the pair of launches that uses the
// flags: cnt[0], rest[] = the unflagged boxes and every class's first flagged box (conv_wino_rest: computed in full);
// cnt[1], uni[] = the other flagged boxes (conv_wino_uniform: their class's sums through the epilogue).  One workgroup.
__global__ void __launch_bounds__(1024) uniform_lists_kernel(const unsigned char* __restrict__ flags, int n,
                                                             int* __restrict__ first, int* __restrict__ cnt,
                                                             int* __restrict__ rest, int* __restrict__ uni) {
    __shared__ int sh[17];
    __shared__ int fst[27];
    if (threadIdx.x < 27) fst[threadIdx.x] = n;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int f = flags[i];
        if (f != 0) atomicMin(&fst[f - 1], i);              // integer minimum: order independent
    }
    __syncthreads();
    if (threadIdx.x < 27) first[threadIdx.x] = fst[threadIdx.x];
    const int n0 = wino_compact(n, [&](int mt) { const int f = flags[mt]; return f == 0 || fst[f - 1] == mt; }, rest, sh);
    if (threadIdx.x == 0) cnt[0] = n0;
    __syncthreads();
    const int n1 = wino_compact(n, [&](int mt) { const int f = flags[mt]; return f != 0 && fst[f - 1] != mt; }, uni, sh);
    if (threadIdx.x == 0) cnt[1] = n1;
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

// The same packing with the 64 (co) x 16 (ci) x 27 block of one (N tile, K chunk) staged through LDS (coalesced reads
// of 64 rows of 432 floats; every element is read once instead of 4 positions x hi/lo times with a 108-byte stride).
// Training re-packs every layer each iteration: pack_wino took 3 ms for the 2048 x 1024 layer.
constexpr int PKW_ROW = KC * 27 + 1;
__global__ void __launch_bounds__(256) pack_wino_tiled(const float* __restrict__ w, int Cin, int Cout, int wexp, int npl,
                                                       uint4* __restrict__ out) {
    extern __shared__ float pkw_lds[];                      // [32][PKW_ROW]: one 32-channel column block (2 blocks / CU)
    const int KCN = Cin / KC;
    const int nf = 2 * npl;
    const int nb = blockIdx.x & 1;
    const int kc = (blockIdx.x >> 1) % KCN, ntile = (blockIdx.x >> 1) / KCN;
    const float s = ldexpf(1.0f, wexp);
    const float* src0 = w + ((int64_t)(ntile * 64 + nb * 32) * Cin + kc * KC) * 27;
    bfm_stage_rows<32, KC * 27, PKW_ROW, 256>(src0, (int64_t)Cin * 27, pkw_lds, 1.0f,
                                              ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && (Cin & 3) == 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = wave; q < 4 * 9; q += 4) {                  // (ps, t) of this column block: hi and lo from one transform
        const int t = q % 9;
        const int ps = q / 9;
        const float* src = pkw_lds + (lane & 31) * PKW_ROW + (8 * (lane >> 5)) * 27 + t * 3;
        half8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g0 = src[j * 27], g1 = src[j * 27 + 1], g2 = src[j * 27 + 2];
            const float u = ps == 0 ? g0 : ps == 1 ? ((g0 + g1) + g2) * 0.5f : ps == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
            const float x = u * s;
            const _Float16 hh = (_Float16)x;
            vh[j] = hh;
            vl[j] = (_Float16)(x - (float)hh);
        }
        uint4* dst = out + ((((int64_t)(ntile * 4 + ps) * KCN + kc) * 9 + t) * nf + nb * npl) * 64 + lane;
        dst[0] = __builtin_bit_cast(uint4, vh);
        if (npl == 2) dst[64] = __builtin_bit_cast(uint4, vl);
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
        cost = cost * 1024 + npos;                                 // among equal box counts: the fewest halo'd positions
                                                                   // to transform, split and store (8 x 8 x 4: 200, 4 x 4 x 16: 288)
        if (best < 0 || cost < best) { best = cost; TD = o[0]; TH = o[1]; TW = o[2]; }
    }
    return best >= 0;
}

}  // namespace

bool bfm_wino_choose_box(int D, int H, int W, int npl, int& TD, int& TH, int& TW) { return choose_box(D, H, W, npl, TD, TH, TW); }

int bfm_wino_mask_list(const float* mask_img, int D, int H, int W, int TD, int TH, int TW, int nTy, int nTx, int nMt, void* ws,
                       bfm_stream_t stream) {
    unsigned char* act = static_cast<unsigned char*>(ws);
    int* cnt = reinterpret_cast<int*>(act + (((size_t)nMt + 3) & ~(size_t)3));
    hipLaunchKernelGGL(wino_box_active_kernel, dim3((unsigned)bfm_cdiv(nMt, 4)), dim3(256), 0, bfm_s(stream), mask_img, D, H, W,
                       TD, TH, TW, nTy, nTx, nMt, act);
    hipLaunchKernelGGL(wino_mask_list_kernel, dim3(1), dim3(1024), 0, bfm_s(stream), act, nMt, cnt + 1, cnt);
    return bfm_launch_status();
}

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
    const int64_t nblk = (int64_t)(Cout / 64) * (Cin / KC) * 2;
    const size_t smem = (size_t)32 * PKW_ROW * sizeof(float);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_wino_tiled), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return BFM_E_LAUNCH;
        attr = true;
    }
    if (nblk <= 0x7fffffff) {
        hipLaunchKernelGGL(pack_wino_tiled, dim3((unsigned)nblk), dim3(256), smem, bfm_s(stream), w_oidhw, Cin, Cout, wexp, npl,
                           static_cast<uint4*>(wpacked));
    } else {
        const int64_t n = (nblk / 2) * 4 * 9 * 2 * npl * 64;
        int nb = (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256));
        hipLaunchKernelGGL(pack_wino, dim3(nb), dim3(256), 0, bfm_s(stream), w_oidhw, Cin, Cout, wexp, npl,
                           static_cast<uint4*>(wpacked));
    }
    return bfm_launch_status();
}

// rows of the output-moment table the 4-wave kernel writes for this volume (its own box choice), 0 if it cannot run
extern "C" int bfm_conv3x3x3_wino_rows(int D, int H, int W, int passes) {
    int TD, TH, TW;
    if (D <= 0 || H <= 0 || W <= 0 || !choose_box(D, H, W, passes == 3 ? 2 : 1, TD, TH, TW)) return 0;
    return bfm_cdiv(D, TD) * bfm_cdiv(H, TH) * bfm_cdiv(W, TW);
}

extern "C" int bfm_conv3x3x3_wino_box(int D, int H, int W, int passes, int* box) {
    int TD, TH, TW;
    if (!box || D <= 0 || H <= 0 || W <= 0 || !choose_box(D, H, W, passes == 3 ? 2 : 1, TD, TH, TW)) return BFM_E_ARG;
    box[0] = TD; box[1] = TH; box[2] = TW;
    return BFM_OK;
}

extern "C" int bfm_conv3x3x3_wino_ex(const float* A, int CA, int D, int H, int W, const float* scale,
                                     const float* shift, const float* bound, int G, const void* wpacked, int wexp,
                                     int Cout, float slope, int passes, int flags, float* out, void* moment_rows,
                                     bfm_stream_t stream);

extern "C" int bfm_conv3x3x3_wino(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                  const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                  int passes, int flags, float* out, bfm_stream_t stream) {
    return bfm_conv3x3x3_wino_ex(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out,
                                 nullptr, stream);
}

static int wino_launch(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                       const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes, int flags,
                       float* out, void* moment_rows, const float* mask_img, bfm_stream_t stream,
                       const unsigned char* uni_flags = nullptr, float* uni_acc = nullptr, void* mask_ws = nullptr,
                       float* pool_out = nullptr, void* pool_rows = nullptr);

static bool pool_ok_box(int TD, int TH, int TW, int D, int H, int W) {
    return TD == 8 && TH == 8 && TW == 4 && D % 8 == 0 && H % 8 == 0 && W % 4 == 0;
}

// 1 when bfm_conv3x3x3_wino_pool / _wino_uniform_pool take a (D,H,W) tensor: the kernel's box there is 8 x 8 x 4 and tiles the
// tensor exactly (a 2 x 2 x 2 pooling window is then whole inside one box); 0: pool with bfm_maxpool2 afterwards
extern "C" int bfm_conv3x3x3_wino_pool_ok(int D, int H, int W, int passes) {
    int TD, TH, TW;
    if (D <= 0 || H <= 0 || W <= 0 || !choose_box(D, H, W, passes == 3 ? 2 : 1, TD, TH, TW)) return 0;
    return pool_ok_box(TD, TH, TW, D, H, W) ? 1 : 0;
}

// bfm_conv3x3x3_wino_ex that also writes nn.MaxPool3d(2) of its output (buildingblocks.py:185-186: what the next encoder
// level reads) into pooled (D/2,H/2,W/2,Cout) and that tensor's moment rows [bfm_conv3x3x3_wino_rows()][Cout] into
// pooled_rows (or NULL): out and moment_rows as bfm_conv3x3x3_wino_ex writes them, pooled the bits of bfm_maxpool2(out)
extern "C" int bfm_conv3x3x3_wino_pool(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                       const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                       int passes, int flags, float* out, void* moment_rows, float* pooled,
                                       void* pooled_rows, bfm_stream_t stream) {
    if (!pooled) return BFM_E_ARG;
    return wino_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, moment_rows,
                       nullptr, stream, nullptr, nullptr, nullptr, pooled, pooled_rows);
}

// the same for the uniform-box pair (bfm_conv3x3x3_wino_uniform)
extern "C" int bfm_conv3x3x3_wino_uniform_pool(const float* A, int CA, int D, int H, int W, const float* scale,
                                               const float* shift, const float* bound, int G, const void* wpacked,
                                               int wexp, int Cout, float slope, int passes, int flags, float* out,
                                               void* moment_rows, const unsigned char* uniform_flags, void* scratch,
                                               float* pooled, void* pooled_rows, bfm_stream_t stream) {
    if (!uniform_flags || !scratch || !pooled || (flags & ~1)) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(uniform_flags) & 3) || (reinterpret_cast<uintptr_t>(scratch) & 15)) return BFM_E_ARG;
    return wino_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, moment_rows,
                       nullptr, stream, uniform_flags, static_cast<float*>(scratch), nullptr, pooled, pooled_rows);
}

extern "C" size_t bfm_uniform_boxes_bytes(int D, int H, int W, int passes) {
    int TD, TH, TW;
    if (D <= 0 || H <= 0 || W <= 0 || !choose_box(D, H, W, passes == 3 ? 2 : 1, TD, TH, TW)) return 0;
    const size_t n = (size_t)bfm_cdiv(D, TD) * bfm_cdiv(H, TH) * bfm_cdiv(W, TW);
    // flags, the first box of each of the 27 classes, the two counts, the two work lists (uniform_lists_kernel)
    return ((n + 3) & ~(size_t)3) + 27 * 4 + 2 * 4 + 2 * n * 4;
}

extern "C" int bfm_uniform_boxes_level(const float* image, int D, int H, int W, int level, int radius, int passes,
                                       unsigned char* flags, bfm_stream_t stream) {
    if (!image || !flags || D <= 0 || H <= 0 || W <= 0 || radius < 0 || radius > 64 || level < 0 || level > 4) return BFM_E_ARG;
    if (reinterpret_cast<uintptr_t>(flags) & 3) return BFM_E_ARG;
    const int d = D >> level, h = H >> level, w = W >> level;          // MaxPool3d(2) floors
    int TD, TH, TW;
    if (d <= 0 || h <= 0 || w <= 0 || !choose_box(d, h, w, passes == 3 ? 2 : 1, TD, TH, TW)) return BFM_E_SHAPE;
    const int nTz = bfm_cdiv(d, TD), nTy = bfm_cdiv(h, TH), nTx = bfm_cdiv(w, TW);
    const size_t n = (size_t)nTz * nTy * nTx;
    int* first = reinterpret_cast<int*>(flags + ((n + 3) & ~(size_t)3));
    // a class is well defined when a box side outreaches the radius at its level (else a middle box could see a face)
    if ((TD << level) < radius || (TH << level) < radius || (TW << level) < radius) return BFM_E_SHAPE;
    hipLaunchKernelGGL(uniform_boxes_kernel, dim3((unsigned)n), dim3(256), 0, bfm_s(stream), image, D, H, W, TD, TH, TW, nTz,
                       nTy, nTx, level, radius, flags);
    hipLaunchKernelGGL(uniform_lists_kernel, dim3(1), dim3(1024), 0, bfm_s(stream), flags, (int)n, first, first + 27,
                       first + 29, first + 29 + (int)n);
    return bfm_launch_status();
}

extern "C" int bfm_uniform_boxes(const float* image, int D, int H, int W, int radius, int passes, unsigned char* flags,
                                 bfm_stream_t stream) {
    return bfm_uniform_boxes_level(image, D, H, W, 0, radius, passes, flags, stream);
}

extern "C" size_t bfm_conv3x3x3_wino_uniform_scratch(int Cout) {
    return Cout > 0 && Cout % 64 == 0 ? (size_t)27 * (Cout / 64) * 2 * 16 * NTHR * 2 * sizeof(float) : 0;
}

extern "C" int bfm_conv3x3x3_wino_uniform(const float* A, int CA, int D, int H, int W, const float* scale,
                                          const float* shift, const float* bound, int G, const void* wpacked, int wexp,
                                          int Cout, float slope, int passes, int flags, float* out, void* moment_rows,
                                          const unsigned char* uniform_flags, void* scratch, bfm_stream_t stream) {
    if (!uniform_flags || !scratch || (flags & ~1)) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(uniform_flags) & 3) || (reinterpret_cast<uintptr_t>(scratch) & 15)) return BFM_E_ARG;
    return wino_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, moment_rows,
                       nullptr, stream, uniform_flags, static_cast<float*>(scratch));
}

extern "C" int bfm_conv3x3x3_wino_ex(const float* A, int CA, int D, int H, int W, const float* scale,
                                     const float* shift, const float* bound, int G, const void* wpacked, int wexp,
                                     int Cout, float slope, int passes, int flags, float* out, void* moment_rows,
                                     bfm_stream_t stream) {
    return wino_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, moment_rows,
                       nullptr, stream);
}

extern "C" size_t bfm_conv3x3x3_wino_masked_workspace(int D, int H, int W, int passes) {
    int TD, TH, TW;
    if (D <= 0 || H <= 0 || W <= 0 || !choose_box(D, H, W, passes == 3 ? 2 : 1, TD, TH, TW)) return 0;
    const size_t n = (size_t)bfm_cdiv(D, TD) * bfm_cdiv(H, TH) * bfm_cdiv(W, TW);
    return ((n + 3) & ~(size_t)3) + 4 + n * 4;                 // box activity bytes, the count, the list
}

extern "C" int bfm_conv3x3x3_wino_masked(const float* A, int CA, int D, int H, int W, const float* scale,
                                         const float* shift, const float* bound, int G, const void* wpacked, int wexp,
                                         int Cout, float slope, int passes, int flags, float* out,
                                         const float* mask_image, void* workspace, size_t workspace_bytes,
                                         bfm_stream_t stream) {
    if (!mask_image || (flags & ~1)) return BFM_E_ARG;         // no moment rows (boxes are left out)
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 3) ||
        workspace_bytes < bfm_conv3x3x3_wino_masked_workspace(D, H, W, passes))
        return BFM_E_ARG;
    return wino_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, nullptr,
                       mask_image, stream, nullptr, nullptr, workspace);
}

static int wino_launch(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                       const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes, int flags,
                       float* out, void* moment_rows, const float* mask_img, bfm_stream_t stream,
                       const unsigned char* uni_flags, float* uni_acc, void* mask_ws, float* pool_out, void* pool_rows) {
    const int accumulate = flags & 1;
    if (flags & ~1) return BFM_E_ARG;                           // bit 0 = accumulate; nothing else is defined
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
    p.mask_img = mask_img;
    p.uni_flags = uni_flags;
    p.uni_acc = uni_acc;
    if (!choose_box(D, H, W, npl, p.TD, p.TH, p.TW)) return BFM_E_SHAPE;
    p.HT = p.TH + 2; p.PW = p.TW / 2;
    p.pw_shift = ilog2i(p.PW); p.thp_shift = ilog2i(p.TH * p.PW);
    const int nTz = bfm_cdiv(D, p.TD);
    p.nTy = bfm_cdiv(H, p.TH); p.nTx = bfm_cdiv(W, p.TW);
    p.nMt = nTz * p.nTy * p.nTx;
    p.uni_first = uni_flags ? reinterpret_cast<const int*>(uni_flags + (((size_t)p.nMt + 3) & ~(size_t)3)) : nullptr;
    p.NT = Cout / 64;
    p.KCN = CA / KC;
    p.npos_lds = (p.TD + 2) * p.HT * p.PW;
    p.plane_stride = ((p.npos_lds * 16 + 255) / 256) * 256 + 16;
    size_t smem = (size_t)8 * npl * p.plane_stride;
    const size_t epi = (size_t)4 * 128 * MLD * sizeof(float) + 6144;    // output-transform scratch + moment-row fold
    if (smem < epi) smem = epi;
    if (moment_rows) {
        if (reinterpret_cast<uintptr_t>(moment_rows) & 7) return BFM_E_ARG;
        char* rb = static_cast<char*>(moment_rows);
        const size_t n = (size_t)p.nMt * Cout;
        p.rsum = reinterpret_cast<double*>(rb);
        p.rsq = reinterpret_cast<double*>(rb + n * 8);
        p.rmn = reinterpret_cast<float*>(rb + n * 16);
        p.rmx = reinterpret_cast<float*>(rb + n * 20);
    }
    if (pool_out) {                                            // the POOL kernels: box 8 x 8 x 4, every box inside the tensor
        if (mask_img || !pool_ok_box(p.TD, p.TH, p.TW, D, H, W)) return BFM_E_SHAPE;
        if (reinterpret_cast<uintptr_t>(pool_out) & 15) return BFM_E_ARG;
        p.pool_out = pool_out;
        if (pool_rows) {
            if (reinterpret_cast<uintptr_t>(pool_rows) & 7) return BFM_E_ARG;
            char* rb = static_cast<char*>(pool_rows);
            const size_t n = (size_t)p.nMt * Cout;
            p.prsum = reinterpret_cast<double*>(rb);
            p.prsq = reinterpret_cast<double*>(rb + n * 8);
            p.prmn = reinterpret_cast<float*>(rb + n * 16);
            p.prmx = reinterpret_cast<float*>(rb + n * 20);
        }
    } else if (pool_rows) return BFM_E_ARG;
    if ((int64_t)p.nMt * p.NT > 0x7fffffff) return BFM_E_SHAPE;
    if (smem > 80 * 1024) return BFM_E_SHAPE;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_pool<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_pool<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_rest_pool<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_rest_pool<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_masked<3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_masked<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_uniform<3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_uniform<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_rest<3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_rest<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_done = true;
    }
    dim3 grid((unsigned)(p.nMt * p.NT));
    // the sparse forms take their boxes from a list built on the device; workgroups beyond the list end at once
    if (mask_img) {
        if (!mask_ws) return BFM_E_ARG;
        unsigned char* act = static_cast<unsigned char*>(mask_ws);
        int* cnt = reinterpret_cast<int*>(act + (((size_t)p.nMt + 3) & ~(size_t)3));
        hipLaunchKernelGGL(wino_box_active_kernel, dim3((unsigned)bfm_cdiv(p.nMt, 4)), dim3(256), 0, bfm_s(stream), mask_img,
                           D, H, W, p.TD, p.TH, p.TW, p.nTy, p.nTx, p.nMt, act);
        hipLaunchKernelGGL(wino_mask_list_kernel, dim3(1), dim3(1024), 0, bfm_s(stream), act, p.nMt, cnt + 1, cnt);
        p.list = cnt + 1; p.list_n = cnt;
        if (passes == 3) hipLaunchKernelGGL(conv_wino_masked<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        else hipLaunchKernelGGL(conv_wino_masked<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        return bfm_launch_status();
    }
    if (uni_flags && pool_out) {
        const int* cnt = p.uni_first + 27;
        p.list = cnt + 2; p.list_n = cnt;
        if (passes == 3) hipLaunchKernelGGL(conv_wino_rest_pool<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        else hipLaunchKernelGGL(conv_wino_rest_pool<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        p.list = cnt + 2 + p.nMt; p.list_n = cnt + 1;
        if (passes == 3) hipLaunchKernelGGL(conv_wino_uniform_pool<3>, grid, dim3(NTHR), 20480, bfm_s(stream), p);
        else hipLaunchKernelGGL(conv_wino_uniform_pool<1>, grid, dim3(NTHR), 20480, bfm_s(stream), p);
        return bfm_launch_status();
    }
    if (pool_out) {
        if (passes == 3) hipLaunchKernelGGL(conv_wino_pool<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        else hipLaunchKernelGGL(conv_wino_pool<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        return bfm_launch_status();
    }
    if (uni_flags) {                                           // disjoint boxes: the two launches may overlap
        const int* cnt = p.uni_first + 27;                     // uniform_lists_kernel: the two counts, then the two lists
        p.list = cnt + 2; p.list_n = cnt;
        if (passes == 3) hipLaunchKernelGGL(conv_wino_rest<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        else hipLaunchKernelGGL(conv_wino_rest<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
        p.list = cnt + 2 + p.nMt; p.list_n = cnt + 1;
        if (passes == 3) hipLaunchKernelGGL(conv_wino_uniform<3>, grid, dim3(NTHR), 6144, bfm_s(stream), p);
        else hipLaunchKernelGGL(conv_wino_uniform<1>, grid, dim3(NTHR), 6144, bfm_s(stream), p);
        return bfm_launch_status();
    }
    if (passes == 3) hipLaunchKernelGGL(conv_wino<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    else hipLaunchKernelGGL(conv_wino<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    return bfm_launch_status();
}
