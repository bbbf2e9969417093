// Fused network tail: one pass over the last decoder feature map.
//
//   F.normalize(feat, dim=1)                      unet3d/model.py:207-208
//   TaskHead: n_out x c_feat 1x1x1 conv + bias    head.py:52-59
//   SegProcessor softmax, DistProcessor clamp     joiner.py:69-77,149-157
//   get_postprocessor                             Trainer/models/__init__.py:272-354
//
// The reference reads the 64-channel feature map once per head (9x) plus once
// for normalize; here each voxel's 64 features are read once, and only what the
// caller asks for is written (15 fp32 maps + int64 label = 68 B/voxel; the
// 56-channel softmax and the normalised features are optional).
//
// One wave owns 64 voxels at a time and never synchronises with another wave.  Each lane pulls its half of a
// voxel's feature row straight into registers in the layout v_mfma_f32_32x32x2_f32 wants for A (the K order of a
// dot product is free, so lane (row, half) simply takes 32 consecutive channels), normalises it with one
// cross-half shuffle, and the head GEMM runs on the fp32 matrix core (bit-for-bit an fp32 FMA chain) against
// weights held in registers as B fragments.  The logits go to a wave-private LDS slab [64][n_out|1] (odd stride:
// conflict-free), from which one lane per voxel applies softmax / argmax / exp / tanh / clamp and writes the maps
// coalesced.  raw_out != NULL turns the call into TaskHead.forward alone (raw logits, [nvox][n_out]).
#include "bfm_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8t __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

constexpr int WPB = 4;       // waves per block (independent)
constexpr int TPBT = 64 * WPB;
constexpr int CMAX = 64;     // c_feat upper bound
constexpr int OMAX = 96;     // n_out upper bound (3 MFMA column blocks)
constexpr int NMAPS = 64;    // upper bound on output maps

struct TailParams {
    const float* feat;
    const float* input;
    int64_t nvox;
    bfm_tail_desc_t d;
    float* feat_norm;
    float* const* maps;
    float* maps_rows;       // maps == NULL: map i is maps_rows + i * row_stride
    int64_t row_stride;
    float* seg_prob;
    int64_t* label;
    float* raw_out;
    int n_maps;
    int64_t raw_stride;     // > 0: raw_out as [n_out] rows of this pitch (the training losses' layout), else [nvox][n_out]
};

// exp on the hardware exp2 unit: |rel err| ~ 1e-6 for the argument ranges below (softmax <= 0, tanh via exp)
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = fast_exp(2.f * fminf(fmaxf(x, -15.f), 15.f));
    return (e - 1.f) * __builtin_amdgcn_rcpf(e + 1.f);
}

__device__ __forceinline__ float fake_term(float w_or_p, float add, float gain) {
    // gain * (1 - (tanh(2*(v+add)) + 1) / 2)      (__init__.py:329-336, a = 2)
    return gain * (1.f - (fast_tanh(2.f * (w_or_p + add)) + 1.f) / 2.f);
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving accesses across
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// SPLIT (used when unit_feat: the normalised row is bounded by 1, so one power-of-two scale fits every voxel): the
// head GEMM runs as three v_mfma_f32_32x32x16_f16 passes on hi/lo halves (a_lo*w_hi + a_hi*w_lo + a_hi*w_hi, fp32
// accumulate: ~2^-22 relative per product, the same scheme as conv_mfma) -- 5x fewer matrix-core cycles than the
// fp32 MFMA chain, which is kept for unnormalised features whose range is unknown.
// NS > 0: the segmentation head has exactly NS channels and the heads fill NNB 32-wide column blocks -- compile-time
// bounds for the softmax / argmax and MFMA loops of the shipped head sets (56 and 18 classes); NS = 0: runtime bounds
// (any head set; every s < n_seg test is then a scalar branch).
template <bool SPLIT, int NS = 0, int NNB = 0>
__global__ void __launch_bounds__(TPBT, 2) tail_kernel(TailParams p, int ld, int wexp) {
    extern __shared__ float smem[];                 // [WPB][64][ld] logits, then the per-head tables
    int* s_role = reinterpret_cast<int*>(smem + WPB * 64 * ld);        // [OMAX]
    int* s_slot = s_role + OMAX;                                       // [OMAX]
    int* s_plain = s_slot + OMAX;                                      // [OMAX] non-segmentation outputs, compacted
    int* s_lut = s_plain + OMAX;                                       // [OMAX]
    float** s_map = reinterpret_cast<float**>(s_lut + OMAX);           // [NMAPS]
    __shared__ int s_nplain, s_srfirst;
    const int C = p.d.c_feat;
    const int NO = p.d.n_out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l32 = lane & 31, lh = lane >> 5;
    const int half = C >> 1;                        // channels per lane (K of the MFMA chain per half)
    const int nnb = NNB > 0 ? NNB : (NO + 31) >> 5; // 32-wide output column blocks

    // ---- head weights as B fragments.  fp32 chain: B[k = lh][n = l32] of step kk, block nb = W[nb*32+l32][lh*half+kk];
    // split: k-step ks (16 wide), lane slot 8*lh + j  <->  channel lh*half + 8*ks + j  (the same channel order as A)
    float wfrag[SPLIT ? 1 : 3][SPLIT ? 1 : CMAX / 2];
    half8t whi[SPLIT ? 3 : 1][SPLIT ? CMAX / 16 : 1], wlo[SPLIT ? 3 : 1][SPLIT ? CMAX / 16 : 1];
    float bias[3];
    const float wscale = ldexpf(1.f, wexp);
    const float ascale = 16384.f;                   // |normalised feature| <= 1
    const float dq = ldexpf(1.f, -(wexp + 14));
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int o = nb * 32 + l32;
        bias[nb] = (nb < nnb && o < NO) ? p.d.head_b[o] : 0.f;
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < CMAX / 16; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = 8 * ks + j;
                    const float w = (nb < nnb && o < NO && kk < half) ? p.d.head_w[(size_t)o * C + lh * half + kk] * wscale : 0.f;
                    const _Float16 h = (_Float16)w;
                    whi[nb][ks][j] = h;
                    wlo[nb][ks][j] = (_Float16)(w - (float)h);
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < CMAX / 2; ++kk)
                wfrag[nb][kk] = (nb < nnb && o < NO && kk < half) ? p.d.head_w[(size_t)o * C + lh * half + kk] : 0.f;
        }
    }
    for (int i = threadIdx.x; i < NO; i += TPBT) { s_role[i] = p.d.roles[i]; s_slot[i] = p.d.out_slot[i]; }
    for (int i = threadIdx.x; i < p.n_maps; i += TPBT) s_map[i] = p.maps ? p.maps[i] : p.maps_rows + (int64_t)i * p.row_stride;
    for (int i = threadIdx.x; i < p.d.n_seg; i += TPBT) s_lut[i] = p.d.seg_lut[i];
    if (threadIdx.x == 0) {
        int n = 0, sr = -1;
        for (int o = 0; o < NO; ++o) {
            if (p.d.roles[o] != BFM_ROLE_SEG) s_plain[n++] = o;
            if (p.d.roles[o] == BFM_ROLE_SR && sr < 0) sr = o;
        }
        s_nplain = n;
        s_srfirst = sr < 0 ? 0 : sr;
    }
    __syncthreads();
    const int nplain = s_nplain;
    const int sr_first = s_srfirst;                 // rows with BFM_ROLE_SR are one head's channels: contiguous
    float* slab = smem + wave * 64 * ld;            // this wave's [64][ld] logits
    float* row = slab + lane * ld;

    // wave-uniform tables of the first PMAX non-segmentation outputs: head row, role, destination map (scalar registers)
    constexpr int PMAX = 16;
    int po[PMAX], prole[PMAX];
    float* pptr[PMAX];
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
        int o = 0, role = BFM_ROLE_PLAIN;
        unsigned long long u = 0;
        if (j < nplain) {
            o = s_plain[j];
            role = s_role[o];
            const int slot = s_slot[o];
            if (slot >= 0) u = reinterpret_cast<unsigned long long>(s_map[slot]);
        }
        po[j] = __builtin_amdgcn_readfirstlane(o);
        prole[j] = __builtin_amdgcn_readfirstlane(role);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        pptr[j] = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    }

    const int64_t nchunks = (p.nvox + 63) >> 6;
    const int64_t cstride = (int64_t)gridDim.x * WPB;
    const bool gate = p.d.skip_zero_input && p.input;
    // chunks in groups of four: the input values of a group (the skip test, and the SR head's addend) are requested
    // together, so a run of skipped chunks costs one memory round trip per four instead of one each
    for (int64_t cbase = (int64_t)blockIdx.x * WPB + wave; cbase < nchunks; cbase += 4 * cstride) {
      float ivq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
          const int64_t c = cbase + k * cstride;
          ivq[k] = (p.input && c < nchunks && (c << 6) + lane < p.nvox) ? p.input[(c << 6) + lane] : 0.f;
      }
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        const int64_t cix = cbase + k * cstride;
        if (cix >= nchunks) break;
        const int64_t v0 = cix << 6;
        const int nv = (int)min<int64_t>(64, p.nvox - v0);
        const float iv = k == 0 ? ivq[0] : (k == 1 ? ivq[1] : (k == 2 ? ivq[2] : ivq[3]));
        // the caller keeps this chunk's outputs only where the input image is non-zero (the tile loop's mask,
        // scripts/demo_test.py:88-100): nothing of a chunk of 64 zero voxels is looked at
        if (gate && !__any(iv != 0.f)) continue;
#pragma unroll 1
        for (int mb = 0; mb < 2; ++mb) {
            const int r = mb * 32 + l32;            // this lane's voxel row inside the chunk
            const bool rlive = r < nv;
            float av[CMAX / 2];
            const float4* src = reinterpret_cast<const float4*>(p.feat + (v0 + (rlive ? r : 0)) * C + lh * half);
#pragma unroll
            for (int j = 0; j < CMAX / 8; ++j) {
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (4 * j < half) q = src[j];       // a row past the end reads the chunk's first voxel: rows do not mix
                av[4 * j] = q.x; av[4 * j + 1] = q.y; av[4 * j + 2] = q.z; av[4 * j + 3] = q.w;
            }
            float inv = 1.f;                        // applied where the row is consumed (one multiply less per channel)
            if (p.d.unit_feat) {
                float ss = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX / 2; ++c) ss = fmaf(av[c], av[c], ss);
                ss += __shfl_xor(ss, 32);
                inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);               // F.normalize eps; one division per voxel
            }
            if (p.feat_norm && rlive) {               // optional: normalised features
                float4* dst = reinterpret_cast<float4*>(p.feat_norm + (v0 + r) * C + lh * half);
#pragma unroll
                for (int j = 0; j < CMAX / 8; ++j)
                    if (4 * j < half)
                        dst[j] = make_float4(av[4 * j] * inv, av[4 * j + 1] * inv, av[4 * j + 2] * inv, av[4 * j + 3] * inv);
            }
            if (NO > 0) {
                floatx16 acc[3];
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
                if constexpr (SPLIT) {
                    const float sinv = inv * ascale;
#pragma unroll
                    for (int ks = 0; ks < CMAX / 16; ++ks) {
                        if (8 * ks < half) {
                            // a = (x * inv) * 2^14 = x * (inv * 2^14) bit for bit; hi by truncation (v_cvt_pkrtz: two
                            // values per instruction, a - hi exact in fp32), lo likewise -- conv3d_wino.hip's split
                            unsigned hw[4], lw[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float a0 = av[8 * ks + 2 * j] * sinv, a1 = av[8 * ks + 2 * j + 1] * sinv;
                                const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(a0, a1);
                                const fp16x2_t l = __builtin_amdgcn_cvt_pkrtz(a0 - (float)h[0], a1 - (float)h[1]);
                                hw[j] = __builtin_bit_cast(unsigned, h);
                                lw[j] = __builtin_bit_cast(unsigned, l);
                            }
                            const half8t ahi = __builtin_bit_cast(half8t, make_uint4(hw[0], hw[1], hw[2], hw[3]));
                            const half8t alo = __builtin_bit_cast(half8t, make_uint4(lw[0], lw[1], lw[2], lw[3]));
#pragma unroll
                            for (int nb = 0; nb < 3; ++nb)
                                if (nb < nnb) {
                                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, whi[nb][ks], acc[nb], 0, 0, 0);
                                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, wlo[nb][ks], acc[nb], 0, 0, 0);
                                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, whi[nb][ks], acc[nb], 0, 0, 0);
                                }
                        }
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < CMAX / 2; ++kk) {
                        if (kk < half) {
                            const float a = av[kk] * inv;
#pragma unroll
                            for (int nb = 0; nb < 3; ++nb)
                                if (nb < nnb)
                                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wfrag[nb][kk], acc[nb], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    if (nb < nnb) {
                        const int o = nb * 32 + l32;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int rr = (i & 3) + 8 * (i >> 2) + 4 * lh;
                            // split: acc * dq is exact (a power of two), so the fused form rounds once, as mul + add did
                            if (o < NO) slab[(mb * 32 + rr) * ld + o] = SPLIT ? fmaf(acc[nb][i], dq, bias[nb]) : acc[nb][i] + bias[nb];
                        }
                    }
                }
            }
        }
        wave_lds_sync();

        if (p.raw_out) {                                      // TaskHead.forward only: raw logits, channels-last
            if (p.raw_stride > 0) {                           // one 256-byte row segment per output
                if (lane < nv) {
                    float* dst = p.raw_out + v0 + lane;
                    for (int o = 0; o < NO; ++o) dst[(int64_t)o * p.raw_stride] = row[o];
                }
            } else {
                const int n = nv * NO;
                float* dst = p.raw_out + v0 * NO;
                for (int i = lane; i < n; i += 64) {
                    int r = i / NO, o = i - r * NO;
                    dst[i] = slab[r * ld + o];
                }
            }
            wave_lds_sync();
            continue;
        }

        const bool live = lane < nv;
        const int64_t v = v0 + lane;
        float dist[4] = {0.f, 0.f, 0.f, 0.f};
        // ---- processors + post-processor per role (segmentation rows are handled below).  The first PMAX outputs run
        // from the wave-uniform tables hoisted out of the chunk loop (scalar registers: the logit reads of all outputs are
        // independent LDS loads, the stores take a scalar base); any further ones from the LDS tables
        auto plain_out = [&](int o, int role, float* mp) __attribute__((always_inline)) {
            const float a = row[o];
            float r = a;
            if (role == BFM_ROLE_CT) r = a * 1000.f;
            else if (role == BFM_ROLE_BIAS_LOG) r = expf(a);
            else if (role == BFM_ROLE_PATHOL) r = 1.f / (1.f + expf(-a));
            else if (role == BFM_ROLE_DIST) {
                r = fminf(fmaxf(a, -p.d.max_dist), p.d.max_dist);
                const int k = o - p.d.dist_first;
                if (k == 0) dist[0] = r; else if (k == 1) dist[1] = r; else if (k == 2) dist[2] = r; else dist[3] = r;
            }
            if (live && mp) mp[v] = r;
            if (role == BFM_ROLE_SR && p.d.slot_high_res >= 0 && live && p.input)      // channel c -> slot_high_res + c
                s_map[min(p.d.slot_high_res + (o - sr_first), p.n_maps - 1)][v] = a + iv;
        };
#pragma unroll
        for (int j = 0; j < PMAX; ++j)
            if (j < nplain) plain_out(po[j], prole[j], pptr[j]);
        for (int j = PMAX; j < nplain; ++j) {
            const int o = s_plain[j];
            const int slot = s_slot[o];
            plain_out(o, s_role[o], slot >= 0 ? s_map[slot] : nullptr);
        }

        if (p.d.n_dist > 0 && p.d.slot_fake_cortical >= 0 && live) {
            // order lp, lw[, rp, rw]  (__init__.py:321-337)
            float fake = fake_term(dist[1], 0.3f, 70.f) + fake_term(dist[0], 0.f, 40.f);
            if (p.d.n_dist == 4) fake = fake + (fake_term(dist[3], 0.3f, 70.f) + fake_term(dist[2], 0.f, 40.f));
            s_map[p.d.slot_fake_cortical][v] = fake;
        }

        if (p.d.n_seg > 0) {
            const int ns = NS > 0 ? NS : p.d.n_seg;
            float* sl = row + p.d.seg_first;
            if (live && ns <= 64) {
                // the whole logit row in registers: one batch of LDS reads, then max / exp / sum / argmax without
                // a load in any dependence chain (padded entries are -inf -> exp 0, never the maximum)
                constexpr int LIM = NS > 0 ? NS : 64;              // compile-time class count: no padded entries, no branches
                float sv[LIM];
#pragma unroll
                for (int s = 0; s < LIM; ++s) sv[s] = (NS > 0 || s < ns) ? sl[s] : -INFINITY;
                // largest and second largest logit.  The label is argmax over the softmax PROBABILITIES (first maximum,
                // __init__.py:347-349); when nobody asks for the probabilities and the top logit leads by more than
                // ARGMAX_GAP, its probability exp2(0) * rs = rs exceeds every other exp2(-gap * log2 e) * rs by >= 160 ulps
                // (v_exp_f32 is good to 1 ulp, the multiply by rs is monotone), so the 56 exponentials are skipped.  A closer
                // call anywhere in the wave takes the full softmax for the whole wave.
                constexpr float ARGMAX_GAP = 1e-5f;
                float m = -INFINITY, m2 = -INFINITY, chk = 0.f;
                int best = 0;
#pragma unroll
                for (int s = 0; s < LIM; ++s) {
                    const float x = sv[s];
                    chk += x;                                      // NaN / +inf / mixed infinities end up non-finite here
                    if (x > m) { m2 = m; m = x; best = s; } else m2 = fmaxf(m2, x);
                }
                const bool sure = p.seg_prob == nullptr && (m - m2) > ARGMAX_GAP && fabsf(chk) < INFINITY;
                if (__any(!sure)) {
                    float sum = 0.f;
#pragma unroll
                    for (int s = 0; s < LIM; ++s) { sv[s] = fast_exp(sv[s] - m); sum += sv[s]; }
                    const float rs = 1.f / sum;
                    float bp = -1.f;
                    best = 0;
#pragma unroll
                    for (int s = 0; s < LIM; ++s) {
                        sv[s] = sv[s] * rs;
                        if (sv[s] > bp) { bp = sv[s]; best = s; }    // first maximum wins (torch.argmax)
                    }
                    if (p.seg_prob) {
#pragma unroll
                        for (int s = 0; s < LIM; ++s)
                            if (NS > 0 || s < ns) sl[s] = sv[s];
                    }
                }
                if (p.label) p.label[v] = (int64_t)s_lut[best];
            } else if (live) {
                int best = 0;
                float m = -INFINITY;
                for (int s = 0; s < ns; ++s) m = fmaxf(m, sl[s]);
                float sum = 0.f;
                for (int s = 0; s < ns; ++s) { float e = fast_exp(sl[s] - m); sl[s] = e; sum += e; }
                const float rs = 1.f / sum;
                float bp = -1.f;
                for (int s = 0; s < ns; ++s) {
                    float pr = sl[s] * rs;
                    sl[s] = pr;
                    if (pr > bp) { bp = pr; best = s; }      // first maximum wins (torch.argmax)
                }
                if (p.label) p.label[v] = (int64_t)s_lut[best];
            }
            if (p.seg_prob) {
                wave_lds_sync();
                const int n = nv * ns;
                float* dst = p.seg_prob + v0 * ns;
                for (int i = lane; i < n; i += 64) {
                    int r = i / ns, s2 = i - r * ns;
                    dst[i] = slab[r * ld + p.d.seg_first + s2];
                }
            }
        }
        wave_lds_sync();                                      // slab fully consumed before the next chunk's logits
      }
    }
}

}  // namespace

static int tail_launch(const float* feat, const float* input, int64_t nvox, const bfm_tail_desc_t* desc,
                       float* feat_norm, float* const* maps_tab, float* maps_rows, int64_t row_stride, float* seg_prob,
                       int64_t* label, float* raw_out, int skip_zero_input, bfm_stream_t stream, int64_t raw_stride = 0) {
    if (!feat || !desc || nvox <= 0 || !desc->head_w || !desc->head_b || !desc->roles || !desc->out_slot)
        return BFM_E_ARG;
    const bool maps = maps_tab != nullptr || maps_rows != nullptr;
    if (maps_rows && (row_stride < nvox || (reinterpret_cast<uintptr_t>(maps_rows) & 3))) return BFM_E_ARG;
    if (!maps && !raw_out && !feat_norm) return BFM_E_ARG;
    if (desc->n_out < 0 || desc->n_out > OMAX || (desc->n_out > 0 && !maps && !raw_out)) return BFM_E_SHAPE;
    if (desc->n_seg > 0 && (desc->seg_first < 0 || desc->seg_first + desc->n_seg > desc->n_out)) return BFM_E_SHAPE;
    if (desc->c_feat <= 0 || desc->c_feat > CMAX || desc->c_feat % 8 != 0) return BFM_E_SHAPE;
    if (desc->n_seg < 0 || (desc->n_seg > 0 && !desc->seg_lut)) return BFM_E_SHAPE;
    if (desc->n_dist != 0 && desc->n_dist != 2 && desc->n_dist != 4) return BFM_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(feat) & 15 || (feat_norm && (reinterpret_cast<uintptr_t>(feat_norm) & 15)))
        return BFM_E_ARG;
    int n_maps = maps ? desc->n_maps : 0;
    if (n_maps < 0 || n_maps > NMAPS) return BFM_E_SHAPE;
    for (int o = 0; o < desc->n_out; ++o) (void)o;
    if (maps && (desc->slot_high_res >= n_maps || desc->slot_fake_cortical >= n_maps)) return BFM_E_SHAPE;
    TailParams p{feat, input, nvox, *desc, feat_norm, maps_tab, maps_tab ? nullptr : maps_rows, row_stride, seg_prob, label,
                 raw_out, n_maps, raw_stride};
    if (skip_zero_input) p.d.skip_zero_input = 1;
    const int ld = (desc->n_out > 0 ? desc->n_out : 1) | 1;        // odd row stride: conflict-free column access
    int64_t nb = bfm_cdiv64(bfm_cdiv64(nvox, 64), WPB);
    if (nb > 256 * 2) nb = 256 * 2;                  // persistent: the weight fragments are loaded once per wave
    const size_t smem = (size_t)WPB * 64 * ld * sizeof(float) + (size_t)4 * OMAX * sizeof(int) +
                        (size_t)NMAPS * sizeof(float*);
    if (smem > 160 * 1024) return BFM_E_SHAPE;
    if (smem > 64 * 1024 &&
        (hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
         hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<true, 56, 3>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
         hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<true, 18, 1>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
         hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess))
        return BFM_E_LAUNCH;
    // split-f16 path: weights scaled so that max|w| * 2^wexp < 2^15 (head_wmax from the descriptor)
    int wexp = 0;
    const bool split = desc->unit_feat && desc->head_wmax > 0.f && desc->head_wmax < INFINITY && desc->c_feat % 16 == 0;
    if (split) {
        int ex;
        (void)frexpf(desc->head_wmax, &ex);
        wexp = 14 - ex;
        wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp);
    }
    const int nnb = (desc->n_out + 31) >> 5;
    if (split && desc->n_seg == 56 && nnb == 3)                // the shipped head set (69 outputs, 56 classes)
        hipLaunchKernelGGL((tail_kernel<true, 56, 3>), dim3((unsigned)nb), dim3(TPBT), smem, bfm_s(stream), p, ld, wexp);
    else if (split && desc->n_seg == 18 && nnb == 1)           // left-hemisphere head set (18 classes, 27 outputs)
        hipLaunchKernelGGL((tail_kernel<true, 18, 1>), dim3((unsigned)nb), dim3(TPBT), smem, bfm_s(stream), p, ld, wexp);
    else if (split)
        hipLaunchKernelGGL((tail_kernel<true>), dim3((unsigned)nb), dim3(TPBT), smem, bfm_s(stream), p, ld, wexp);
    else
        hipLaunchKernelGGL((tail_kernel<false>), dim3((unsigned)nb), dim3(TPBT), smem, bfm_s(stream), p, ld, 0);
    return bfm_launch_status();
}

extern "C" int bfm_tail_heads(const float* feat, const float* input, int64_t nvox, const bfm_tail_desc_t* desc,
                              float* feat_norm, float* const* maps, float* seg_prob, int64_t* label,
                              float* raw_out, bfm_stream_t stream) {
    return tail_launch(feat, input, nvox, desc, feat_norm, maps, nullptr, 0, seg_prob, label, raw_out, 0, stream);
}

extern "C" int bfm_tail_heads_rows(const float* feat, const float* input, int64_t nvox, const bfm_tail_desc_t* desc,
                                   float* feat_norm, float* maps_rows, int64_t row_stride, float* seg_prob,
                                   int64_t* label, int flags, bfm_stream_t stream) {
    if (!maps_rows || (flags & ~1)) return BFM_E_ARG;
    return tail_launch(feat, input, nvox, desc, feat_norm, nullptr, maps_rows, row_stride, seg_prob, label, nullptr,
                       flags & 1, stream);
}

// TaskHead.forward alone with the logits as [n_out] rows of nvox values (row pitch row_stride): the layout the training
// losses and bfm_head_bwd_rows walk (train_heads.hip).  Same arithmetic as bfm_tail_heads(..., raw_out).
extern "C" int bfm_tail_raw_rows(const float* feat, int64_t nvox, const bfm_tail_desc_t* desc, float* feat_norm,
                                 float* raw_rows, int64_t row_stride, bfm_stream_t stream) {
    if (!raw_rows || row_stride < nvox) return BFM_E_ARG;
    return tail_launch(feat, nullptr, nvox, desc, feat_norm, nullptr, nullptr, 0, nullptr, nullptr, raw_rows, 0, stream,
                       row_stride);
}
