// Gather / resample kernels of the synthesis path (all HBM-bound, one thread per output voxel,
// lanes along the fastest axis so coordinate reads and result writes are coalesced).
//
//   interp3d_linear / nearest : fast_3D_interp_torch          Generator/utils.py:119-196
//   zoom_linear               : myzoom_torch (3 passes fused)  Generator/utils.py:200-257
//   conv1d_axis               : gaussian_blur_3d (one axis)    Generator/utils.py:74-94
//   grid_pull3d_linear        : interpol iso1.pull3d           utils/interpol/iso1.py:28-133, bounds.py:24-89
//   deform_grid               : BaseGen.deform_grid            Generator/datasets.py:264-303
//   label_gauss / onehot_lut  : generate_sample / seg targets  Generator/datasets.py:366-372, utils.py:408-411
//
// Built with -ffp-contract=off: a*b + c*d stays mul, mul, add like the reference's eager torch ops,
// which is what makes the fp32 results bit-identical to the CPU path.
#include "bfm_common.h"
#include <cstdlib>

namespace {

inline int grid_for(int64_t n, int tpb = 256, int cap = 8192) {
    int64_t b = bfm_cdiv64(n, tpb);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// Texel loads of the gathers (data-dependent addresses, lines re-used from the L1 by neighbouring lanes and waves) go past
// the per-CU vector L1: agent scope = global_load_dword sc1, served by the XCD's L2.  Round 3 cornered what round 2 had
// only worked around (HISTORY.md section 3.3, tests/diag/diag_atlas_repro.py, profiles/r03_atlas_gather_hazard.txt): with
// ordinary loads such a gather gets wrong texels -- whole 16-lane groups -- whenever a kernel that fills its LDS by LDS-DMA
// (global_load_lds, every conv kernel here) runs beside it on another stream; an L1 invalidate at kernel start does not
// help, L1-bypassing loads (agent or system scope) do.  The value type is float or a 4-byte bit pattern.
__device__ __forceinline__ float ld_tex(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_tex(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// C == 1.  MODE 0 ships: the two z-neighbours of a corner in ONE 8-byte load through a buffer descriptor (aux 16 = sc1), four
// loads per sample instead of eight (20 against 25 us at 160^3).  MODE 1: the same with sc0 sc1; MODE 2: eight 4-byte
// agent-scope loads (what round 5 shipped); MODE 3: an 8-byte global load at system scope.  Modes other than 0 are reachable
// in -DBFM_DIAG builds only (BFM_INTERP_MODE).
typedef int v2i_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void interp_linear1(const float* __restrict__ X, int nx, int ny, int nz, uint32_t vol_bytes,
                               const float* __restrict__ II, const float* __restrict__ JJ,
                               const float* __restrict__ KK, int64_t n, float defv, float* __restrict__ out) {
    const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, vol_bytes, 0x00020000);
    GRID_STRIDE(i, n) {
        const float x = II[i], y = JJ[i], z = KK[i];
        const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= (float)(nx - 1)) && (y <= (float)(ny - 1)) &&
                        (z <= (float)(nz - 1));
        if (!ok) { out[i] = defv; continue; }
        const float fxf = floorf(x), fyf = floorf(y), fzf = floorf(z);
        const int fx = (int)fxf, fy = (int)fyf, fz = (int)fzf;
        const int cx = min(fx + 1, nx - 1), cy = min(fy + 1, ny - 1), cz = min(fz + 1, nz - 1);
        const float wcx = x - fxf, wcy = y - fyf, wcz = z - fzf;
        const float wfx = 1.f - wcx, wfy = 1.f - wcy, wfz = 1.f - wcz;
        const int64_t sx = (int64_t)ny * nz, sy = nz;
        const bool zp = cz != fz;
        auto pair = [&](int a, int b, float& lo, float& hi) {
            if constexpr (MODE == 0 || MODE == 1) {
                const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(R, (uint32_t)((a * sx + b * sy + fz) << 2), 0, MODE == 0 ? 16 : 17);
                lo = __int_as_float(v.x);
                hi = zp ? __int_as_float(v.y) : lo;
            } else if constexpr (MODE == 2) {
                const float* q = X + (a * sx + b * sy + fz);
                lo = ld_tex(q);
                hi = zp ? ld_tex(q + 1) : lo;
            } else {
                // one 8-byte global load at system scope (a volatile access: global_load_dwordx2 sc0 sc1), the pair moved one
                // back where the upper corner is clamped so that nothing past the row is read
                const float* q = X + (a * sx + b * sy + (zp ? fz : max(fz - 1, 0)));
                typedef v2i_t __attribute__((aligned(4))) v2i_a4;
                const v2i_t v = *(const volatile __attribute__((address_space(1))) v2i_a4*)q;
                lo = zp ? __int_as_float(v.x) : (fz >= 1 ? __int_as_float(v.y) : __int_as_float(v.x));
                hi = zp ? __int_as_float(v.y) : lo;
            }
        };
        float t000, t001, t100, t101, t010, t011, t110, t111;
        pair(fx, fy, t000, t001); pair(cx, fy, t100, t101); pair(fx, cy, t010, t011); pair(cx, cy, t110, t111);
        const float c00 = t000 * wfx + t100 * wcx;
        const float c01 = t001 * wfx + t101 * wcx;
        const float c10 = t010 * wfx + t110 * wcx;
        const float c11 = t011 * wfx + t111 * wcx;
        const float c0 = c00 * wfy + c10 * wcy;
        const float c1 = c01 * wfy + c11 * wcy;
        out[i] = c0 * wfz + c1 * wcz;
    }
}

__global__ void interp_linear(const float* __restrict__ X, int nx, int ny, int nz, int C,
                              const float* __restrict__ II, const float* __restrict__ JJ,
                              const float* __restrict__ KK, int64_t n, float defv, float* __restrict__ out) {
    GRID_STRIDE(i, n) {
        const float x = II[i], y = JJ[i], z = KK[i];
        float* o = out + i * C;
        const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= (float)(nx - 1)) && (y <= (float)(ny - 1)) &&
                        (z <= (float)(nz - 1));
        if (!ok) {
            for (int c = 0; c < C; ++c) o[c] = defv;
            continue;
        }
        const float fxf = floorf(x), fyf = floorf(y), fzf = floorf(z);
        const int fx = (int)fxf, fy = (int)fyf, fz = (int)fzf;
        const int cx = min(fx + 1, nx - 1), cy = min(fy + 1, ny - 1), cz = min(fz + 1, nz - 1);
        const float wcx = x - fxf, wcy = y - fyf, wcz = z - fzf;
        const float wfx = 1.f - wcx, wfy = 1.f - wcy, wfz = 1.f - wcz;
        const int64_t sx = (int64_t)ny * nz * C, sy = (int64_t)nz * C;
        const float* p000 = X + fx * sx + fy * sy + (int64_t)fz * C;
        const float* p100 = X + cx * sx + fy * sy + (int64_t)fz * C;
        const float* p010 = X + fx * sx + cy * sy + (int64_t)fz * C;
        const float* p110 = X + cx * sx + cy * sy + (int64_t)fz * C;
        const float* p001 = X + fx * sx + fy * sy + (int64_t)cz * C;
        const float* p101 = X + cx * sx + fy * sy + (int64_t)cz * C;
        const float* p011 = X + fx * sx + cy * sy + (int64_t)cz * C;
        const float* p111 = X + cx * sx + cy * sy + (int64_t)cz * C;
        for (int c = 0; c < C; ++c) {
            const float c00 = ld_tex(p000 + c) * wfx + ld_tex(p100 + c) * wcx;
            const float c01 = ld_tex(p001 + c) * wfx + ld_tex(p101 + c) * wcx;
            const float c10 = ld_tex(p010 + c) * wfx + ld_tex(p110 + c) * wcx;
            const float c11 = ld_tex(p011 + c) * wfx + ld_tex(p111 + c) * wcx;
            const float c0 = c00 * wfy + c10 * wcy;
            const float c1 = c01 * wfy + c11 * wcy;
            o[c] = c0 * wfz + c1 * wcz;
        }
    }
}

// get_deformed_atlas (utils/test_utils.py:45-57) fused: mask -> affine of 100*reg -> trilinear sample of the atlas
struct Aff34 { float a[12]; };
// NZ: the mask operand is a tile's input image and M = (im != 0), the mask scripts/demo_test.py:88-89 builds
// (1 where the image is non-zero, also where it is negative or NaN) before it calls get_deformed_atlas
// The atlas texels are read with L1-bypassing loads (ld_tex above explains why; this kernel keeps round 2's system-scope
// form, global_load_dword sc0 sc1, which 90 volumes of 256^3 and 6 of 512^3 have been checked with; the agent-scope form
// measured as clean: profiles/r03_atlas_gather_hazard.txt).
__device__ __forceinline__ float ld_l2(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// LOADS (diagnostic forms behind BFM_ATLAS_PLAIN_LOADS, tests/diag/diag_atlas_flow.py): 0 = what ships (sc0 sc1: system
// scope, served past the non-coherent caches); 1 = ordinary global_load_dword, the form that misbehaved; 2 = ordinary loads
// behind an agent-scope acquire at the start of the kernel (buffer_inv sc1: this CU's L1 invalidated); 3 = agent-scope
// loads (sc1: past the L1, served by the XCD's L2)
// round 5 bisection (diagnostics only): 4 = ordinary texel loads with the mask predicate dropped (every voxel sampled, the
// mask applied to the result: no divergent, exec-masked gather); 5 = ordinary texel loads, the four coalesced ROW loads at
// agent scope; 7 = ordinary loads everywhere with the row loads drained (s_waitcnt vmcnt(0)) before the first texel load
template <bool NZ, int LOADS = 0>
__global__ void deformed_atlas(const float* __restrict__ mask, const float* __restrict__ rx,
                               const float* __restrict__ ry, const float* __restrict__ rz, const float* X, int nx,
                               int ny, int nz, Aff34 A, int64_t n, float* __restrict__ out) {
    if (LOADS == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    GRID_STRIDE(i, n) {
        float r = 0.f;
        const float mk = LOADS == 5 ? __hip_atomic_load(mask + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mask[i];
        const bool keep = NZ ? (mk != 0.f) : (mk > 0.f);
        if (keep || LOADS == 4) {
            float r0 = rx[i], r1 = ry[i], r2 = rz[i];
            if (LOADS == 5) {
                r0 = __hip_atomic_load(rx + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r1 = __hip_atomic_load(ry + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r2 = __hip_atomic_load(rz + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (LOADS == 7) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2));
            const float xx = 100.f * r0, yy = 100.f * r1, zz = 100.f * r2;
            const float x = ((A.a[0] * xx + A.a[1] * yy) + A.a[2] * zz) + A.a[3];
            const float y = ((A.a[4] * xx + A.a[5] * yy) + A.a[6] * zz) + A.a[7];
            const float z = ((A.a[8] * xx + A.a[9] * yy) + A.a[10] * zz) + A.a[11];
            const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= (float)(nx - 1)) && (y <= (float)(ny - 1)) &&
                            (z <= (float)(nz - 1));
            if (ok) {
                const float fxf = floorf(x), fyf = floorf(y), fzf = floorf(z);
                const int fx = (int)fxf, fy = (int)fyf, fz = (int)fzf;
                const int cx = min(fx + 1, nx - 1), cy = min(fy + 1, ny - 1), cz = min(fz + 1, nz - 1);
                const float wcx = x - fxf, wcy = y - fyf, wcz = z - fzf;
                const float wfx = 1.f - wcx, wfy = 1.f - wcy, wfz = 1.f - wcz;
                auto at = [&](int a, int b, int c) {
                    const float* q = X + (((int64_t)a * ny + b) * nz + c);
                    if (LOADS == 1 || LOADS == 2 || LOADS == 4 || LOADS == 5 || LOADS == 7) return *q;
                    if (LOADS == 3) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return ld_l2(q);
                };
                const float c00 = at(fx, fy, fz) * wfx + at(cx, fy, fz) * wcx;
                const float c01 = at(fx, fy, cz) * wfx + at(cx, fy, cz) * wcx;
                const float c10 = at(fx, cy, fz) * wfx + at(cx, cy, fz) * wcx;
                const float c11 = at(fx, cy, cz) * wfx + at(cx, cy, cz) * wcx;
                const float c0 = c00 * wfy + c10 * wcy;
                const float c1 = c01 * wfy + c11 * wcy;
                r = c0 * wfz + c1 * wcz;
            }
        }
        out[i] = (LOADS == 4 && !keep) ? 0.f : r;
    }
}

// bit copy of 4-byte elements (int32 labels or fp32 values)
__global__ void interp_nearest(const uint32_t* __restrict__ X, int nx, int ny, int nz, int C,
                               const float* __restrict__ II, const float* __restrict__ JJ,
                               const float* __restrict__ KK, int64_t n, uint32_t* __restrict__ out) {
    GRID_STRIDE(i, n) {
        int x = (int)rintf(II[i]), y = (int)rintf(JJ[i]), z = (int)rintf(KK[i]);    // half to even, like torch.round
        x = min(max(x, 0), nx - 1); y = min(max(y, 0), ny - 1); z = min(max(z, 0), nz - 1);
        const uint32_t* p = X + (((int64_t)x * ny + y) * nz + z) * C;
        uint32_t* o = out + i * C;
        for (int c = 0; c < C; ++c) o[c] = ld_tex(p + c);
    }
}

struct ZoomTabs {
    const int32_t *fx, *cx, *fy, *cy, *fz, *cz;
    const float *wfx, *wcx, *wfy, *wcy, *wfz, *wcz;
};

// One wave per output (x, y) row, several rows per wave.  The x and y passes of a row depend on the source z only:
//   B[zs] = wfy * (wfx * X[fx,fy,zs] + wcx * X[cx,fy,zs]) + wcy * (wfx * X[fx,cy,zs] + wcx * X[cx,cy,zs])
// is formed once per source position in the wave's LDS row, and every output is wfz * B[fz] + wcz * B[cz] -- the
// reference's pass order (x, y, z) and expressions, so the same bits as the per-output form, with 4 global loads per
// SOURCE element instead of 8 per OUTPUT element.  The z tables of the row (index pair and weights per output element)
// are staged in LDS once per workgroup, and a lane writes four consecutive outputs with one 16-byte store: the kernel
// is a write stream (round 3: 59 us for 6^3 x 3 -> 160^3 x 3; per-output form with wave rows: 36 us).
constexpr int ZOOM_ROW = 1024;                                  // source row (nz * C floats) that fits the wave's LDS slot
constexpr int ZOOM_OUT = 2048;                                  // output row (oz * C elements) whose z tables fit LDS
__global__ void __launch_bounds__(256) zoom_linear(const float* __restrict__ X, int nx, int ny, int nz, int C, ZoomTabs t,
                                                   int ox, int oy, int oz, float* __restrict__ out) {
    extern __shared__ float zsm[];                                // [4][srclen] B rows, then the four z tables [rowlen]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = ox * oy, rowlen = oz * C, srclen = nz * C;
    int* sF = reinterpret_cast<int*>(zsm + 4 * srclen);
    int* sC = sF + rowlen;
    float* sWf = reinterpret_cast<float*>(sC + rowlen);
    float* sWc = sWf + rowlen;
    const int64_t sx = (int64_t)ny * nz * C, sy = (int64_t)nz * C;
    for (int e = threadIdx.x; e < rowlen; e += 256) {
        const int k = C == 1 ? e : e / C, c = C == 1 ? 0 : e - k * C;
        sF[e] = t.fz[k] * C + c; sC[e] = t.cz[k] * C + c; sWf[e] = t.wfz[k]; sWc[e] = t.wcz[k];
    }
    __syncthreads();
    float* B = zsm + w * srclen;
    const bool vec = (rowlen & 3) == 0 && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const int ii = r / oy, j = r - ii * oy;
        const int fx = t.fx[ii], cx = t.cx[ii], fy = t.fy[j], cy = t.cy[j];
        const float wfx = t.wfx[ii], wcx = t.wcx[ii], wfy = t.wfy[j], wcy = t.wcy[j];
        const float* p00 = X + fx * sx + fy * sy;
        const float* p10 = X + cx * sx + fy * sy;
        const float* p01 = X + fx * sx + cy * sy;
        const float* p11 = X + cx * sx + cy * sy;
        for (int e = lane; e < srclen; e += 64) {
            const float a0 = wfx * p00[e] + wcx * p10[e];
            const float a1 = wfx * p01[e] + wcx * p11[e];
            B[e] = wfy * a0 + wcy * a1;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the wave's own LDS writes have landed
        float* o = out + (int64_t)r * rowlen;
        if (vec) {
            for (int e = lane * 4; e < rowlen; e += 256) {
                float4 v;
                v.x = sWf[e] * B[sF[e]] + sWc[e] * B[sC[e]];
                v.y = sWf[e + 1] * B[sF[e + 1]] + sWc[e + 1] * B[sC[e + 1]];
                v.z = sWf[e + 2] * B[sF[e + 2]] + sWc[e + 2] * B[sC[e + 2]];
                v.w = sWf[e + 3] * B[sF[e + 3]] + sWc[e + 3] * B[sC[e + 3]];
                *reinterpret_cast<float4*>(o + e) = v;
            }
        } else {
            for (int e = lane; e < rowlen; e += 64) o[e] = sWf[e] * B[sF[e]] + sWc[e] * B[sC[e]];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Upsampling one channel from a few lattice points (the bias field 5^3 -> 160^3; three channels measured slower here): the
// WHOLE source and the three axis tables sit in LDS, a thread forms four consecutive outputs from eight LDS reads each --
// no dependent global load anywhere -- and stores them as one 16-byte word.  Same expressions, same order.
constexpr int ZOOM_SRC = 2048;                                  // source elements (nx * ny * nz * C) the LDS form takes
constexpr int ZOOM_TAB = 1024;                                  // ox + oy + oz (24 KB of LDS in all: six workgroups per CU)
__global__ void __launch_bounds__(256) zoom_linear_small(const float* __restrict__ X, int nx, int ny, int nz, int C, ZoomTabs t,
                                                         int ox, int oy, int oz, float* __restrict__ out) {
    __shared__ float sX[ZOOM_SRC];
    __shared__ int sF[ZOOM_TAB], sC[ZOOM_TAB];
    __shared__ float sWf[ZOOM_TAB], sWc[ZOOM_TAB];
    const int nsrc = nx * ny * nz * C;
    for (int i = threadIdx.x; i < nsrc; i += 256) sX[i] = X[i];
    for (int i = threadIdx.x; i < ox + oy + oz; i += 256) {
        const int a = i < ox ? 0 : (i < ox + oy ? 1 : 2);
        const int k = a == 0 ? i : (a == 1 ? i - ox : i - ox - oy);
        const int32_t* f = a == 0 ? t.fx : (a == 1 ? t.fy : t.fz);
        const int32_t* c = a == 0 ? t.cx : (a == 1 ? t.cy : t.cz);
        const float* wf = a == 0 ? t.wfx : (a == 1 ? t.wfy : t.wfz);
        const float* wc = a == 0 ? t.wcx : (a == 1 ? t.wcy : t.wcz);
        sF[i] = f[k]; sC[i] = c[k]; sWf[i] = wf[k]; sWc[i] = wc[k];
    }
    __syncthreads();
    const int rowlen = oz * C;
    const int64_t n = (int64_t)ox * oy * rowlen;
    const int sxs = ny * nz * C, sys = nz * C;
    auto one = [&](int fx, int cx, int fy, int cy, float wfx, float wcx, float wfy, float wcy, int e) -> float {
        const int k = C == 1 ? e : e / C, c = C == 1 ? 0 : e - k * C;
        const int fz = sF[ox + oy + k] * C + c, cz = sC[ox + oy + k] * C + c;
        const float wfz = sWf[ox + oy + k], wcz = sWc[ox + oy + k];
        const float a00 = wfx * sX[fx * sxs + fy * sys + fz] + wcx * sX[cx * sxs + fy * sys + fz];
        const float a10 = wfx * sX[fx * sxs + cy * sys + fz] + wcx * sX[cx * sxs + cy * sys + fz];
        const float a01 = wfx * sX[fx * sxs + fy * sys + cz] + wcx * sX[cx * sxs + fy * sys + cz];
        const float a11 = wfx * sX[fx * sxs + cy * sys + cz] + wcx * sX[cx * sxs + cy * sys + cz];
        const float b0 = wfy * a00 + wcy * a10;
        const float b1 = wfy * a01 + wcy * a11;
        return wfz * b0 + wcz * b1;
    };
    // rowlen % 4 == 0 (checked by the host): a group of four outputs never straddles two rows
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q * 4 < n; q += (int64_t)gridDim.x * 256) {
        const int64_t i = q * 4;
        const int r = (int)(i / rowlen), e = (int)(i - (int64_t)r * rowlen);
        const int ii = r / oy, j = r - ii * oy;
        const int fx = sF[ii], cx = sC[ii], fy = sF[ox + j], cy = sC[ox + j];
        const float wfx = sWf[ii], wcx = sWc[ii], wfy = sWf[ox + j], wcy = sWc[ox + j];
        float4 v;
        v.x = one(fx, cx, fy, cy, wfx, wcx, wfy, wcy, e);
        v.y = one(fx, cx, fy, cy, wfx, wcx, wfy, wcy, e + 1);
        v.z = one(fx, cx, fy, cy, wfx, wcx, wfy, wcy, e + 2);
        v.w = one(fx, cx, fy, cy, wfx, wcx, wfy, wcy, e + 3);
        *reinterpret_cast<float4*>(out + i) = v;
    }
}

// the same for source rows longer than the LDS slot: per output element
__global__ void __launch_bounds__(256) zoom_linear_long(const float* __restrict__ X, int nx, int ny, int nz, int C,
                                                        ZoomTabs t, int ox, int oy, int oz, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int rows = ox * oy, rowlen = oz * C;
    const int64_t sx = (int64_t)ny * nz * C, sy = (int64_t)nz * C;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        const int ii = r / oy, j = r - ii * oy;
        const int fx = t.fx[ii], cx = t.cx[ii], fy = t.fy[j], cy = t.cy[j];
        const float wfx = t.wfx[ii], wcx = t.wcx[ii], wfy = t.wfy[j], wcy = t.wcy[j];
        const float* p00 = X + fx * sx + fy * sy;
        const float* p10 = X + cx * sx + fy * sy;
        const float* p01 = X + fx * sx + cy * sy;
        const float* p11 = X + cx * sx + cy * sy;
        float* o = out + (int64_t)r * rowlen;
        for (int e = lane; e < rowlen; e += 64) {
            const int k = C == 1 ? e : e / C, c = C == 1 ? 0 : e - k * C;
            const int fz = t.fz[k] * C + c, cz = t.cz[k] * C + c;
            const float wfz = t.wfz[k], wcz = t.wcz[k];
            const float a00 = wfx * p00[fz] + wcx * p10[fz];
            const float a10 = wfx * p01[fz] + wcx * p11[fz];
            const float a01 = wfx * p00[cz] + wcx * p10[cz];
            const float a11 = wfx * p01[cz] + wcx * p11[cz];
            const float b0 = wfy * a00 + wcy * a10;
            const float b1 = wfy * a01 + wcy * a11;
            o[e] = wfz * b0 + wcz * b1;
        }
    }
}

// gaussian_blur_3d, one axis: a workgroup stages a [len][64] slab (the whole extent along the filtered axis x 64
// consecutive z for axes 0 / 1; 64 rows x the whole z extent for axis 2) in LDS and forms every output from it -- each
// input element leaves HBM / L2 once instead of once per tap (round 3: one global load per tap and output, 34 us per pass
// at 160^3).  The taps are added in the reference's order (ascending, those outside the volume skipped), with fmaf, as
// before: the same bits.
constexpr int C1D_TAPS = 64;
constexpr int C1D_RESIDENT = 2048;                              // workgroups of 256 threads the chip holds at once (256 CUs x 8)
constexpr int C1D_NZ = 1024;                                    // longest z row of the axis-2 slab form
// seg: outputs along the filtered axis per workgroup (axes 0, 1) or rows per workgroup (axis 2) -- chosen by the host so
// that the slabs number about C1D_RESIDENT: every workgroup is resident at once and the kernel lasts one slab's latency
// (load -> LDS -> taps -> store) instead of one per round of workgroups (32-row slabs: 2 400 of them in 1.2 rounds, 17 us).
template <int AXIS>
__global__ void __launch_bounds__(256) conv1d_slab(const float* __restrict__ in, int nx, int ny, int nz,
                                                   const float* __restrict__ kern, int klen, int seg,
                                                   float* __restrict__ out) {
    const int C1D_SEG = seg, C1D_ROWS = seg;
    extern __shared__ float slab[];                               // [SEG + klen - 1][65] (axes 0, 1) or [ROWS][nz | 1] (axis 2)
    __shared__ float taps[C1D_TAPS];
    const int half = klen / 2;
    for (int j = threadIdx.x; j < klen; j += 256) taps[j] = kern[j];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (AXIS != 2) {
        const int len = AXIS == 0 ? nx : ny;
        const int other = AXIS == 0 ? ny : nx;                    // the axis that is neither filtered nor z
        const int64_t stride = AXIS == 0 ? (int64_t)ny * nz : nz;
        const int zchunks = (nz + 63) / 64, segs = (len + C1D_SEG - 1) / C1D_SEG;
        const int nslab = other * zchunks * segs;
        for (int sl = blockIdx.x; sl < nslab; sl += gridDim.x) {
            const int sg = sl % segs, rest = sl / segs;
            const int o = rest / zchunks, z0 = (rest - o * zchunks) * 64;
            const int p0 = sg * C1D_SEG, p1 = min(len, p0 + C1D_SEG);
            const int lo = max(0, p0 - half), hi = min(len, p1 + half);      // rows of the volume the segment's taps touch
            const int64_t base = (AXIS == 0 ? (int64_t)o * nz : (int64_t)o * ny * nz) + z0;
            const int z = z0 + lane;
            __syncthreads();
            // eight rows in flight per wave before the first LDS store (one load -> wait -> store per row serialised the
            // slab's ~12 rows per wave behind 12 memory round trips: 15 us per pass whatever the tap count)
            for (int p = lo + w; p < hi; p += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = p + 4 * u;
                    v[u] = (q < hi && z < nz) ? in[base + (int64_t)q * stride + lane] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = p + 4 * u;
                    if (q < hi) slab[(q - lo) * 65 + lane] = v[u];
                }
            }
            __syncthreads();
            if (z < nz)
                for (int p = p0 + w; p < p1; p += 4) {
                    const int j0 = max(0, half - p), j1 = min(klen, len + half - p);
                    float acc = 0.f;
#pragma unroll 4
                    for (int j = j0; j < j1; ++j) acc = fmaf(taps[j], slab[(p + j - half - lo) * 65 + lane], acc);
                    out[base + (int64_t)p * stride + lane] = acc;
                }
        }
    } else {
        const int len = nz, ld = nz | 1;
        const int rows = nx * ny;
        const int nslab = (rows + C1D_ROWS - 1) / C1D_ROWS;
        for (int sl = blockIdx.x; sl < nslab; sl += gridDim.x) {
            const int r0 = sl * C1D_ROWS, nr = min(C1D_ROWS, rows - r0);
            const int64_t base = (int64_t)r0 * nz;
            __syncthreads();
            for (int e0 = threadIdx.x; e0 < nr * len; e0 += 256 * 8) {      // consecutive rows are one contiguous run
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + 256 * u;
                    v[u] = e < nr * len ? in[base + e] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + 256 * u;
                    if (e < nr * len) {
                        const int r = e / len, z = e - r * len;
                        slab[r * ld + z] = v[u];
                    }
                }
            }
            __syncthreads();
            for (int e = threadIdx.x; e < nr * len; e += 256) {
                const int r = e / len, z = e - r * len;
                const int j0 = max(0, half - z), j1 = min(klen, len + half - z);
                const float* p = slab + r * ld + z - half;
                float acc = 0.f;
#pragma unroll 4
                for (int j = j0; j < j1; ++j) acc = fmaf(taps[j], p[j], acc);
                out[base + e] = acc;
            }
        }
    }
}

// any extent: one wave per (x, y) row, lanes along z, one global load per tap
__global__ void __launch_bounds__(256) conv1d_axis(const float* __restrict__ in, int nx, int ny, int nz, int axis,
                                                   const float* __restrict__ kern, int klen, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int rows = nx * ny;
    const int half = klen / 2;
    const int64_t stride = axis == 0 ? (int64_t)ny * nz : (axis == 1 ? nz : 1);
    const int len = axis == 0 ? nx : (axis == 1 ? ny : nz);
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        const int x = r / ny, y = r - x * ny;
        const int64_t base = (int64_t)r * nz;
        for (int z = lane; z < nz; z += 64) {
            const int pos = axis == 0 ? x : (axis == 1 ? y : z);
            const int j0 = max(0, half - pos), j1 = min(klen, len + half - pos);       // taps inside the volume
            const float* p = in + base + z + (int64_t)(j0 - half) * stride;
            float acc = 0.f;
            for (int j = j0; j < j1; ++j, p += stride) acc = fmaf(kern[j], *p, acc);
            out[base + z] = acc;
        }
    }
}

// ---- interpol bounds (utils/interpol/bounds.py:24-89)
__device__ __forceinline__ int imod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

__device__ __forceinline__ int bound_index(int i, int n, int b) {
    switch (b) {
        case 0: case 1: return min(max(i, 0), n - 1);
        case 3: case 5: {
            const int n2 = n * 2;
            i = i < 0 ? n2 - 1 - imod(-i - 1, n2) : imod(i, n2);
            return i >= n ? n2 - 1 - i : i;
        }
        case 2: {
            if (n == 1) return 0;
            const int n2 = (n - 1) * 2;
            i = imod(abs(i), n2);
            return i >= n ? n2 - i : i;
        }
        case 4: {
            const int n2 = 2 * (n + 1);
            i = i < 0 ? -i - 2 : i;
            i = imod(i, n2);
            i = i > n ? n2 - 2 - i : i;
            i = i == -1 ? 0 : i;
            return i == n ? n - 1 : i;
        }
        case 6: return imod(i, n);
        default: return i;
    }
}

__device__ __forceinline__ int bound_sign(int i, int n, int b) {
    switch (b) {
        case 4: {
            if (n == 1) return 1;
            const int n2 = 2 * (n + 1);
            i = i < 0 ? n - 1 - i : i;
            i = imod(i, n2);
            int x = i == 0 ? 0 : 1;
            x = (imod(i, n + 1) == n) ? 0 : x;
            i = i / (n + 1);
            return (i & 1) ? -x : x;
        }
        case 5: {
            i = i < 0 ? n - 1 - i : i;
            i = i / n;
            return (i & 1) ? -1 : 1;
        }
        case 0: return (i < 0 || i >= n) ? 0 : 1;
        default: return 1;
    }
}

__global__ void grid_pull3d(const float* __restrict__ inp, int Bi, int C, int nx, int ny, int nz,
                            const float* __restrict__ grid, int Bg, int64_t nout, int bx, int by, int bz, int extrap,
                            int B, float* __restrict__ out) {
    const int64_t n = (int64_t)B * nout;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / nout);
        const int64_t v = i - (int64_t)b * nout;
        const float* g = grid + ((int64_t)(Bg == 1 ? 0 : b) * nout + v) * 3;
        const float gx = ld_tex(g), gy = ld_tex(g + 1), gz = ld_tex(g + 2);   // 4-byte agent-scope loads (HISTORY.md section 3.3, round 5)
        float mask = 1.f;
        if (extrap == 0 || extrap == 2) {
            const float thr = extrap == 2 ? 0.5f + 5e-2f : 5e-2f;
            const bool in = (gx > -thr) && (gx < (float)(nx - 1) + thr) && (gy > -thr) && (gy < (float)(ny - 1) + thr) &&
                            (gz > -thr) && (gz < (float)(nz - 1) + thr);
            mask = in ? 1.f : 0.f;
        }
        const float fxf = floorf(gx), fyf = floorf(gy), fzf = floorf(gz);
        const int x0 = (int)fxf, y0 = (int)fyf, z0 = (int)fzf;
        const float wx = gx - fxf, wy = gy - fyf, wz = gz - fzf;
        int ix[2] = {bound_index(x0, nx, bx), bound_index(x0 + 1, nx, bx)};
        int iy[2] = {bound_index(y0, ny, by), bound_index(y0 + 1, ny, by)};
        int iz[2] = {bound_index(z0, nz, bz), bound_index(z0 + 1, nz, bz)};
        int sx[2] = {bound_sign(x0, nx, bx), bound_sign(x0 + 1, nx, bx)};
        int sy[2] = {bound_sign(y0, ny, by), bound_sign(y0 + 1, ny, by)};
        int sz[2] = {bound_sign(z0, nz, bz), bound_sign(z0 + 1, nz, bz)};
        const float ux[2] = {1.f - wx, wx}, uy[2] = {1.f - wy, wy}, uz[2] = {1.f - wz, wz};
        const int64_t vol = (int64_t)nx * ny * nz;
        for (int c = 0; c < C; ++c) {
            const float* src = inp + ((int64_t)(Bi == 1 ? 0 : b) * C + c) * vol;
            float acc = 0.f;
            bool first = true;
            // corner order of iso1.pull3d: 000, 001, 010, 011, 100, 101, 110, 111 (x slowest)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float val = ld_tex(src + ((int64_t)ix[a] * ny + iy[bb]) * nz + iz[d]);
                        val = val * (float)(sx[a] * sy[bb] * sz[d]);
                        val = val * ((ux[a] * uy[bb]) * uz[d]);
                        acc = first ? val : acc + val;
                        first = false;
                    }
            out[((int64_t)b * C + c) * nout + v] = acc * mask;
        }
    }
}

// iso1.push3d (utils/interpol/iso1.py:136-265): the adjoint of pull3d -- every source voxel scatters its value to the
// 8 corners around its grid coordinate.  out [B][C][nx*ny*nz] must be zero on entry.  fp32 atomics: the summation order
// (and so the last bit) is not reproducible, like torch's scatter_add_ on a GPU.
__global__ void grid_push3d(const float* __restrict__ inp, int Bi, int C, const float* __restrict__ grid, int Bg,
                            int64_t nin, int nx, int ny, int nz, int bx, int by, int bz, int extrap, int B,
                            float* __restrict__ out) {
    const int64_t n = (int64_t)B * nin;
    const int64_t vol = (int64_t)nx * ny * nz;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / nin);
        const int64_t v = i - (int64_t)b * nin;
        const float* g = grid + ((int64_t)(Bg == 1 ? 0 : b) * nin + v) * 3;
        const float gx = ld_tex(g), gy = ld_tex(g + 1), gz = ld_tex(g + 2);   // 4-byte agent-scope loads (HISTORY.md section 3.3, round 5)
        float mask = 1.f;
        if (extrap == 0 || extrap == 2) {
            const float thr = extrap == 2 ? 0.5f + 5e-2f : 5e-2f;
            const bool in = (gx > -thr) && (gx < (float)(nx - 1) + thr) && (gy > -thr) && (gy < (float)(ny - 1) + thr) &&
                            (gz > -thr) && (gz < (float)(nz - 1) + thr);
            mask = in ? 1.f : 0.f;
        }
        const float fxf = floorf(gx), fyf = floorf(gy), fzf = floorf(gz);
        const int x0 = (int)fxf, y0 = (int)fyf, z0 = (int)fzf;
        const float wx = gx - fxf, wy = gy - fyf, wz = gz - fzf;
        int ix[2] = {bound_index(x0, nx, bx), bound_index(x0 + 1, nx, bx)};
        int iy[2] = {bound_index(y0, ny, by), bound_index(y0 + 1, ny, by)};
        int iz[2] = {bound_index(z0, nz, bz), bound_index(z0 + 1, nz, bz)};
        int sx[2] = {bound_sign(x0, nx, bx), bound_sign(x0 + 1, nx, bx)};
        int sy[2] = {bound_sign(y0, ny, by), bound_sign(y0 + 1, ny, by)};
        int sz[2] = {bound_sign(z0, nz, bz), bound_sign(z0 + 1, nz, bz)};
        const float ux[2] = {1.f - wx, wx}, uy[2] = {1.f - wy, wy}, uz[2] = {1.f - wz, wz};
        for (int c = 0; c < C; ++c) {
            const float val = inp[((int64_t)(Bi == 1 ? 0 : b) * C + c) * nin + v];
            float* dst = out + ((int64_t)b * C + c) * vol;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float t = val * (float)(sx[a] * sy[bb] * sz[d]);
                        t = t * mask;
                        t = t * ((ux[a] * uy[bb]) * uz[d]);
                        atomicAdd(dst + ((int64_t)ix[a] * ny + iy[bb]) * nz + iz[d], t);
                    }
        }
    }
}

// iso1.grad3d (utils/interpol/iso1.py:268-387): spatial gradient of the trilinear interpolant at the grid coordinates,
// out [B][C][nout][3].
__global__ void grid_grad3d(const float* __restrict__ inp, int Bi, int C, int nx, int ny, int nz,
                            const float* __restrict__ grid, int Bg, int64_t nout, int bx, int by, int bz, int extrap,
                            int B, float* __restrict__ out) {
    const int64_t n = (int64_t)B * nout;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / nout);
        const int64_t v = i - (int64_t)b * nout;
        const float* g = grid + ((int64_t)(Bg == 1 ? 0 : b) * nout + v) * 3;
        const float gx = ld_tex(g), gy = ld_tex(g + 1), gz = ld_tex(g + 2);   // 4-byte agent-scope loads (HISTORY.md section 3.3, round 5)
        float mask = 1.f;
        if (extrap == 0 || extrap == 2) {
            const float thr = extrap == 2 ? 0.5f + 5e-2f : 5e-2f;
            const bool in = (gx > -thr) && (gx < (float)(nx - 1) + thr) && (gy > -thr) && (gy < (float)(ny - 1) + thr) &&
                            (gz > -thr) && (gz < (float)(nz - 1) + thr);
            mask = in ? 1.f : 0.f;
        }
        const float fxf = floorf(gx), fyf = floorf(gy), fzf = floorf(gz);
        const int x0 = (int)fxf, y0 = (int)fyf, z0 = (int)fzf;
        const float wx = gx - fxf, wy = gy - fyf, wz = gz - fzf;
        int ix[2] = {bound_index(x0, nx, bx), bound_index(x0 + 1, nx, bx)};
        int iy[2] = {bound_index(y0, ny, by), bound_index(y0 + 1, ny, by)};
        int iz[2] = {bound_index(z0, nz, bz), bound_index(z0 + 1, nz, bz)};
        int sx[2] = {bound_sign(x0, nx, bx), bound_sign(x0 + 1, nx, bx)};
        int sy[2] = {bound_sign(y0, ny, by), bound_sign(y0 + 1, ny, by)};
        int sz[2] = {bound_sign(z0, nz, bz), bound_sign(z0 + 1, nz, bz)};
        const float ux[2] = {1.f - wx, wx}, uy[2] = {1.f - wy, wy}, uz[2] = {1.f - wz, wz};
        const float dx[2] = {-1.f, 1.f};
        const int64_t vol = (int64_t)nx * ny * nz;
        for (int c = 0; c < C; ++c) {
            const float* src = inp + ((int64_t)(Bi == 1 ? 0 : b) * C + c) * vol;
            float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float val = ld_tex(src + ((int64_t)ix[a] * ny + iy[bb]) * nz + iz[d]);
                        val = val * (float)(sx[a] * sy[bb] * sz[d]);
                        ax = fmaf(val, dx[a] * (uy[bb] * uz[d]), ax);
                        ay = fmaf(val, dx[bb] * (ux[a] * uz[d]), ay);
                        az = fmaf(val, dx[d] * (ux[a] * uy[bb]), az);
                    }
            float* o = out + (((int64_t)b * C + c) * nout + v) * 3;
            o[0] = ax * mask; o[1] = ay * mask; o[2] = az * mask;
        }
    }
}

struct Affine { float a[9]; float c[3]; int shp[3]; };

// xx2 = A[r,0]*xx1 + A[r,1]*yy1 + A[r,2]*zz1 + c2[r], clamped to the source shape; block min/max partials
__global__ void deform_grid_k(const float* __restrict__ F, int sx, int sy, int sz, Affine P, float* __restrict__ xx,
                              float* __restrict__ yy, float* __restrict__ zz, float* __restrict__ part) {
    const int64_t n = (int64_t)sx * sy * sz;
    const float cx = (float)((sx - 1) / 2.0), cy = (float)((sy - 1) / 2.0), cz = (float)((sz - 1) / 2.0);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    GRID_STRIDE(i, n) {
        const int z = (int)(i % sz);
        const int y = (int)((i / sz) % sy);
        const int x = (int)(i / ((int64_t)sy * sz));
        float x1 = (float)x - cx, y1 = (float)y - cy, z1 = (float)z - cz;
        if (F) { x1 = x1 + F[i * 3 + 0]; y1 = y1 + F[i * 3 + 1]; z1 = z1 + F[i * 3 + 2]; }
        float r[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = ((P.a[k * 3 + 0] * x1 + P.a[k * 3 + 1] * y1) + P.a[k * 3 + 2] * z1) + P.c[k];
            v = v < 0.f ? 0.f : v;
            const float hi = (float)(P.shp[k] - 1);
            v = v > hi ? hi : v;
            r[k] = v;
            mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v);
        }
        xx[i] = r[0]; yy[i] = r[1]; zz[i] = r[2];
    }
    __shared__ float red[6][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float a = wave_reduce_min(mn[k]), b = wave_reduce_max(mx[k]);
        if ((threadIdx.x & 63) == 0) { red[k][threadIdx.x >> 6] = a; red[3 + k][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[k][0];
        for (int w = 1; w < 4; ++w) v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
        part[(size_t)blockIdx.x * 6 + k] = v;
    }
}

__global__ void minmax6_final(const float* __restrict__ part, int nb, float* __restrict__ out) {
    const int k = threadIdx.x >> 6, l = threadIdx.x & 63;                // 6 waves, one per component
    float v = k < 3 ? INFINITY : -INFINITY;
    for (int b = l; b < nb; b += 64) v = k < 3 ? fminf(v, part[(size_t)b * 6 + k]) : fmaxf(v, part[(size_t)b * 6 + k]);
    v = k < 3 ? wave_reduce_min(v) : wave_reduce_max(v);
    if (l == 0) out[k] = v;
}

__global__ void label_gauss(const float* __restrict__ G, const float* __restrict__ mus,
                            const float* __restrict__ sigmas, const float* __restrict__ rn, int64_t n, int ntab,
                            float* __restrict__ out) {
    GRID_STRIDE(i, n) {
        float g = G[i];
        g = g == 77.f ? 2.f : g;                               // merge WM lesion into WM (datasets.py:368)
        int l = (int)rintf(g);
        l = min(max(l, 0), ntab - 1);
        float v = mus[l] + sigmas[l] * rn[i];
        out[i] = v < 0.f ? 0.f : v;
    }
}

// generate_sample's pathology branch (datasets.py:388-396): cerebral copy + class sums; every block leaves its four
// partial sums, label_class_fold adds them in block order (no atomics: gm_mean > wm_mean must not depend on the run)
__global__ void label_class_stats(const float* __restrict__ G, const float* __restrict__ syn, int64_t n,
                                  float* __restrict__ cerebral, double* __restrict__ partials) {
    __shared__ double sh[4][4];
    double a[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n) {
        float g = G[i];
        g = g == 77.f ? 2.f : g;
        const int l = (int)rintf(g);
        const float v = syn[i];
        cerebral[i] = l == 0 ? 0.f : v;
        if (l == 2 || l == 41) { a[0] += (double)v; a[1] += 1.; }
        else if (l != 0) { a[2] += (double)v; a[3] += 1.; }
    }
    for (int k = 0; k < 4; ++k) {
        double v = a[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0.;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += sh[threadIdx.x][w];
        partials[(size_t)blockIdx.x * 4 + threadIdx.x] = v;
    }
}

// 4 waves, wave k folds quantity k: lane l adds its strided partials, then a fixed xor tree (deterministic; the single
// thread walking 1024 partials of round 3 took 110 us per call)
__global__ void label_class_fold(const double* __restrict__ partials, int nb, double* __restrict__ stats) {
    const int k = threadIdx.x >> 6, l = threadIdx.x & 63;
    double v = 0.;
    for (int b = l; b < nb; b += 64) v += partials[(size_t)b * 4 + k];
    v = wave_reduce_sum(v);
    if (l == 0) stats[k] = v;
}

__global__ void onehot_lut(const int32_t* __restrict__ S, const int32_t* __restrict__ lut, int nlut, int nl, int64_t n,
                           float* __restrict__ out) {
    const int64_t tot = n * nl;
    GRID_STRIDE(i, tot) {
        const int64_t v = i / nl;
        const int c = (int)(i - v * nl);
        int s = S[v];
        s = min(max(s, 0), nlut - 1);
        out[i] = lut[s] == c ? 1.f : 0.f;
    }
}

}  // namespace

extern "C" int bfm_interp3d_linear(const float* X, int nx, int ny, int nz, int C, const float* II, const float* JJ,
                                   const float* KK, int64_t n, float default_value, float* out, bfm_stream_t stream) {
    if (!X || !II || !JJ || !KK || !out || nx <= 0 || ny <= 0 || nz <= 0 || C <= 0 || n <= 0) return BFM_E_ARG;
    const int64_t vbytes = (int64_t)nx * ny * nz * 4;
    if (C == 1 && vbytes < ((int64_t)1 << 32)) {
        int mode = 0;
#ifdef BFM_DIAG
        if (const char* e = getenv("BFM_INTERP_MODE")) mode = atoi(e);          // diagnostics: the candidate load forms
#endif
#define BFM_IL1(M) hipLaunchKernelGGL(interp_linear1<M>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), X, nx, ny, nz, (uint32_t)vbytes, \
                                      II, JJ, KK, n, default_value, out)
        // MODE 0 ships again (round 6): one 8-byte load per z pair through a buffer descriptor.  Round 5 shipped MODE 2 (4-byte
        // loads) because the paired forms "failed beside conv_wino4d"; what failed was a packed-FP32 multiply of the corner
        // weights (profiles/r06_hazard_root_cause.txt), and the library holds no such instruction any more.
        if (mode == 1) BFM_IL1(1); else if (mode == 2) BFM_IL1(2); else if (mode == 3) BFM_IL1(3); else BFM_IL1(0);
#undef BFM_IL1
    }
    else
        hipLaunchKernelGGL(interp_linear, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), X, nx, ny, nz, C, II, JJ, KK, n,
                           default_value, out);
    return bfm_launch_status();
}

extern "C" int bfm_interp3d_nearest(const void* X, int nx, int ny, int nz, int C, const float* II, const float* JJ,
                                    const float* KK, int64_t n, void* out, bfm_stream_t stream) {
    if (!X || !II || !JJ || !KK || !out || nx <= 0 || ny <= 0 || nz <= 0 || C <= 0 || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(interp_nearest, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), (const uint32_t*)X, nx, ny, nz,
                       C, II, JJ, KK, n, (uint32_t*)out);
    return bfm_launch_status();
}

extern "C" int bfm_deformed_atlas(const float* mask, const float* regx, const float* regy, const float* regz,
                                  const float* atlas, int nx, int ny, int nz, const float* A_host /*3x4 row-major*/,
                                  int64_t n, float* out, bfm_stream_t stream) {
    if (!mask || !regx || !regy || !regz || !atlas || !A_host || !out || nx <= 0 || ny <= 0 || nz <= 0 || n <= 0)
        return BFM_E_ARG;
    Aff34 A;
    for (int i = 0; i < 12; ++i) A.a[i] = A_host[i];
    hipLaunchKernelGGL(deformed_atlas<false>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), mask, regx, regy, regz,
                       atlas, nx, ny, nz, A, n, out);
    return bfm_launch_status();
}

extern "C" int bfm_deformed_atlas_tile(const float* tile_in, const float* regx, const float* regy, const float* regz,
                                       const float* atlas, int nx, int ny, int nz, const float* A_host, int64_t n,
                                       float* out, bfm_stream_t stream) {
    if (!tile_in || !regx || !regy || !regz || !atlas || !A_host || !out || nx <= 0 || ny <= 0 || nz <= 0 || n <= 0)
        return BFM_E_ARG;
    Aff34 A;
    for (int i = 0; i < 12; ++i) A.a[i] = A_host[i];
#define BFM_ATLAS_LAUNCH(V) hipLaunchKernelGGL((deformed_atlas<true, V>), dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), \
                                               tile_in, regx, regy, regz, atlas, nx, ny, nz, A, n, out)
#ifdef BFM_DIAG
    // diagnostics build only (BFM_HIPCC_EXTRA=-DBFM_DIAG, tests/diag/diag_atlas_flow.py): the load forms that
    // profiles/r03_atlas_gather_hazard.txt documents as returning wrong texels beside LDS-DMA kernels.  The shipped
    // library has no switch: a stray environment variable cannot select them.
    static int loads = -1;
    if (loads < 0) { const char* e = getenv("BFM_ATLAS_PLAIN_LOADS"); loads = e ? atoi(e) : 0; }
    if (const char* e = getenv("BFM_ATLAS_PLAIN_LOADS_NOW")) loads = atoi(e);      // re-read per launch (one-process sweeps)
    if (loads == 1) BFM_ATLAS_LAUNCH(1);
    else if (loads == 2) BFM_ATLAS_LAUNCH(2);
    else if (loads == 3) BFM_ATLAS_LAUNCH(3);
    else if (loads == 4) BFM_ATLAS_LAUNCH(4);
    else if (loads == 5) BFM_ATLAS_LAUNCH(5);
    else if (loads == 7) BFM_ATLAS_LAUNCH(7);
    else BFM_ATLAS_LAUNCH(0);
#else
    BFM_ATLAS_LAUNCH(0);
#endif
#undef BFM_ATLAS_LAUNCH
    return bfm_launch_status();
}

extern "C" int bfm_zoom_linear(const float* X, int nx, int ny, int nz, int C, const bfm_zoom_axis_t* ax, int ox,
                               int oy, int oz, float* out, bfm_stream_t stream) {
    if (!X || !ax || !out || nx <= 0 || ny <= 0 || nz <= 0 || C <= 0 || ox <= 0 || oy <= 0 || oz <= 0) return BFM_E_ARG;
    for (int a = 0; a < 3; ++a)
        if (!ax[a].f || !ax[a].c || !ax[a].wf || !ax[a].wc) return BFM_E_ARG;
    ZoomTabs t{ax[0].f, ax[0].c, ax[1].f, ax[1].c, ax[2].f, ax[2].c, ax[0].wf, ax[0].wc, ax[1].wf, ax[1].wc,
               ax[2].wf, ax[2].wc};
    if ((int64_t)ox * oy > INT32_MAX || (int64_t)oz * C > INT32_MAX) return BFM_E_SHAPE;
    if (C == 1 && (int64_t)nx * ny * nz <= ZOOM_SRC && ox + oy + oz <= ZOOM_TAB && (oz & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(zoom_linear_small, dim3(grid_for((int64_t)ox * oy * oz * C / 4, 256, 1024)), dim3(256), 0,
                           bfm_s(stream), X, nx, ny, nz, C, t, ox, oy, oz, out);
    else if ((int64_t)nz * C <= ZOOM_ROW && (int64_t)oz * C <= ZOOM_OUT)
        hipLaunchKernelGGL(zoom_linear, dim3(grid_for((int64_t)ox * oy, 8, 4096)), dim3(256),
                           (size_t)(4 * nz * C + 4 * oz * C) * sizeof(float), bfm_s(stream), X, nx, ny, nz, C, t, ox, oy, oz,
                           out);
    else
        hipLaunchKernelGGL(zoom_linear_long, dim3(grid_for((int64_t)ox * oy, 4)), dim3(256), 0, bfm_s(stream), X, nx, ny,
                           nz, C, t, ox, oy, oz, out);
    return bfm_launch_status();
}

extern "C" int bfm_conv1d_axis(const float* in, int nx, int ny, int nz, int axis, const float* kern, int klen,
                               float* out, bfm_stream_t stream) {
    if (!in || !kern || !out || nx <= 0 || ny <= 0 || nz <= 0 || axis < 0 || axis > 2 || klen <= 0 || !(klen & 1))
        return BFM_E_ARG;
    if ((int64_t)nx * ny > INT32_MAX) return BFM_E_SHAPE;
    if (klen <= C1D_TAPS && (axis != 2 || nz <= C1D_NZ)) {
        hipStream_t st = bfm_s(stream);
        if (axis == 2) {
            const int rows = nx * ny;
            int seg = std::max(4, bfm_cdiv(rows, C1D_RESIDENT));
            while (seg > 4 && (size_t)seg * (nz | 1) * sizeof(float) > 24 * 1024) --seg;    // <= 24 KB of LDS: 6 per CU
            const int nslab = bfm_cdiv(rows, seg);
            const size_t smem = (size_t)seg * (nz | 1) * sizeof(float);
            hipLaunchKernelGGL(conv1d_slab<2>, dim3(nslab), dim3(256), smem, st, in, nx, ny, nz, kern, klen, seg, out);
        } else {
            const int len = axis == 0 ? nx : ny;
            const int lines = (axis == 0 ? ny : nx) * ((nz + 63) / 64);
            int segs = std::max(1, C1D_RESIDENT / std::max(lines, 1));
            int seg = std::max(8, bfm_cdiv(len, segs));
            while (seg > 8 && (size_t)(seg + klen - 1) * 65 * sizeof(float) > 24 * 1024) --seg;
            const int nslab = lines * bfm_cdiv(len, seg);
            const size_t smem = (size_t)(seg + klen - 1) * 65 * sizeof(float);
            if (axis == 0)
                hipLaunchKernelGGL(conv1d_slab<0>, dim3(nslab), dim3(256), smem, st, in, nx, ny, nz, kern, klen, seg, out);
            else
                hipLaunchKernelGGL(conv1d_slab<1>, dim3(nslab), dim3(256), smem, st, in, nx, ny, nz, kern, klen, seg, out);
        }
        return bfm_launch_status();
    }
    hipLaunchKernelGGL(conv1d_axis, dim3(grid_for((int64_t)nx * ny, 4)), dim3(256), 0, bfm_s(stream), in, nx, ny, nz, axis,
                       kern, klen, out);
    return bfm_launch_status();
}

extern "C" int bfm_grid_pull3d_linear(const float* inp, int Bi, int C, int nx, int ny, int nz, const float* grid,
                                      int Bg, int ox, int oy, int oz, const int* bound, int extrapolate, float* out,
                                      bfm_stream_t stream) {
    if (!inp || !grid || !out || !bound || Bi <= 0 || Bg <= 0 || C <= 0 || nx <= 0 || ny <= 0 || nz <= 0 || ox <= 0 ||
        oy <= 0 || oz <= 0)
        return BFM_E_ARG;
    if (Bi != Bg && Bi != 1 && Bg != 1) return BFM_E_SHAPE;
    for (int a = 0; a < 3; ++a) if (bound[a] < 0 || bound[a] > 6) return BFM_E_ARG;
    if (extrapolate < 0 || extrapolate > 2) return BFM_E_ARG;
    const int B = Bi > Bg ? Bi : Bg;
    const int64_t nout = (int64_t)ox * oy * oz;
    hipLaunchKernelGGL(grid_pull3d, dim3(grid_for(B * nout)), dim3(256), 0, bfm_s(stream), inp, Bi, C, nx, ny, nz, grid,
                       Bg, nout, bound[0], bound[1], bound[2], extrapolate, B, out);
    return bfm_launch_status();
}

extern "C" size_t bfm_deform_grid_workspace(int sx, int sy, int sz) {
    return (size_t)grid_for((int64_t)sx * sy * sz, 256, 1024) * 6 * sizeof(float);
}

extern "C" int bfm_deform_grid(const float* F, int sx, int sy, int sz, const float* A_host, const float* c2_host,
                               const int* shp_host, float* xx, float* yy, float* zz, float* minmax, void* workspace,
                               size_t workspace_bytes, bfm_stream_t stream) {
    if (!A_host || !c2_host || !shp_host || !xx || !yy || !zz || !minmax || !workspace || sx <= 0 || sy <= 0 || sz <= 0)
        return BFM_E_ARG;
    const int nb = grid_for((int64_t)sx * sy * sz, 256, 1024);
    if (workspace_bytes < (size_t)nb * 6 * sizeof(float)) return BFM_E_WORKSPACE;
    Affine P;
    for (int i = 0; i < 9; ++i) P.a[i] = A_host[i];
    for (int i = 0; i < 3; ++i) { P.c[i] = c2_host[i]; P.shp[i] = shp_host[i]; }
    hipLaunchKernelGGL(deform_grid_k, dim3(nb), dim3(256), 0, bfm_s(stream), F, sx, sy, sz, P, xx, yy, zz,
                       static_cast<float*>(workspace));
    hipLaunchKernelGGL(minmax6_final, dim3(1), dim3(384), 0, bfm_s(stream), static_cast<const float*>(workspace), nb,
                       minmax);
    return bfm_launch_status();
}

extern "C" int bfm_label_gauss(const float* G, const float* mus, const float* sigmas, const float* randn, int64_t n,
                               int ntab, float* out, bfm_stream_t stream) {
    if (!G || !mus || !sigmas || !randn || !out || n <= 0 || ntab <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(label_gauss, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), G, mus, sigmas, randn, n, ntab, out);
    return bfm_launch_status();
}

extern "C" int bfm_label_class_stats(const float* G, const float* syn, int64_t n, float* cerebral, double* stats,
                                     double* partials, bfm_stream_t stream) {
    if (!G || !syn || !cerebral || !stats || !partials || n <= 0) return BFM_E_ARG;
    const int nb = grid_for(n, 256, BFM_CLASS_STATS_BLOCKS);
    hipLaunchKernelGGL(label_class_stats, dim3(nb), dim3(256), 0, bfm_s(stream), G, syn, n, cerebral, partials);
    hipLaunchKernelGGL(label_class_fold, dim3(1), dim3(256), 0, bfm_s(stream), partials, nb, stats);
    return bfm_launch_status();
}

extern "C" int bfm_onehot_lut(const int32_t* S, const int32_t* lut, int nlut, int n_labels, int64_t n, float* out,
                              bfm_stream_t stream) {
    if (!S || !lut || !out || nlut <= 0 || n_labels <= 0 || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(onehot_lut, dim3(grid_for(n * n_labels)), dim3(256), 0, bfm_s(stream), S, lut, nlut, n_labels, n,
                       out);
    return bfm_launch_status();
}

extern "C" int bfm_grid_push3d_linear(const float* inp, int Bi, int C, int ix, int iy, int iz, const float* grid, int Bg,
                                      int nx, int ny, int nz, const int* bound, int extrapolate, float* out_zeroed,
                                      bfm_stream_t stream) {
    if (!inp || !grid || !out_zeroed || !bound || Bi <= 0 || Bg <= 0 || C <= 0 || ix <= 0 || iy <= 0 || iz <= 0 ||
        nx <= 0 || ny <= 0 || nz <= 0)
        return BFM_E_ARG;
    if (Bi != Bg && Bi != 1 && Bg != 1) return BFM_E_SHAPE;
    for (int a = 0; a < 3; ++a) if (bound[a] < 0 || bound[a] > 6) return BFM_E_ARG;
    if (extrapolate < 0 || extrapolate > 2) return BFM_E_ARG;
    const int B = Bi > Bg ? Bi : Bg;
    const int64_t nin = (int64_t)ix * iy * iz;
    hipLaunchKernelGGL(grid_push3d, dim3(grid_for(B * nin)), dim3(256), 0, bfm_s(stream), inp, Bi, C, grid, Bg, nin, nx,
                       ny, nz, bound[0], bound[1], bound[2], extrapolate, B, out_zeroed);
    return bfm_launch_status();
}

extern "C" int bfm_grid_grad3d_linear(const float* inp, int Bi, int C, int nx, int ny, int nz, const float* grid,
                                      int Bg, int ox, int oy, int oz, const int* bound, int extrapolate, float* out,
                                      bfm_stream_t stream) {
    if (!inp || !grid || !out || !bound || Bi <= 0 || Bg <= 0 || C <= 0 || nx <= 0 || ny <= 0 || nz <= 0 || ox <= 0 ||
        oy <= 0 || oz <= 0)
        return BFM_E_ARG;
    if (Bi != Bg && Bi != 1 && Bg != 1) return BFM_E_SHAPE;
    for (int a = 0; a < 3; ++a) if (bound[a] < 0 || bound[a] > 6) return BFM_E_ARG;
    if (extrapolate < 0 || extrapolate > 2) return BFM_E_ARG;
    const int B = Bi > Bg ? Bi : Bg;
    const int64_t nout = (int64_t)ox * oy * oz;
    hipLaunchKernelGGL(grid_grad3d, dim3(grid_for(B * nout)), dim3(256), 0, bfm_s(stream), inp, Bi, C, nx, ny, nz, grid,
                       Bg, nout, bound[0], bound[1], bound[2], extrapolate, B, out);
    return bfm_launch_status();
}
