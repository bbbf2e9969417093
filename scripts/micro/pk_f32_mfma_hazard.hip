// Self-contained reproducer of the packed-FP32 / MFMA hazard on gfx950 (profiles/r06_hazard_root_cause.txt; HISTORY.md section 3.3):
// no library, no torch -- one victim kernel, one aggressor kernel, a host loop.
//
//   victim   : grid_pull3d below, copied verbatim from brainfm_amd/csrc/synth_interp.hip (trilinear pull, interpol iso1.pull3d).
//              Compiled WITH packed-FP32 instructions (hipcc's default) the block that forms the eight corner weights holds
//                  v_cvt_f32_i32_e32 v20, v22 ; v_pk_mul_f32 v[22:23], v[36:37], v[32:33] op_sel:[0,1] op_sel_hi:[0,1]
//                  v_cvt_f32_i32_e32 v12, v24 ; v_pk_mul_f32 v[24:25], v[32:33], v[36:37] op_sel:[0,1] op_sel_hi:[0,1]
//              -- packed multiplies whose two source pairs share VGPR banks.  Their LOW halves (the weights of corners 010 and
//              100) are lost in lanes 48..63 of a wave when the aggressor shares the compute unit: the output is the exact
//              trilinear sum minus one corner's term.
//   aggressor: a dense chain of v_mfma_f32_32x32x16_f16 on six accumulators, ~215 VGPRs and 76.8 KB of LDS per workgroup
//              (two workgroups per CU, two waves per SIMD -- the shape of conv_wino4d's tap loop, nothing else of it).
//              Mode bits: 1 = global_load_dwordx4 of the next B operands into VGPRs, 2 = ds_read_b128 of the A operands,
//              4 = the MFMAs.  4 alone is enough.
//   host     : the victim on two streams, six aggressor launches on a third, all three confined to the same half of the CUs
//              (hipExtStreamCreateWithCUMask) so that victim and aggressor waves share SIMDs; every round's output is compared
//              bit for bit with a quiet run of the same kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o scripts/micro/pk_f32_mfma_hazard scripts/micro/pk_f32_mfma_hazard.hip
//   ./scripts/micro/pk_f32_mfma_hazard [rounds [mask|nomask [fix [lines]]]]   # packed build: wrong rounds > 0 beside every aggressor with MFMAs
//   hipcc ... -Xclang -target-feature -Xclang -packed-fp32-ops ...   # the same source without packed FP32: 0 wrong rounds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float ld_tex(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------ the victim: verbatim from brainfm_amd/csrc/synth_interp.hip
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

template <int FIX>
__global__ void grid_pull3d(const float* __restrict__ inp, int Bi, int C, int nx, int ny, int nz,
                            const float* __restrict__ grid, int Bg, int64_t nout, int bx, int by, int bz, int extrap,
                            int B, float* __restrict__ out) {
    if (FIX) { bx = by = bz = 0; extrap = 0; }            // variant: the 2 000 instructions of the bound switch compile away
#ifdef VICTIM_VGPRS_128
    asm volatile("v_mov_b32 v127, 0" ::: "v127");          // mapping experiment: the victim wave is allocated 128 registers
#endif
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


// ------------------------------------------------------------------ the aggressor
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
#ifndef BALLAST_N
#define BALLAST_N 6
#endif
#ifndef LDS_BYTES
#define LDS_BYTES 76800
#endif
constexpr int BALLAST = BALLAST_N;        // -DBALLAST_N=0: ~120 VGPRs instead of ~215 (mapping experiments, profiles/r06_hazard_root_cause.txt item 13)

__global__ void __launch_bounds__(256, 2) aggressor(const uint4* __restrict__ w, int nfrag, int steps, int mode,
                                                    float* __restrict__ sink) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = w[(blockIdx.x * 64 + i) % (nfrag * 64)];
    __syncthreads();
    float16v acc[6];
    for (int j = 0; j < 6; ++j) acc[j] = float16v{0};
    float16v ballast[BALLAST];                                   // registers held live: two waves fill a SIMD's register file
    for (int j = 0; j < BALLAST; ++j)
        for (int i = 0; i < 16; ++i) ballast[j][i] = (float)(lane + i + j);
    uint4 q[2][6];
    const uint4* wb = w + (size_t)((blockIdx.x * 131 + wave * 17) % (nfrag - 6)) * 64 + lane;
    for (int f = 0; f < 6; ++f) q[0][f] = q[1][f] = wb[f * 64];
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        for (int j = 0; j < BALLAST; ++j) asm volatile("" : "+v"(ballast[j]));
        if (mode & 1) {
            const uint4* nx = w + (size_t)((blockIdx.x * 131 + wave * 17 + (s + 1) * 6) % (nfrag - 6)) * 64 + lane;
#pragma unroll
            for (int f = 0; f < 6; ++f) q[cur ^ 1][f] = nx[f * 64];
        }
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            half8 a0;
            if (mode & 2) a0 = *reinterpret_cast<const half8*>(lds + ((s * 6 + f) & 15) * 1024 + lane * 16);
            else a0 = __builtin_bit_cast(half8, q[cur][(f + 1) % 6]);
            const half8 b = __builtin_bit_cast(half8, q[cur][f]);
            if (mode & 4) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[f], 0, 0, 0);
                acc[(f + 1) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[(f + 1) % 6], 0, 0, 0);
                acc[(f + 2) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[(f + 2) % 6], 0, 0, 0);
            } else {
                acc[f][0] += (float)a0[0] + (float)b[1];
            }
        }
    }
    float t = 0.f;
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 16; ++i) t += acc[j][i];
    for (int j = 0; j < BALLAST; ++j) {
        asm volatile("" : "+v"(ballast[j]));
        for (int i = 0; i < 16; ++i) t += ballast[j][i];
    }
    if (t == 12345.678f) sink[0] = t;
}

static int make_stream(hipStream_t* s, bool masked) {
    if (!masked) return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    uint32_t words[8] = {0};
    for (int i = 0; i < 256; ++i)
        if ((i / 8) % 2 == 0) words[i / 32] |= 1u << (i % 32);     // shader engines 0 and 2 of every XCD
    return (int)hipExtStreamCreateWithCUMask(s, 8, words);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    const bool masked = !(argc > 2 && !strcmp(argv[2], "nomask"));
    // the victim's problem: a (2, 2, 5, 6, 7) volume pulled at 2 x (64 x 5 x 6) coordinates (15 workgroups of 256 threads)
    const int Bn = 2, Cn = 2, nx = 5, ny = 6, nz = 7, ox = 64, oy = 5, oz = 6;
    const int64_t nout = (int64_t)ox * oy * oz, nvol = (int64_t)Bn * Cn * nx * ny * nz;
    std::vector<float> hv(nvol), hg(Bn * nout * 3);
    unsigned x = 2463534242u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return (float)(x >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : hv) v = rnd() * 2.f - 1.f;
    for (int64_t i = 0; i < Bn * nout; ++i) {
        hg[i * 3] = rnd() * (nx + 0.6f) - 0.8f; hg[i * 3 + 1] = rnd() * (ny + 0.6f) - 0.8f; hg[i * 3 + 2] = rnd() * (nz + 0.6f) - 0.8f;
    }
    float *vol, *grid[2], *out[2], *ref, *sink;
    uint4* w;
    const int nfrag = 64 * 1024;
    CK(hipMalloc(&vol, nvol * 4)); CK(hipMalloc(&ref, Bn * Cn * nout * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&w, (size_t)nfrag * 1024)); CK(hipMemset(w, 0x3c, (size_t)nfrag * 1024));
    CK(hipMemcpy(vol, hv.data(), nvol * 4, hipMemcpyHostToDevice));
    for (int l = 0; l < 2; ++l) {
        CK(hipMalloc(&grid[l], Bn * nout * 12)); CK(hipMalloc(&out[l], Bn * Cn * nout * 4));
        CK(hipMemcpy(grid[l], hg.data(), Bn * nout * 12, hipMemcpyHostToDevice));
    }
    CK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipStream_t s[3];
    for (int i = 0; i < 3; ++i) CK((hipError_t)make_stream(&s[i], masked));
    const int nb = (int)((Bn * nout + 255) / 256);
    const int fix = argc > 3 ? atoi(argv[3]) : 0;
    auto pull = [&](hipStream_t st, float* g, float* o) {
        if (fix) hipLaunchKernelGGL(grid_pull3d<1>, dim3(nb), dim3(256), 0, st, vol, Bn, Cn, nx, ny, nz, g, Bn, nout, 0, 0, 0, 0, Bn, o);
        else hipLaunchKernelGGL(grid_pull3d<0>, dim3(nb), dim3(256), 0, st, vol, Bn, Cn, nx, ny, nz, g, Bn, nout, 0, 0, 0, 0, Bn, o);
    };
    pull(s[0], grid[0], ref);                                      // the quiet run: the reference bits
    CK(hipDeviceSynchronize());
    std::vector<float> href(Bn * Cn * nout), hout(Bn * Cn * nout);
    CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
    printf("# victim: grid_pull3d, %d workgroups, on two streams; aggressor: 6 launches of 1024 workgroups per round on a third;\n"
           "# %s; %d rounds per line; a wrong round = at least one output element differs from the quiet run\n",
           nb, masked ? "all three streams on the same half of the CUs" : "no CU masks", rounds);
    const int modes[6] = {-1, 7, 4, 5, 6, 3};
    const int nmodes = argc > 4 ? atoi(argv[4]) : 6;              // 3: none, the full tap loop, the MFMA chain alone
    const char* mname[8] = {"", "", "", "loads + LDS reads, NO MFMA", "MFMA chain only", "MFMA + global loads -> VGPR", "MFMA + LDS operand reads",
                            "MFMA + global loads + LDS reads"};
    printf("# victim variant: %s\n", fix ? "bounds fixed at compile time (zero bound, no extrapolation)" : "as in the library (run-time bounds)");
    for (int mi = 0; mi < nmodes; ++mi) {
        const int mode = modes[mi];
        long wrong_rounds = 0, wrong_el = 0, lanes_hi = 0, lanes_lo = 0;
        for (int r = 0; r < rounds; ++r) {
            for (int l = 0; l < 2; ++l) CK(hipMemsetAsync(out[l], 0xff, Bn * Cn * nout * 4, 0));
            CK(hipDeviceSynchronize());
            if (mode >= 0)
                for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), LDS_BYTES, s[2], w, nfrag, 400, mode, sink);
            for (int l = 0; l < 2; ++l) pull(s[l], grid[l], out[l]);
            CK(hipDeviceSynchronize());
            bool bad = false;
            for (int l = 0; l < 2; ++l) {
                CK(hipMemcpy(hout.data(), out[l], hout.size() * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hout.size(); ++i)
                    if (memcmp(&hout[i], &href[i], 4)) {
                        bad = true; ++wrong_el;
                        const int64_t b = i / (Cn * nout), v = i % nout, thread = b * nout + v;
                        if ((thread & 63) >= 48) ++lanes_hi; else ++lanes_lo;
                    }
            }
            wrong_rounds += bad;
        }
        printf("aggressor: %-34s wrong rounds %4ld of %d, wrong elements %6ld (lanes 48..63: %ld, lanes 0..47: %ld)\n",
               mode < 0 ? "none" : mname[mode], wrong_rounds, rounds, wrong_el, lanes_hi, lanes_lo);
        fflush(stdout);
    }
    return 0;
}
