// The x2 synthesis layer in ONE kernel (fp16 storage, fp32 accumulation), gfx950:
//     conv_transpose2d(stride 2)  ->  4x4 low-pass (pad 1, gain 4)  ->  + noise  ->  + bias  ->  lrelu * gain  ->  clamp
// (reference: torch_utils/ops/conv2d_resample.py:114-131 for the first two, training/networks_stylegan2.py:319-332 for the rest).
// The two-kernel form (convT_h2_f16_kernel + fir4_cl_fused_kernel) writes the (2H+1)^2 transposed-conv image to HBM and reads it
// back: 269 MB each way for the 256 -> 128 channel, 512^2 layer of the SR heads, more time in the FIR than in the matrix cores.
//
// Work split.  A block owns a 28 x 28 tile of FINAL pixels x 32 output channels.  The FIR needs transposed-conv rows
// 28 ty - 1 .. 28 ty + 29, i.e. the 16 x 16 patch of class positions starting at 14 ty - 1 in EACH of the four output-parity
// classes (class (py, px), position (i, j) = transposed-conv pixel (2 i + py, 2 j + px)); positions outside the image evaluate to
// zero by themselves because every tap then reads the zero page, which is exactly the FIR's zero padding.  (16 / 14)^2 = 1.31x
// recomputation is the price of never storing the intermediate.
//
// All four classes are accumulated AT ONCE: the nine taps of the 3x3 kernel fall into the classes 4 / 2 / 2 / 1, every tap's input
// offset is in {-1, 0}^2, so one 17 x 17 halo slab per 32-channel chunk feeds all of them and the K loop is the plain 3x3 loop of
// conv3x3_h2_f16_kernel with the accumulator chosen by the tap: 4 waves x (64 positions x 32 channels x 4 classes) = 128 accumulator
// VGPRs, 36 MFMAs per wave and chunk.  32 output channels per block is what makes the four classes fit; the weights of a chunk are
// 18 KB, re-filled a third at a time (three barriers per chunk, each a full chunk ahead of its use), the slab is double-buffered:
// 62 KB of LDS, two blocks per CU, so one block's FIR (VALU) runs under the other's K loop (MFMA).
//
// Epilogue: the four class tiles go to LDS as ONE fp16 image [32][32][32 ch] (the rounding the reference's fp16 conv_transpose2d
// output has), then each thread slides the separable 4 + 4 tap filter down a 14-row strip for 8 channels (16-byte LDS reads, fp32
// arithmetic) and applies noise / bias / activation / clamp on the way out.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include "p3d_common.h"
#include "../../include/p3d_hip.h"

namespace p3d {

typedef _Float16 uh8 __attribute__((ext_vector_type(8)));
typedef float uf16 __attribute__((ext_vector_type(16)));
typedef float uf4 __attribute__((ext_vector_type(4)));
typedef float uf2 __attribute__((ext_vector_type(2)));

constexpr int UF_PITCH = 20;                                    // slab pixels per slab row (a multiple of 4: see conv3x3_h2_f16_kernel)
constexpr int UF_SLAB_ROWS = 17 * UF_PITCH;                     // 340 pixel rows of 64 bytes
constexpr int UF_SLAB_PIECES = (UF_SLAB_ROWS + 15) / 16;        // 22 DMA pieces of 1 KB
constexpr int UF_SLAB_BUF = UF_SLAB_PIECES * 1024;              // 22528
constexpr int UF_BN = 32;                                       // output channels per block
constexpr int UF_TAP = UF_BN * 64;                              // bytes of one tap's weight tile (32 rows x 32 channels)
constexpr int UF_WT_BASE = 2 * UF_SLAB_BUF;                     // 45056
constexpr int UF_TILE = 28;                                     // final pixels per tile side
constexpr int UF_CP = 32 * 32 + 8;                              // halfs per channel of the channel-major tile image (2064 B: 16 channels = 16 bank quads)
constexpr int UF_CT_BYTES = UF_BN * UF_CP * 2;                  // 66048: the transposed-conv tile, fp16 (the pixel-major image of the VALU epilogue needs 65536)
constexpr int UF_NZ_FLOATS = UF_TILE * UF_TILE + 16;            // the tile's noise (+ slack: lanes 28..31 of a row read past their row)
constexpr int UF_LDS = UF_CT_BYTES + (UF_NZ_FLOATS + UF_BN) * 4;  // + the bias row = 69376 (>= 45056 + 9 * 2048); two blocks per CU

struct Up2Args {
    const void* x;         // [N][H][W][Ci] fp16
    const void* w;         // [N or 1][Co][9][Ci] fp16 (modulated, tap-major: what p3d_conv2d_nhwc takes for resample = 1)
    void* y;               // [N][2H][2W][Co] fp16
    const void* zeros;     // >= 128 zero bytes
    const float* bias;     // [Co] or null
    const float* noise;    // [2H][2W] or null
    const float* noise_strength;
    int N, H, W, Ci, Co;
    int64_t w_img_stride;
    float conv_gain;       // on the transposed conv's output, before the fp16 rounding
    float fy[4], fx[4];    // separable FIR, already in correlation order and with its gain: out[o] = sum_k f[k] * ct[o - 1 + k]
    int act;               // 0 linear, 1 lrelu(0.2)
    float act_gain, clamp; // clamp < 0: off
    int tiles_x, tiles;    // 28 x 28 tiles per row / per image
    int debug;             // measurement only (P3D_UP2_DEBUG): 1 stop after the K loop, 2 skip the K loop, 4 no stores, 8 no FIR, 16 no tile writes
};

template <int N> __device__ __forceinline__ void uf_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// slot s of a chunk's nine weight tiles holds tap kTapW[s] (ky * 3 + kx); phase p = slots 3p .. 3p+2.  The tap's class is
// (ky & 1) * 2 + (kx & 1) and its input offset (-(ky >> 1), -(kx >> 1)); phase 0 reads offset (0,0) only, phase 1 (0,0) and (-1,0),
// phase 2 (0,-1) and (-1,-1).
__device__ constexpr int kTapW[9]   = {0, 1, 3,   4, 6, 7,   2, 5, 8};
__device__ constexpr int kTapCls[9] = {0, 1, 2,   3, 0, 1,   0, 2, 0};
__device__ constexpr int kTapA[9]   = {0, 0, 0,   0, 1, 1,   0, 0, 1};      // which of the phase's (up to two) A offsets
__device__ constexpr int kPhaseDy[3][2] = {{0, 0}, {0, -1}, {0, -1}};
__device__ constexpr int kPhaseDx[3][2] = {{0, 0}, {0, 0}, {-1, -1}};
__device__ constexpr int kPhaseNA[3] = {1, 2, 2};

// MFMA_FIR: the 4 x 4 FIR of the epilogue on the matrix cores (below); false = the vector-ALU epilogue, for filters whose tap products are
// not fp16 numbers.
template <bool MFMA_FIR>
__global__ void __launch_bounds__(256, 2) up2_fir_f16_kernel(Up2Args a)
{
    __shared__ __attribute__((aligned(16))) char lds_b[UF_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.z;
    const int ncb = a.Co / UF_BN;
    int mt, cb;
    {   // consecutive block ids go round the 8 XCDs: ids 8q .. 8q+7 are eight tiles of one channel block, 8(q+1).. the same tiles of
        // the next channel block — an XCD's L2 serves a tile's slab to all of its channel blocks
        const int L = blockIdx.x, q = L >> 3, r = L & 7;
        cb = q % ncb; mt = (q / ncb) * 8 + r;
    }
    if (mt >= a.tiles) return;
    const int ty = mt / a.tiles_x, tx = mt - ty * a.tiles_x;
    const int cy0 = 14 * ty - 1, cx0 = 14 * tx - 1, co0 = cb * UF_BN;           // class-grid origin of the 16 x 16 patch
    const char* const xin_b = (const char*)((const __half*)a.x + (int64_t)n * a.H * a.W * a.Ci);
    const char* const wgt_b = (const char*)((const __half*)a.w + (int64_t)n * a.w_img_stride);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int pos = lane & 3, prow = lane >> 2;                                 // DMA lane: 16-byte position / row within a 16-row piece
    const int kchunks = a.Ci / 32;
    const bool two = wave < 2;                                                  // waves 0, 1 carry the extra weight piece and slab piece

    // ---- DMA sources --------------------------------------------------------------------------------------------------------
    unsigned woff;                                                              // this wave stages rows 16 (wave & 1) .. +15 of a tap tile
    {
        const int row = (wave & 1) * 16 + prow;
        woff = (unsigned)(((co0 + row) * 9 * a.Ci + (pos ^ ((row >> 2) & 3)) * 8) * 2);
    }
    unsigned soff[6]; bool sok[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int q = (wave + 4 * g) * 16 + prow;
        const int sy = q / UF_PITCH, sx = q - sy * UF_PITCH;
        const int iy = cy0 - 1 + sy, ix = cx0 - 1 + sx;
        sok[g] = (q < UF_SLAB_ROWS) & (sx < 17) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
        soff[g] = (unsigned)(((iy * a.W + ix) * a.Ci + (pos ^ ((sx >> 2) & 3)) * 8) * 2);
    }
    auto stage_slab = [&](int cc) {
        const int buf = (cc & 1) * UF_SLAB_BUF;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g < 5 || two) {
                const char* src = sok[g] ? xin_b + soff[g] + cc * 64 : (const char*)a.zeros;
                __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(lds_b + buf + (wave + 4 * g) * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_w = [&](int cc, int p) {                                         // p is a compile-time constant at every call
        const int half = (wave & 1) * 1024;
        const int j0 = wave >> 1;                                               // piece `wave`: tap j0 of the phase, rows of `half`
        const int t0 = j0 ? kTapW[p * 3 + 1] : kTapW[p * 3];
        __builtin_amdgcn_global_load_lds((glb_ptr)(wgt_b + woff + (t0 * a.Ci + cc * 32) * 2),
                                         (lds_ptr)(lds_b + UF_WT_BASE + (p * 3 + j0) * UF_TAP + half), 16, 0, 0);
        if (two)                                                                // piece wave + 4: the phase's third tap
            __builtin_amdgcn_global_load_lds((glb_ptr)(wgt_b + woff + (kTapW[p * 3 + 2] * a.Ci + cc * 32) * 2),
                                             (lds_ptr)(lds_b + UF_WT_BASE + (p * 3 + 2) * UF_TAP + half), 16, 0, 0);
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    const int frow = lane & 31, fk = lane >> 5, acol = frow & 15;
    int preA[2][2], preB[2];                                                    // [1 + dx][kk]
#pragma unroll
    for (int tx2 = 0; tx2 < 2; ++tx2)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            preA[tx2][kk] = ((wave * 4 + (frow >> 4)) * UF_PITCH + acol + tx2) * 64 + (((kk * 2 + fk) ^ (((acol + tx2) >> 2) & 3)) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) preB[kk] = UF_WT_BASE + frow * 64 + (((kk * 2 + fk) ^ ((frow >> 2) & 3)) << 4);

    uf4 fa[2][2][2], fb[2][3];                                                  // [register buffer][A offset of the phase][i], [..][tap of the phase]
    // loads of micro-step (p, kk) of the chunk whose slab sits at `sbuf`; p, kk, reg are compile-time constants
#define UF_LOAD(p, kk, reg, sbuf)                                                                                                        \
    do {                                                                                                                                \
        _Pragma("unroll") for (int ao = 0; ao < kPhaseNA[p]; ++ao) {                                                                    \
            const int base = preA[1 + kPhaseDx[p][ao]][kk] + (sbuf);                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[reg][ao][i]) : "v"(base), "n"((2 * i + 1 + kPhaseDy[p][ao]) * UF_PITCH * 64) : "memory"); \
        }                                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                                   \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[reg][j]) : "v"(preB[kk]), "n"(((p) * 3 + j) * UF_TAP) : "memory");   \
    } while (0)
#define UF_MFMA(p, reg)                                                                                                                  \
    do {                                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
                acc[kTapCls[(p) * 3 + j]][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(uh8, fa[reg][kTapA[(p) * 3 + j]][i]), \
                                                                                      __builtin_bit_cast(uh8, fb[reg][j]), acc[kTapCls[(p) * 3 + j]][i], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                                              \
    } while (0)

    uf16 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

    // the epilogue's noise tile and bias: loaded NOW, into five registers, so that their global-memory latency passes under the K loop (as part
    // of the epilogue it was ~2 us of every block's life with nothing to hide it)
    float nz_pre[4], bias_pre = 0.f, ns_pre = 0.f;                                // (+ the noise strength: read behind the K loop it was one exposed memory round trip of every block's ~25 us)
    if constexpr (MFMA_FIR) {
        const int OH = 2 * a.H, OW = 2 * a.W;
        if (a.noise) ns_pre = a.noise_strength[0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            const int oy = UF_TILE * ty + e / UF_TILE, ox = UF_TILE * tx + e % UF_TILE;
            nz_pre[q] = (a.noise && e < UF_TILE * UF_TILE && oy < OH && ox < OW) ? a.noise[(int64_t)oy * OW + ox] : 0.f;
        }
        if (a.bias && tid < UF_BN) bias_pre = a.bias[co0 + tid];
    }

    // ---- K loop --------------------------------------------------------------------------------------------------------------
    // DMA groups of a wave: W = 2 (waves 0, 1) or 1 pieces, S = 6 or 5.  At the rendezvous that ends phase p of chunk c everything
    // but the NEWEST group must have landed: after phase 0 that is W(c, 2) [-> W(c, 1) is in], after phase 1 W(c+1, 0) + slab(c+1)
    // [-> W(c, 2) is in], after phase 2 W(c+1, 1) [-> W(c+1, 0) and the slab are in]; then the phase's slots take chunk c+1's taps.
    stage_slab(0);
    stage_w(0, 0);
    stage_w(0, 1);
    stage_w(0, 2);
    if (two) uf_wait_vmcnt<4>(); else uf_wait_vmcnt<2>();                        // slab 0 and phase 0's taps
    __builtin_amdgcn_s_barrier();
    UF_LOAD(0, 0, 0, 0);
    for (int cc = 0; cc < ((a.debug & 2) ? 0 : kchunks); ++cc) {
        const bool more = cc + 1 < kchunks;
        const int sb = (cc & 1) * UF_SLAB_BUF, sn = UF_SLAB_BUF - sb;
        // phase 0
        UF_LOAD(0, 1, 1, sb);
        asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
        UF_MFMA(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (two) uf_wait_vmcnt<2>(); else uf_wait_vmcnt<1>();
        __builtin_amdgcn_s_barrier();
        if (more) { stage_w(cc + 1, 0); stage_slab(cc + 1); }
        UF_LOAD(1, 0, 0, sb);
        UF_MFMA(0, 1);
        // phase 1
        UF_LOAD(1, 1, 1, sb);
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        UF_MFMA(1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) { if (two) uf_wait_vmcnt<8>(); else uf_wait_vmcnt<6>(); }
        else uf_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (more) stage_w(cc + 1, 1);
        UF_LOAD(2, 0, 0, sb);
        UF_MFMA(1, 1);
        // phase 2
        UF_LOAD(2, 1, 1, sb);
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        UF_MFMA(2, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) {
            if (two) uf_wait_vmcnt<2>(); else uf_wait_vmcnt<1>();
            __builtin_amdgcn_s_barrier();
            stage_w(cc + 1, 2);
            UF_LOAD(0, 0, 0, sn);
        }
        UF_MFMA(2, 1);
    }
#undef UF_LOAD
#undef UF_MFMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uf_wait_vmcnt<0>();
    if (a.debug & 1) { if (acc[0][0][0] == 123.f) ((float*)a.y)[0] = 1.f; return; }
    __syncthreads();                                                            // slabs and taps are dead: the LDS becomes the epilogue's

    const int OH = 2 * a.H, OW = 2 * a.W;
    const int oy0 = UF_TILE * ty, ox0 = UF_TILE * tx;
    float* const nz = (float*)(lds_b + UF_CT_BYTES);
    if constexpr (!MFMA_FIR) {
        const float ns = a.noise ? a.noise_strength[0] : 0.f;
        for (int e = tid; e < UF_NZ_FLOATS; e += 256) {
            const int oy = oy0 + e / UF_TILE, ox = ox0 + e % UF_TILE;
            nz[e] = (a.noise && e < UF_TILE * UF_TILE && oy < OH && ox < OW) ? a.noise[(int64_t)oy * OW + ox] * ns : 0.f;
        }
    }
    if constexpr (MFMA_FIR) {
        // everything a finished value still needs, folded: y = clamp(max(v, slope v)) with v = (acc + noise + bias) * act_gain
        //   = fma(acc, act_gain, (noise * strength + bias) * act_gain); the LDS noise tile and bias row hold the second term's pieces pre-scaled
        const float slope = (a.act == 1) ? 0.2f : 1.f, lim = (a.clamp >= 0.f) ? a.clamp : INFINITY;
        float* const bs = nz + UF_NZ_FLOATS;                                    // [32] bias * act_gain
        {
            const float ns = a.noise ? ns_pre * a.act_gain : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (tid + 256 * q < UF_NZ_FLOATS) nz[tid + 256 * q] = nz_pre[q] * ns;
            if (tid < UF_BN) bs[tid] = bias_pre * a.act_gain;
        }
        // ---- the FIR on the matrix cores --------------------------------------------------------------------------------------
        // out[m][xo] = sum_ky sum_kx fy[ky] fx[kx] ct[1 + m + ky][1 + xo + kx] is, for one output row m and one tap row ky, a product of the tile row
        // ct[1 + m + ky][.] (M = 32 channels x K = 32 columns) with the BANDED constant matrix B_ky[xin][xo] = fy[ky] fx[xin - xo - 1]: eight
        // 32x32x16 MFMAs per output row and 32 channels, fp16 inputs (the tile is fp16 anyway, the tap products are fp16 numbers — host check),
        // fp32 accumulation.  The vector-ALU version of this epilogue was the kernel's busiest pipe (~2 800 VALU instructions per thread, 408
        // conversions, 128 two-byte LDS stores: profiles/round3_g_kernel_pmc_infer_train.txt); here a thread issues 56 MFMAs and ~1 100 VALU.
        // The tile image is CHANNEL-major [32 ch][32 rows][32 columns] (pitch 2064 B), so
        //   * the accumulators leave as 16-byte LDS stores: a lane's 4 consecutive accumulator rows are 4 consecutive class columns of one
        //     channel, and the two x-parity classes interleave them into 8 consecutive tile columns;
        //   * an A fragment (channel = lane & 31, eight consecutive columns) is one ds_read_b128;
        //   * 16 consecutive channels start on 16 different bank quads for both.
        __half* const ct = (__half*)lds_b;
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uh8 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e]     = (_Float16)(acc[py * 2 + 0][i][q * 4 + e] * a.conv_gain);
                        v[2 * e + 1] = (_Float16)(acc[py * 2 + 1][i][q * 4 + e] * a.conv_gain);
                    }
                    const int row = 2 * (wave * 4 + i * 2 + (q >> 1)) + py, col = 16 * (q & 1) + 8 * fk;
                    *(uh8*)(ct + frow * UF_CP + row * 32 + col) = v;
                }
        // B fragments: lane (xo = lane & 31, K group fk) holds B_ky[xin = 16 kh + 8 fk + e][xo], e = 0 .. 7
        uh8 bf[4][2];
        {
            float band[2][8];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kx = kh * 16 + fk * 8 + e - frow - 1;
                    band[kh][e] = (frow < UF_TILE) ? (kx == 0 ? a.fx[0] : kx == 1 ? a.fx[1] : kx == 2 ? a.fx[2] : kx == 3 ? a.fx[3] : 0.f) : 0.f;
                }
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bf[ky][kh][e] = (_Float16)(a.fy[ky] * band[kh][e]);
        }
        // what a lane finishes: pixel column xo = lane & 31 of an output row, channels (r & 3) + 8 (r >> 2) + 4 fk
        const int ox = ox0 + frow;
        const bool col_ok = frow < UF_TILE && ox < OW;
        __half* const yout = (__half*)a.y + (int64_t)n * OH * OW * a.Co + co0 + 8 * fk;
        __syncthreads();
        if (a.debug & 8) return;
        float bias[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uf4 b4 = *(const uf4*)(bs + 8 * q + 4 * fk);
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[q * 4 + e] = b4[e];
        }
        // wave w: output rows 7 w .. 7 w + 6 of the tile from tile rows 7 w + 1 .. 7 w + 10; four rows in flight, one accumulator each (a row's
        // first MFMA starts from the constant 0)
        uf16 o[4];
        uf16 zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
        const __half* const arow = ct + frow * UF_CP + (wave * 7 + 1) * 32 + fk * 8;
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const uh8 a0 = *(const uh8*)(arow + t * 32), a1 = *(const uh8*)(arow + t * 32 + 16);
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
                if (t - ky >= 0 && t - ky < 7) o[(t - ky) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf[ky][0], ky == 0 ? zero16 : o[(t - ky) & 3], 0, 0, 0);
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
                if (t - ky >= 0 && t - ky < 7) o[(t - ky) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf[ky][1], o[(t - ky) & 3], 0, 0, 0);
            if (t >= 3) {                                                       // output row t - 3 is complete
                const int m = wave * 7 + t - 3, oy = oy0 + m, s = (t - 3) & 3;
                const float nzv = nz[m * UF_TILE + frow];
                unsigned pk[8];                                                 // [q][2]: channels 8 q + 4 fk + 0 .. 3 as two half pairs
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        float v2[2];
#pragma unroll
                        for (int z = 0; z < 2; ++z) {
                            float v = fmaf(o[s][q * 4 + e2 * 2 + z], a.act_gain, nzv + bias[q * 4 + e2 * 2 + z]);
                            v = fmaxf(v, slope * v);
                            v2[z] = __builtin_amdgcn_fmed3f(v, -lim, lim);
                        }
                        typedef _Float16 uh2 __attribute__((ext_vector_type(2)));
                        uh2 hv; hv[0] = (_Float16)v2[0]; hv[1] = (_Float16)v2[1];
                        pk[q * 2 + e2] = __builtin_bit_cast(unsigned, hv);
                    }
                // the fk = 0 half-wave holds channels {0-3, 8-11, 16-19, 24-27}, fk = 1 {4-7, ...}: trade q = 1, 3 of the lower half against
                // q = 0, 2 of the upper one and every lane owns 8 consecutive channels twice: 16-byte stores, a pixel's 64 bytes from two lanes
                uf4 st[2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const unsigned mine_lo = pk[(2 * pr) * 2 + e2], mine_hi = pk[(2 * pr + 1) * 2 + e2];
                        const unsigned send = fk ? mine_lo : mine_hi;
                        const unsigned recv = (unsigned)__shfl_xor((int)send, 32, 64);
                        st[pr][e2]     = __uint_as_float(fk ? recv : mine_lo);
                        st[pr][2 + e2] = __uint_as_float(fk ? mine_hi : recv);
                    }
                if (col_ok && oy < OH && !(a.debug & 4)) {
                    __half* dst = yout + ((int64_t)oy * OW + ox) * a.Co;
                    *(uf4*)dst = st[0];
                    *(uf4*)(dst + 16) = st[1];
                }
            }
        }
        return;
    }
    // ---- the four class tiles -> one fp16 image [32][32][32 ch] --------------------------------------------------------------
    __half* const ct = (__half*)lds_b;
    if (!(a.debug & 16))
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                const int row = 2 * (p >> 4) + py, col = 2 * (p & 15) + px;
                ct[(row * 32 + col) * UF_BN + frow] = __float2half(acc[c][i][r] * a.conv_gain);
            }
    }
    __syncthreads();

    // ---- separable 4-tap FIR down a 14-row strip, 8 channels per thread ------------------------------------------------------
    const int slot = tid >> 2, chunk = tid & 3;
    const int strip = slot / UF_TILE, xo = slot - strip * UF_TILE;
    if (strip >= 2) return;
    const int ox = ox0 + xo;
    if (ox >= OW) return;
    float bias[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias[k] = a.bias ? a.bias[co0 + chunk * 8 + k] : 0.f;
    __half* const yout = (__half*)a.y + (int64_t)n * OH * OW * a.Co + co0 + chunk * 8;
    float out[4][8];                                                            // rows m .. m+3 of the strip, rotating
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) out[q][k] = 0.f;
    if (a.debug & 8) return;
#pragma unroll
    for (int j = 0; j < 17; ++j) {                                              // transposed-conv row strip * 14 + 1 + j of the tile
        float h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = 0.f;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const uh8 v = *(const uh8*)(ct + (((strip * 14 + 1 + j) * 32 + xo + 1 + kx) * UF_BN + chunk * 8));
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] = fmaf((float)v[k], a.fx[kx], h[k]);
        }
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int m = j - ky;                                               // the strip row this input row feeds through tap ky
            if (m >= 0 && m < 14) {
#pragma unroll
                for (int k = 0; k < 8; ++k) out[m & 3][k] = fmaf(h[k], a.fy[ky], out[m & 3][k]);
            }
        }
        const int m = j - 3;                                                    // complete now
        if (m >= 0) {
            const int oy = oy0 + strip * 14 + m;
            const float nzv = nz[(strip * 14 + m) * UF_TILE + xo];
            uh8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = out[m & 3][k] + nzv + bias[k];
                if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                v *= a.act_gain;
                if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
                o[k] = (_Float16)v;
                out[m & 3][k] = 0.f;
            }
            if (oy < OH && (!(a.debug & 4) || o[0] == (_Float16)123.f)) *(uh8*)(yout + ((int64_t)oy * OW + ox) * a.Co) = o;
        }
    }
}

} // namespace p3d

extern "C" int p3d_up2_fir_f16(const void* x, const void* w, void* y, const void* zeros128, const float* bias, const float* noise,
                               const float* noise_strength, const float* fir_yx_host, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                               int64_t w_img_stride, float conv_gain, int32_t act, float act_gain, float clamp, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && w && y && zeros128 && fir_yx_host, "up2_fir_f16: null pointer");
    P3D_REQUIRE(n_img >= 1 && h >= 1 && wdt >= 1, "up2_fir_f16: bad sizes");
    P3D_REQUIRE(act == 0 || act == 1, "up2_fir_f16: act must be 0 (linear) or 1 (lrelu)");
    if (ci % 32 != 0 || co % UF_BN != 0) return fail(P3D_ERR_UNSUPPORTED, "up2_fir_f16: Ci=%d and Co=%d must be multiples of 32", ci, co);
    P3D_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)zeros128) & 15u) == 0, "up2_fir_f16: pointers must be 16-byte aligned");
    P3D_REQUIRE((int64_t)h * wdt * ci * 2 < (1ll << 31) && (int64_t)co * 9 * ci * 2 < (1ll << 31), "up2_fir_f16: image or weight panel beyond 32-bit offsets");
    Up2Args a{};
    a.x = x; a.w = w; a.y = y; a.zeros = zeros128; a.bias = bias; a.noise = noise; a.noise_strength = noise_strength;
    a.N = n_img; a.H = h; a.W = wdt; a.Ci = ci; a.Co = co; a.w_img_stride = w_img_stride; a.conv_gain = conv_gain;
    for (int k = 0; k < 4; ++k) { a.fy[k] = fir_yx_host[k]; a.fx[k] = fir_yx_host[4 + k]; }
    a.act = act; a.act_gain = act_gain; a.clamp = clamp;
    const int tiles_y = (2 * h + UF_TILE - 1) / UF_TILE;
    a.tiles_x = (2 * wdt + UF_TILE - 1) / UF_TILE;
    a.tiles = a.tiles_x * tiles_y;
    { static const int dbg = [] { const char* d = getenv("P3D_UP2_DEBUG"); return d ? atoi(d) : 0; }(); a.debug = dbg; }      // phase switches of tests/gpu_probe_up2.py (0 = the layer)
    const int64_t blocks = (int64_t)((a.tiles + 7) / 8 * 8) * (co / UF_BN);
    P3D_REQUIRE(blocks < (1ll << 31) && n_img < 65536, "up2_fir_f16: bad launch size");
    // the matrix-core FIR multiplies the fp16 tile by fy[ky] * fx[kx] as fp16 numbers: exact for setup_filter([1, 3, 3, 1]) with gain 4 ({1, 3, 9} / 16) and
    // for any filter whose 16 tap products are fp16 numbers; others take the vector-ALU epilogue (fp32 taps)
    bool taps_fp16 = true;
    for (int ky = 0; ky < 4; ++ky)
        for (int kx = 0; kx < 4; ++kx) {
            const float p = a.fy[ky] * a.fx[kx];
            taps_fp16 = taps_fp16 && (float)(_Float16)p == p && (p == 0.f || fabsf(p) >= 6.2e-5f);
        }
    static const bool force_valu = [] { const char* d = getenv("P3D_UP2_VALU_FIR"); return d && atoi(d) != 0; }();      // A/B switch of tests/gpu_probe_up2.py
    if (taps_fp16 && !force_valu) hipLaunchKernelGGL(up2_fir_f16_kernel<true>, dim3((unsigned)blocks, 1, n_img), dim3(256), 0, (hipStream_t)stream, a);
    else                          hipLaunchKernelGGL(up2_fir_f16_kernel<false>, dim3((unsigned)blocks, 1, n_img), dim3(256), 0, (hipStream_t)stream, a);
    count_launch(FAM_CONV);
    return check_launch("up2_fir_f16");
}

// ---- the same layer for the fp32 backbone (bf16x3) ---------------------------------------------------------------------------------------
// fp32 tensors, every product as three bf16 MFMAs of (hi, lo) splits (csrc/conv2d.hip, "bf16x3"): activations are split in registers,
// the weights arrive pre-split from p3d_modulate_weights (dtype P3D_F32_BF16X3: K rows of [32 x hi | 32 x lo] per 32 input channels).
// Same decomposition as above — a 28 x 28 tile of final pixels x 32 output channels per block, four parity classes accumulated at once
// from one 17 x 17 halo slab per 32-channel chunk — in the simpler two-buffer form of conv3x3_halo_kernel: a chunk's slab (39 KB of fp32)
// and its nine weight tiles (36 KB) stream in under the previous chunk's 108 MFMAs per wave, one rendezvous per chunk.  The
// transposed-conv tile stays fp32 in LDS (131 KB) for the FIR, so there is ONE block per CU; the fp32 (2H+1)^2 intermediate of the
// two-kernel form (135 MB each way at 256^2 x 128 channels, batch 4) never exists.
namespace p3d {

typedef __bf16 ubf8 __attribute__((ext_vector_type(8)));
constexpr int UB_SW = 18;                                       // slab pixels per slab row (17 used)
constexpr int UB_SLAB_PIECES = (17 * UB_SW + 7) / 8;            // 39 DMA pieces of 8 pixels x 128 B
constexpr int UB_SLAB_SLOTS = UB_SLAB_PIECES * 64;              // 16-byte slots per slab buffer (39936 B)
constexpr int UB_WT_SLOTS = 9 * UF_BN * 8;                      // 16-byte slots of a chunk's nine weight tiles (36864 B)
constexpr int UB_CT_FLOATS = 32 * 32 * UF_BN;                   // 131072 B
constexpr int UB_LDS = (2 * (UB_SLAB_SLOTS + UB_WT_SLOTS) * 16 > UB_CT_FLOATS * 4 + UF_TILE * UF_TILE * 4) ? 2 * (UB_SLAB_SLOTS + UB_WT_SLOTS) * 16
                                                                                                           : UB_CT_FLOATS * 4 + UF_TILE * UF_TILE * 4;

__device__ __forceinline__ void ub_split(const uf4& a0, const uf4& a1, ubf8& hi, ubf8& lo)
{
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));        // on pairs, as split_bf16x8 in conv2d.hip: one conversion instruction per two pieces
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2_t x = p < 2 ? f32x2_t{a0[2 * p], a0[2 * p + 1]} : f32x2_t{a1[2 * p - 4], a1[2 * p - 3]};
        h[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf2_t));
        const f32x2_t hf = {__builtin_bit_cast(float, h[p] << 16), __builtin_bit_cast(float, h[p] & 0xffff0000u)};
        l[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x - hf, bf2_t));
    }
    hi = __builtin_bit_cast(ubf8, h); lo = __builtin_bit_cast(ubf8, l);
    asm volatile("s_nop 4" : "+v"(hi), "+v"(lo));              // conversion -> MFMA operand hazard: see split8 in render_device.h
}

__global__ void __launch_bounds__(256, 1) up2_fir_bf16x3_kernel(Up2Args a)
{
    extern __shared__ __attribute__((aligned(16))) char ub_lds[];
    uf4* const slab = (uf4*)ub_lds;                                              // [2][UB_SLAB_SLOTS]
    uf4* const wt = slab + 2 * UB_SLAB_SLOTS;                                    // [2][UB_WT_SLOTS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.z;
    const int ncb = a.Co / UF_BN;
    int mt, cb;
    { const int L = blockIdx.x, q = L >> 3, r = L & 7; cb = q % ncb; mt = (q / ncb) * 8 + r; }
    if (mt >= a.tiles) return;
    const int ty = mt / a.tiles_x, tx = mt - ty * a.tiles_x;
    const int cy0 = 14 * ty - 1, cx0 = 14 * tx - 1, co0 = cb * UF_BN;
    const float* const xin = (const float*)a.x + (int64_t)n * a.H * a.W * a.Ci;
    const float* const wgt = (const float*)a.w + (int64_t)n * a.w_img_stride;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int pos = tid & 7, grow = tid >> 3;                                     // DMA lane: 16-byte position / row within a 32-row pass
    const int kchunks = a.Ci / 32;

    auto stage = [&](int cc, int buf) {
#pragma unroll
        for (int p = 0; p < (UB_SLAB_PIECES * 8 + 31) / 32; ++p) {               // slab: 10 passes of 32 pixel rows
            const int row = grow + 32 * p;
            if (row >= UB_SLAB_PIECES * 8) break;                                // wave-uniform (a wave's rows are whole groups of 8)
            const int sy = row / UB_SW, sx = row - sy * UB_SW;
            const int iy = cy0 - 1 + sy, ix = cx0 - 1 + sx;
            const bool ok = (sy < 17) & (sx < 17) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
            const float* src = ok ? xin + ((int64_t)iy * a.W + ix) * a.Ci + cc * 32 + (pos ^ ((row >> 1) & 7)) * 4 : (const float*)a.zeros;
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)&slab[buf * UB_SLAB_SLOTS + (row - (grow & 7)) * 8], 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 9; ++p) {                                            // weights: tap p, 32 rows (co) of 128 B
            const int row = grow, co = co0 + row;
            const float* src = wgt + ((int64_t)co * 9 + kTapW[p]) * a.Ci + cc * 32 + (pos ^ ((row >> 1) & 7)) * 4;
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)&wt[buf * UB_WT_SLOTS + (p * UF_BN + wave * 8) * 8], 16, 0, 0);
        }
    };

    uf16 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;
    const int frow = lane & 31, fk = lane >> 5;
    int apix[2];                                                                 // slab pixel of this lane's positions under offset (0, 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) apix[i] = (wave * 4 + 2 * i + (frow >> 4) + 1) * UB_SW + (frow & 15) + 1;

    stage(0, 0);
    __syncthreads();
    for (int cc = 0; cc < kchunks; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < kchunks) stage(cc + 1, buf ^ 1);
        const uf4* const sl = slab + buf * UB_SLAB_SLOTS;
        const uf4* const wl = wt + buf * UB_WT_SLOTS;
#pragma unroll
        for (int m = 0; m < 2; ++m) {                                            // two 16-channel K steps per chunk
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                ubf8 ah[2][2], al[2][2];                                         // [A offset of the phase][i]
#pragma unroll
                for (int ao = 0; ao < kPhaseNA[p]; ++ao)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int q = apix[i] + kPhaseDy[p][ao] * UB_SW + kPhaseDx[p][ao], key = (q >> 1) & 7;
                        ub_split(sl[q * 8 + ((4 * m + 2 * fk) ^ key)], sl[q * 8 + ((4 * m + 2 * fk + 1) ^ key)], ah[ao][i], al[ao][i]);
                    }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int slot = p * 3 + j, ao = kTapA[slot], cls = kTapCls[slot];
                    const int rb = slot * UF_BN + frow;
                    const ubf8 bh = __builtin_bit_cast(ubf8, wl[rb * 8 + ((2 * m + fk) ^ ((frow >> 1) & 7))]);
                    const ubf8 bl = __builtin_bit_cast(ubf8, wl[rb * 8 + ((4 + 2 * m + fk) ^ ((frow >> 1) & 7))]);
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[cls][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term >= 2 ? al[ao][i] : ah[ao][i], (term & 1) ? bl : bh, acc[cls][i], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                                         // drains the DMA of chunk cc + 1 and fences the buffer swap
    }

    // ---- the four class tiles -> one fp32 image [32][32][32 ch] ---------------------------------------------------------------
    float* const ct = (float*)ub_lds;
    float* const nz = ct + UB_CT_FLOATS;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                const int row = 2 * (pp >> 4) + py, col = 2 * (pp & 15) + px;
                ct[(row * 32 + col) * UF_BN + frow] = acc[c][i][r] * a.conv_gain;
            }
    }
    const int OH = 2 * a.H, OW = 2 * a.W;
    const int oy0 = UF_TILE * ty, ox0 = UF_TILE * tx;
    {
        const float ns = a.noise ? a.noise_strength[0] : 0.f;
        for (int e = tid; e < UF_TILE * UF_TILE; e += 256) {
            const int oy = oy0 + e / UF_TILE, ox = ox0 + e % UF_TILE;
            nz[e] = (a.noise && oy < OH && ox < OW) ? a.noise[(int64_t)oy * OW + ox] * ns : 0.f;
        }
    }
    __syncthreads();

    // ---- separable 4-tap FIR down the 28 rows of a column, 4 channels per thread -------------------------------------------------
    const int chunk = tid & 7, xo = tid >> 3;
    if (xo >= UF_TILE) return;
    const int ox = ox0 + xo;
    if (ox >= OW) return;
    uf4 bias = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias = *(const uf4*)(a.bias + co0 + chunk * 4);
    float* const yout = (float*)a.y + (int64_t)n * OH * OW * a.Co + co0 + chunk * 4;
    uf4 out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = uf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < UF_TILE + 3; ++j) {                                      // transposed-conv row 1 + j of the tile
        uf4 h = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) h += *(const uf4*)(ct + (((1 + j) * 32 + xo + 1 + kx) * UF_BN + chunk * 4)) * a.fx[kx];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int m = j - ky;
            if (m >= 0 && m < UF_TILE) out[m & 3] += h * a.fy[ky];
        }
        const int m = j - 3;
        if (m >= 0) {
            const int oy = oy0 + m;
            uf4 v = out[m & 3] + nz[m * UF_TILE + xo] + bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e];
                if (a.act == 1) t = t > 0.f ? t : 0.2f * t;
                t *= a.act_gain;
                if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
                v[e] = t;
            }
            out[m & 3] = uf4{0.f, 0.f, 0.f, 0.f};
            if (oy < OH) *(uf4*)(yout + ((int64_t)oy * OW + ox) * a.Co) = v;
        }
    }
}

} // namespace p3d

extern "C" int p3d_up2_fir_bf16x3(const void* x, const void* w, void* y, const void* zeros128, const float* bias, const float* noise,
                                  const float* noise_strength, const float* fir_yx_host, int32_t n_img, int32_t h, int32_t wdt, int32_t ci, int32_t co,
                                  int64_t w_img_stride, float conv_gain, int32_t act, float act_gain, float clamp, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && w && y && zeros128 && fir_yx_host, "up2_fir_bf16x3: null pointer");
    P3D_REQUIRE(n_img >= 1 && h >= 1 && wdt >= 1, "up2_fir_bf16x3: bad sizes");
    P3D_REQUIRE(act == 0 || act == 1, "up2_fir_bf16x3: act must be 0 (linear) or 1 (lrelu)");
    if (ci % 32 != 0 || co % UF_BN != 0) return fail(P3D_ERR_UNSUPPORTED, "up2_fir_bf16x3: Ci=%d and Co=%d must be multiples of 32", ci, co);
    P3D_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)zeros128 | (uintptr_t)bias) & 15u) == 0, "up2_fir_bf16x3: pointers must be 16-byte aligned");
    P3D_REQUIRE((int64_t)h * wdt * ci * 4 < (1ll << 31), "up2_fir_bf16x3: image beyond 32-bit offsets");
    Up2Args a{};
    a.x = x; a.w = w; a.y = y; a.zeros = zeros128; a.bias = bias; a.noise = noise; a.noise_strength = noise_strength;
    a.N = n_img; a.H = h; a.W = wdt; a.Ci = ci; a.Co = co; a.w_img_stride = w_img_stride; a.conv_gain = conv_gain;
    for (int k = 0; k < 4; ++k) { a.fy[k] = fir_yx_host[k]; a.fx[k] = fir_yx_host[4 + k]; }
    a.act = act; a.act_gain = act_gain; a.clamp = clamp;
    const int tiles_y = (2 * h + UF_TILE - 1) / UF_TILE;
    a.tiles_x = (2 * wdt + UF_TILE - 1) / UF_TILE;
    a.tiles = a.tiles_x * tiles_y;
    const int64_t blocks = (int64_t)((a.tiles + 7) / 8 * 8) * (co / UF_BN);
    P3D_REQUIRE(blocks < (1ll << 31) && n_img < 65536, "up2_fir_bf16x3: bad launch size");
    static std::atomic<uint64_t> once_devs{0};
    const hipError_t e = reserve_lds_once((const void*)up2_fir_bf16x3_kernel, UB_LDS, once_devs);
    if (e != hipSuccess) return fail(P3D_ERR_LAUNCH, "up2_fir_bf16x3: cannot reserve %d B of LDS: %s", UB_LDS, hipGetErrorString(e));
    hipLaunchKernelGGL(up2_fir_bf16x3_kernel, dim3((unsigned)blocks, 1, n_img), dim3(256), UB_LDS, (hipStream_t)stream, a);
    count_launch(FAM_CONV);
    return check_launch("up2_fir_bf16x3");
}
