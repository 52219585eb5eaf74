// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate, per (n, c) image.
//
// Contract: torch_utils/ops/upfirdn2d.cpp:20-102 and upfirdn2d.cu:33-204 of the reference.
// Formulation used here (derived from the op definition, upfirdn2d.py:169-213): with the virtual
// upsampled+padded signal u[X] = x[(X - pad0) / up] when that division is exact and in range,
//     y[o] = gain * sum_k u[o*down + k] * g[k],   g[k] = flip ? f[k] : f[fw-1-k]
// so only taps k == (pad0 - o*down) mod up contribute (polyphase), and every output is an
// independent gather: HBM traffic is sizeof(T) * (in + out) elements, neighbours share taps
// through L1/L2.  Lanes run along whichever axis has unit stride (W for NCHW, C for
// channels_last) so both layouts are read and written coalesced.  Compile-time (UP, DOWN, FW, FH)
// instances cover the StyleGAN2 cases ([1,3,3,1] outer product: 4x4 with up/down in {1,2});
// everything else takes the runtime-parameter instance.
#include "p3d_common.h"

namespace p3d {

struct UpfirArgs {
    const void* x; const float* f; void* y;
    int in_w, in_h, C, N;
    int64_t isx, isy, isc, isn;
    int fw, fh; int64_t fsx, fsy;
    int out_w, out_h;
    int64_t osx, osy, osc, osn;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0, flip;
    float gain;
    int lanes_on_c;            // 1: fastest thread axis walks channels (channels_last)
    int accumulate;            // channels-last kernel only: y += result (the skip-image sum of SynthesisBlock in the upsampling launch)
    int64_t total;             // out_w * out_h * C * N
};

__device__ __forceinline__ int pos_mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }
template <class T> struct alignas(16) Pack16 { T v[16 / sizeof(T)]; };

template <class T, int UPX, int UPY, int DNX, int DNY, int FW, int FH>
__global__ void __launch_bounds__(256) upfirdn2d_kernel(UpfirArgs a)
{
    typedef typename Acc<T>::type S;
    const int upx = UPX ? UPX : a.up_x, upy = UPY ? UPY : a.up_y;
    const int dnx = DNX ? DNX : a.down_x, dny = DNY ? DNY : a.down_y;
    const int fw = FW ? FW : a.fw, fh = FH ? FH : a.fh;
    const T* xp = (const T*)a.x; T* yp = (T*)a.y;

    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.total; idx += (int64_t)gridDim.x * blockDim.x) {
        int ox, oy, c, n; int64_t t = idx;
        if (a.lanes_on_c) { c = (int)(t % a.C); t /= a.C; ox = (int)(t % a.out_w); t /= a.out_w; oy = (int)(t % a.out_h); n = (int)(t / a.out_h); }
        else              { ox = (int)(t % a.out_w); t /= a.out_w; oy = (int)(t % a.out_h); t /= a.out_h; c = (int)(t % a.C); n = (int)(t / a.C); }

        const int bx = ox * dnx - a.pad_x0;          // upsampled-grid coordinate of tap 0
        const int by = oy * dny - a.pad_y0;
        const int kx0 = pos_mod(-bx, upx), ky0 = pos_mod(-by, upy);
        const T* img = xp + c * a.isc + n * a.isn;
        S acc = 0;
        constexpr bool kStatic = (UPX && UPY && FW && FH);
        constexpr int NJX = kStatic ? (FW + (UPX ? UPX : 1) - 1) / (UPX ? UPX : 1) : 0;
        constexpr int NJY = kStatic ? (FH + (UPY ? UPY : 1) - 1) / (UPY ? UPY : 1) : 0;
        auto tap = [&](int kx, int ky) {
            const int ix = (bx + kx) / upx, iy = (by + ky) / upy;      // exact divisions by construction
            if (kx < fw && ky < fh && ix >= 0 && ix < a.in_w && iy >= 0 && iy < a.in_h) {
                const int fx = a.flip ? kx : fw - 1 - kx, fy = a.flip ? ky : fh - 1 - ky;
                acc += ld(img + ix * a.isx + iy * a.isy) * (S)a.f[fx * a.fsx + fy * a.fsy];
            }
        };
        if constexpr (kStatic) {
#pragma unroll
            for (int jy = 0; jy < NJY; ++jy)
#pragma unroll
                for (int jx = 0; jx < NJX; ++jx) tap(kx0 + jx * upx, ky0 + jy * upy);
        } else {
            for (int ky = ky0; ky < fh; ky += upy)
                for (int kx = kx0; kx < fw; kx += upx) tap(kx, ky);
        }
        st(yp + ox * a.osx + oy * a.osy + c * a.osc + n * a.osn, acc * (S)a.gain);
    }
}

// ---- LDS-tiled fast path: contiguous NCHW, square F x F filter (F <= 4), up/down in {1, 2} -----------------
// One 256-thread block produces a 64 x 32 output tile of one (n, c) image: the input footprint of the tile is
// staged once in LDS as fp32 (coalesced rows, zero-filled outside the image), then lane x / wave-row y computes a
// vertical strip of 8 outputs.  For up = 2 only every other tap hits a real sample; which ones depends on the
// output parity, so the four polyphase filters are selected per lane with v_cndmask instead of branching.
constexpr int kTileW = 64, kTileH = 32, kStrip = 8;

template <class T, int UP, int DN, int F>
__global__ void __launch_bounds__(256) upfirdn2d_tiled_kernel(UpfirArgs a)
{
    typedef typename Acc<T>::type S;
    constexpr int IW = ((kTileW - 1) * DN + F - 1) / UP + 2;      // input columns a tile can touch (+1 for the phase)
    constexpr int IH = ((kTileH - 1) * DN + F - 1) / UP + 2;
    constexpr int PITCH = IW | 1;                                  // odd pitch: conflict-free column walks
    __shared__ float tile[IH * PITCH];

    const int tiles_x = (a.out_w + kTileW - 1) / kTileW;
    const int tx0 = (blockIdx.x % tiles_x) * kTileW, ty0 = (blockIdx.x / tiles_x) * kTileH;
    const int nc = blockIdx.y;
    const T* img = (const T*)a.x + (int64_t)nc * a.in_h * a.in_w;
    T* out = (T*)a.y + (int64_t)nc * a.out_h * a.out_w;

    // first input column/row the tile can touch: floor((o0*DN - pad0) / UP)
    const int bx0 = tx0 * DN - a.pad_x0, by0 = ty0 * DN - a.pad_y0;
    const int ix0 = (bx0 >= 0 ? bx0 : bx0 - (UP - 1)) / UP, iy0 = (by0 >= 0 ? by0 : by0 - (UP - 1)) / UP;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    // (every load of the tile first — unconditional, clamped addresses, discarded by a select — then the LDS stores: rolled and under its bounds branch, every tile row was a
    // memory round trip of its own)
    constexpr int NR = (IH + 3) / 4, NCC = (IW + 63) / 64;
    float stg[NR][NCC];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int iy = iy0 + k * 4 + ly;
#pragma unroll
        for (int q = 0; q < NCC; ++q) {
            const int ix = ix0 + q * 64 + lx;
            const bool ok = (iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w);
            const float v = (float)ld(img + (int64_t)min(max(iy, 0), a.in_h - 1) * a.in_w + min(max(ix, 0), a.in_w - 1));
            stg[k][q] = ok ? v : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int q = 0; q < NCC; ++q) {
            const int r = k * 4 + ly, c = q * 64 + lx;
            if ((r < IH) & (c < IW)) tile[r * PITCH + c] = stg[k][q];
        }
    // filter taps in registers, already mirrored for convolution unless flip
    float fr[F][F];
#pragma unroll
    for (int ky = 0; ky < F; ++ky)
#pragma unroll
        for (int kx = 0; kx < F; ++kx) {
            const int fx = a.flip ? kx : F - 1 - kx, fy = a.flip ? ky : F - 1 - ky;
            fr[ky][kx] = a.f[fx * a.fsx + fy * a.fsy] * a.gain;
        }
    __syncthreads();

    const int ox = tx0 + lx;
    if (ox >= a.out_w) return;
    const int bx = ox * DN - a.pad_x0;
    const int kx0 = pos_mod(-bx, UP);                              // first tap that lands on a sample
    const int cx = (bx + kx0) / UP - ix0;                          // its column in the tile
    constexpr int NJ = (F + UP - 1) / UP;
#pragma unroll
    for (int r = 0; r < kStrip; ++r) {
        const int oy = ty0 + ly * kStrip + r;
        if (oy >= a.out_h) break;
        const int by = oy * DN - a.pad_y0;
        const int ky0 = pos_mod(-by, UP);
        const int cy = (by + ky0) / UP - iy0;
        float acc = 0.f;
#pragma unroll
        for (int jy = 0; jy < NJ; ++jy)
#pragma unroll
            for (int jx = 0; jx < NJ; ++jx) {
                float w;
                if (UP == 1) w = fr[jy][jx];
                else {                                             // polyphase select: tap (ky0 + 2jy, kx0 + 2jx)
                    const int y0 = 2 * jy, x0 = 2 * jx;
                    const float w00 = fr[y0][x0], w01 = (x0 + 1 < F) ? fr[y0][x0 + 1] : 0.f;
                    const float w10 = (y0 + 1 < F) ? fr[y0 + 1][x0] : 0.f, w11 = (y0 + 1 < F && x0 + 1 < F) ? fr[y0 + 1][x0 + 1] : 0.f;
                    const float wa = kx0 ? w01 : w00, wb = kx0 ? w11 : w10;
                    w = ky0 ? wb : wa;
                }
                acc = fmaf(tile[(cy + jy) * PITCH + cx + jx], w, acc);
            }
        st(out + (int64_t)oy * a.out_w + ox, (S)acc);
    }
}

template <class T, int UP, int DN, int F>
static void launch_tiled(const UpfirArgs& a, hipStream_t s)
{
    const int tiles = ((a.out_w + kTileW - 1) / kTileW) * ((a.out_h + kTileH - 1) / kTileH);
    hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UP, DN, F>), dim3(tiles, a.C * a.N), dim3(256), 0, s, a);
}

// ---- channels-last fast path: lanes walk the channel axis in 16-byte vectors -------------------------------------
// x is [N][H][W][C] in memory (torch channels_last).  One lane owns VEC = 16/sizeof(T) consecutive channels of one
// output pixel; every tap is a single aligned 16-byte load shared by no one else, and consecutive lanes read
// consecutive channels, so each wave instruction covers whole 128-byte lines.  No LDS: the tap overlap between
// neighbouring pixels is served by L1/L2.
template <class T, int UP, int DN, int F>
__global__ void __launch_bounds__(256) upfirdn2d_cl_kernel(UpfirArgs a)
{
    constexpr int VEC = 16 / sizeof(T);
    typedef Pack16<T> P;
    const int cvec = a.C / VEC;
    const int64_t total = (int64_t)a.N * a.out_h * a.out_w * cvec;
    float fr[F][F];
#pragma unroll
    for (int ky = 0; ky < F; ++ky)
#pragma unroll
        for (int kx = 0; kx < F; ++kx) {
            const int fx = a.flip ? kx : F - 1 - kx, fy = a.flip ? ky : F - 1 - ky;
            fr[ky][kx] = a.f[fx * a.fsx + fy * a.fsy] * a.gain;
        }
    constexpr int NJ = (F + UP - 1) / UP;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = idx;
        const int cv = (int)(t % cvec); t /= cvec;
        const int ox = (int)(t % a.out_w); t /= a.out_w;
        const int oy = (int)(t % a.out_h);
        const int n = (int)(t / a.out_h);
        const int bx = ox * DN - a.pad_x0, by = oy * DN - a.pad_y0;
        const int kx0 = pos_mod(-bx, UP), ky0 = pos_mod(-by, UP);
        const int ixb = (bx + kx0) / UP, iyb = (by + ky0) / UP;
        const T* img = (const T*)a.x + (int64_t)n * a.isn + cv * VEC;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        // every tap of the output requested at once, unconditionally (an out-of-image tap reads a clamped address and is skipped below): as `if (inside) { load; fma }`
        // each tap was a branch with its own wait — sixteen dependent memory round trips per output at up = 1 (0.18 of the HBM peak, waves 69 % at s_waitcnt)
        P tap[NJ][NJ];
#pragma unroll
        for (int jy = 0; jy < NJ; ++jy)
#pragma unroll
            for (int jx = 0; jx < NJ; ++jx) {
                const int iy = min(max(iyb + jy, 0), a.in_h - 1), ix = min(max(ixb + jx, 0), a.in_w - 1);
                tap[jy][jx] = *(const P*)(img + (int64_t)iy * a.isy + (int64_t)ix * a.isx);
            }
        P* const dst = (P*)((T*)a.y + (int64_t)n * a.osn + (int64_t)oy * a.osy + (int64_t)ox * a.osx + cv * VEC);
        P old = tap[0][0];
        if (a.accumulate) old = *dst;                                // (with the taps, not behind them)
#pragma unroll
        for (int jy = 0; jy < NJ; ++jy) {
            const int iy = iyb + jy;
#pragma unroll
            for (int jx = 0; jx < NJ; ++jx) {
                const int ix = ixb + jx;
                float w;
                if (UP == 1) w = fr[jy][jx];
                else {
                    const int y0 = 2 * jy, x0 = 2 * jx;
                    const float w00 = fr[y0][x0], w01 = (x0 + 1 < F) ? fr[y0][x0 + 1] : 0.f;
                    const float w10 = (y0 + 1 < F) ? fr[y0 + 1][x0] : 0.f, w11 = (y0 + 1 < F && x0 + 1 < F) ? fr[y0 + 1][x0 + 1] : 0.f;
                    const float wa = kx0 ? w01 : w00, wb = kx0 ? w11 : w10;
                    w = ky0 ? wb : wa;
                }
                if ((iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w)) {
                    const P v = tap[jy][jx];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf((float)ld(&v.v[e]), w, acc[e]);
                }
            }
        }
        P o;
        if (a.accumulate) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += (float)ld(&old.v[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) st(&o.v[e], (typename Acc<T>::type)acc[e]);
        *dst = o;
    }
}

// ---- channels-last 4x4 FIR with the layer epilogue fused (LDS-tiled) ---------------------------------------------------
// The x2 synthesis layers end with: transposed conv -> this FIR (up = down = 1, 4x4, gain 4) -> + noise -> bias_act
// (conv2d_resample.py:128, networks_stylegan2.py:319-332).  One block = 16 x 16 output pixels x 128 bytes of channels
// (64 halfs / 32 floats): the 19 x 19 input footprint is staged once in LDS, each thread slides a 4-column window
// down 8 rows for one 16-byte channel chunk (44 LDS reads for 8 outputs instead of 128), then applies
// v = fir + noise*strength + bias -> lrelu -> * act_gain -> clamp and stores — the activation never makes a second trip
// through HBM.
struct FirEpilogue { const float* bias; const float* noise; const float* noise_strength; int act; float alpha, act_gain, clamp;
                     int y_split; };   // fp32 only: the result leaves as bf16x3 K rows — per 32 channels [32 x bf16 hi | 32 x bf16 lo] — for the next bf16x3 convolution (csrc/conv2d.hip, XS)

template <class T>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 3 : 2) fir4_cl_fused_kernel(UpfirArgs a, FirEpilogue ep)
{
    constexpr int VEC = 16 / sizeof(T);            // elements per 16-byte chunk
    constexpr int CB = 8 * VEC;                    // channels per block (128 bytes)
    constexpr int TI = 19;                         // input tile side for a 16 x 16 output tile and 4 taps
    typedef Pack16<T> P;
    __shared__ P tile[TI * TI * 8];
    const int tiles_x = (a.out_w + 15) / 16, tiles_y = (a.out_h + 15) / 16;
    int b = blockIdx.x;
    const int cblk = b % (a.C / CB); b /= (a.C / CB);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int ox0 = tx * 16, oy0 = ty * 16;
    const int ix0 = ox0 - a.pad_x0, iy0 = oy0 - a.pad_y0;
    const T* img = (const T*)a.x + (int64_t)n * a.isn + cblk * CB;
    const int chunk = threadIdx.x & 7;
    // Everything the block reads besides its tile — the sixteen filter taps, the thread's biases, the noise strength and its eight noise values — is requested FIRST and
    // unconditionally (an absent operand reads the filter's first tap and is discarded), so that it shares the tile's round trip.  Read where they were used — the noise under
    // its bounds branch, the taps behind the staging stores, the biases behind the rendezvous — they were three more dependent round trips of a block that lives for little
    // more than one (waves of this kernel spent 53 % of their life at s_waitcnt: profiles/round6_zk_kernel_pmc_infer.txt).
    float fr[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const int fx = a.flip ? kx : 3 - kx, fy = a.flip ? ky : 3 - ky;
            fr[ky][kx] = a.f[fx * a.fsx + fy * a.fsy] * a.gain;
        }
    const int c0 = cblk * CB + chunk * VEC;
    float bias[VEC];
    {
        const float* const bp = ep.bias ? ep.bias + c0 : a.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { const float v = bp[ep.bias ? k : 0]; bias[k] = ep.bias ? v : 0.f; }
    }
    const int x = (threadIdx.x >> 3) & 15, yh = threadIdx.x >> 7;
    float nzv[8];
    {
        const float ns = (ep.noise ? ep.noise_strength : a.f)[0];
        const float* const np = ep.noise ? ep.noise : a.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int oy = oy0 + yh * 8 + o, ox = ox0 + x;
            const bool ok = ep.noise && oy < a.out_h && ox < a.out_w;
            const float v = np[ok ? (int64_t)oy * a.out_w + ox : 0];
            nzv[o] = ok ? v * ns : 0.f;
        }
    }
    // all twelve footprint loads of a thread are issued before the first one is consumed: the block pays ONE HBM round
    // trip for its tile, not one per load (the rolled loop did, and ran at a quarter of the memory rate)
    constexpr int NLD = (TI * TI + 31) / 32;       // 12
    P stage[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = (threadIdx.x >> 3) + 32 * k;
        const int r = e / TI, c = e - r * TI;
        const int iy = iy0 + r, ix = ix0 + c;
#pragma unroll
        for (int q = 0; q < VEC; ++q) st(&stage[k].v[q], (typename Acc<T>::type)0);
        if ((e < TI * TI) & (iy >= 0) & (iy < a.in_h) & (ix >= 0) & (ix < a.in_w))
            stage[k] = *(const P*)(img + (int64_t)iy * a.isy + (int64_t)ix * a.isx + chunk * VEC);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = (threadIdx.x >> 3) + 32 * k;
        if (e < TI * TI) tile[e * 8 + chunk] = stage[k];
    }
    __syncthreads();
    const int ox = ox0 + x;
    if (ox >= a.out_w) return;
    // One output row at a time: its sixteen taps are read from the LDS tile (16-byte reads), summed in the order (ky, kx) ascending, finished and stored before the next
    // row starts.  (The earlier form walked the eleven INPUT rows once — 44 reads instead of 128 — and the compiler kept all 44 in registers before the first fma:
    // 235 VGPRs in fp32, 256 in fp16, one to two waves per SIMD for a memory pass.  LDS reads are not what this kernel waits for.)
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const int oy = oy0 + yh * 8 + o;
        if (oy >= a.out_h) break;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const P v = tile[((yh * 8 + o + ky) * TI + x + kx) * 8 + chunk];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = fmaf((float)ld(&v.v[k]), fr[ky][kx], acc[k]);
            }
        const float nz = nzv[o];
        P outv;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float v = acc[k] + nz + bias[k];
            if (ep.act == 3) v = v > 0.f ? v : v * ep.alpha;
            v *= ep.act_gain;
            if (ep.clamp >= 0.f) v = fminf(fmaxf(v, -ep.clamp), ep.clamp);
            st(&outv.v[k], (typename Acc<T>::type)v);
        }
        if constexpr (sizeof(T) == 4) {
            if (ep.y_split) {                      // this thread's four channels: hi at bf16 slots [4 chunk ..], lo at [32 + 4 chunk ..] of the 32-channel row
                typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                bf4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = (float)ld(&outv.v[k]);
                    const __bf16 h = (__bf16)v;
                    hi[k] = h; lo[k] = (__bf16)(v - (float)h);
                }
                __bf16* row = (__bf16*)((T*)a.y + (int64_t)n * a.osn + (int64_t)oy * a.osy + (int64_t)ox * a.osx + cblk * CB);
                *(bf4*)(row + chunk * 4) = hi;              // (16-byte stores through a lane-pair exchange measured no faster: profiles/round6_d_*)
                *(bf4*)(row + 32 + chunk * 4) = lo;
                continue;
            }
        }
        *(P*)((T*)a.y + (int64_t)n * a.osn + (int64_t)oy * a.osy + (int64_t)ox * a.osx + c0) = outv;
    }
}

template <class T>
static bool try_channels_last(const UpfirArgs& a, hipStream_t s)
{
    constexpr int VEC = 16 / sizeof(T);
    const bool cl = a.isc == 1 && a.osc == 1 && a.isx == a.C && a.osx == a.C && a.isy == (int64_t)a.in_w * a.C && a.osy == (int64_t)a.out_w * a.C;
    if (!cl || sizeof(T) == 8 || a.C % VEC != 0 || a.fw != a.fh || a.up_x != a.up_y || a.down_x != a.down_y) return false;
    if ((((uintptr_t)a.x) | ((uintptr_t)a.y)) & 15u) return false;
    const int64_t total = (int64_t)a.N * a.out_h * a.out_w * (a.C / VEC);
    int64_t blocks64 = (total + 255) / 256;
    const int blocks = (int)(blocks64 > (int64_t)kNumCU * 32 ? (int64_t)kNumCU * 32 : blocks64);
    const int u = a.up_x, d = a.down_x, f = a.fw;
    if constexpr (sizeof(T) <= 4) {
        // plain 4x4 filtering at full rate (the blur in front of every stride-2 convolution of D and of the label-map Encoder, and its gradient): the
        // LDS-tiled kernel of the fused FIR + epilogue with an empty epilogue — one HBM round trip per 19 x 19 x 128-byte tile instead of sixteen
        // cache-served taps per output (upfirdn2d_cl_kernel<., 1, 1, 4>: 0.18 of the HBM peak, waves 69 % at s_waitcnt in a training iteration)
        constexpr int CB = 8 * VEC;
        static const bool no_fir4 = getenv("P3D_UPFIRDN_NO_FIR4") != nullptr;               // (A/B switch)
        if (!no_fir4 && u == 1 && d == 1 && f == 4 && !a.accumulate && a.C % CB == 0 && a.out_h >= 1 && a.out_w >= 1) {
            FirEpilogue ep{nullptr, nullptr, nullptr, 1, 0.2f, 1.0f, -1.0f, 0};
            const int64_t fb = (int64_t)a.N * ((a.out_h + 15) / 16) * ((a.out_w + 15) / 16) * (a.C / CB);
            if (fb > 0 && fb < (1ll << 31)) {
                hipLaunchKernelGGL(fir4_cl_fused_kernel<T>, dim3((unsigned)fb), dim3(256), 0, s, a, ep);
                return true;
            }
        }
    }
#define P3D_CL(U, D, FF) if (u == U && d == D && f == FF) { hipLaunchKernelGGL((upfirdn2d_cl_kernel<T, U, D, FF>), dim3(blocks), dim3(256), 0, s, a); return true; }
    P3D_CL(1, 1, 4) P3D_CL(2, 1, 4) P3D_CL(1, 2, 4) P3D_CL(2, 2, 4)
#undef P3D_CL
    return false;
}

template <class T>
static bool try_tiled(const UpfirArgs& a, hipStream_t s)
{
    const bool nchw = a.isx == 1 && a.isy == a.in_w && a.isc == (int64_t)a.in_w * a.in_h && a.isn == a.isc * a.C &&
                      a.osx == 1 && a.osy == a.out_w && a.osc == (int64_t)a.out_w * a.out_h && a.osn == a.osc * a.C;
    if (!nchw || a.fw != a.fh || a.up_x != a.up_y || a.down_x != a.down_y || (int64_t)a.C * a.N > 65535) return false;
    if (sizeof(T) == 8) return false;                              // fp64 keeps the exact generic kernel
    const int u = a.up_x, d = a.down_x, f = a.fw;
#define P3D_TILED(U, D, FF) if (u == U && d == D && f == FF) { launch_tiled<T, U, D, FF>(a, s); return true; }
    P3D_TILED(1, 1, 4) P3D_TILED(2, 1, 4) P3D_TILED(1, 2, 4) P3D_TILED(2, 2, 4)
    P3D_TILED(1, 1, 3) P3D_TILED(2, 1, 3) P3D_TILED(1, 2, 3) P3D_TILED(1, 1, 2) P3D_TILED(2, 1, 2) P3D_TILED(1, 2, 2)
#undef P3D_TILED
    return false;
}

template <class T>
static int launch_upfirdn2d(const UpfirArgs& a, hipStream_t s)
{
    const int threads = 256;
    int64_t blocks64 = (a.total + threads - 1) / threads;
    const int64_t cap = (int64_t)kNumCU * 32;
    int blocks = (int)(blocks64 > cap ? cap : blocks64);
    if (blocks < 1) blocks = 1;
    if (a.accumulate) {
        if (try_channels_last<T>(a, s)) { count_launch(FAM_UPFIRDN); return check_launch("upfirdn2d(channels_last, accumulate)"); }
        return fail(P3D_ERR_UNSUPPORTED, "upfirdn2d_acc: only the channels-last 4-tap kernels accumulate");
    }
    if (try_tiled<T>(a, s)) { count_launch(FAM_UPFIRDN); return check_launch("upfirdn2d(tiled)"); }
    if (try_channels_last<T>(a, s)) { count_launch(FAM_UPFIRDN); return check_launch("upfirdn2d(channels_last)"); }
#define P3D_UPFIR_CASE(ux, uy, dx, dy, w, h) \
    if (a.up_x == ux && a.up_y == uy && a.down_x == dx && a.down_y == dy && a.fw == w && a.fh == h) { \
        hipLaunchKernelGGL((upfirdn2d_kernel<T, ux, uy, dx, dy, w, h>), dim3(blocks), dim3(threads), 0, s, a); } else
    P3D_UPFIR_CASE(1, 1, 1, 1, 4, 4)
    P3D_UPFIR_CASE(2, 2, 1, 1, 4, 4)
    P3D_UPFIR_CASE(1, 1, 2, 2, 4, 4)
    P3D_UPFIR_CASE(2, 1, 1, 1, 4, 1)
    P3D_UPFIR_CASE(1, 2, 1, 1, 1, 4)
    P3D_UPFIR_CASE(1, 1, 2, 1, 4, 1)
    P3D_UPFIR_CASE(1, 1, 1, 2, 1, 4)
    { hipLaunchKernelGGL((upfirdn2d_kernel<T, 0, 0, 0, 0, 0, 0>), dim3(blocks), dim3(threads), 0, s, a); }
#undef P3D_UPFIR_CASE
    count_launch(FAM_UPFIRDN);
    return check_launch("upfirdn2d");
}

} // namespace p3d

static int upfirdn2d_run(const void* x, const float* f, void* y, int dtype,
                         const int32_t in_size[4], const int64_t in_stride[4],
                         const int32_t f_size[2], const int64_t f_stride[2],
                         const int32_t out_size[4], const int64_t out_stride[4],
                         int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                         int32_t pad_x0, int32_t pad_y0, int32_t flip, float gain, int accumulate, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && f && y, "upfirdn2d: x, f and y must be non-null");
    P3D_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "upfirdn2d: up/down factors must be >= 1");
    P3D_REQUIRE(f_size[0] >= 1 && f_size[1] >= 1, "upfirdn2d: filter must have at least one tap");
    P3D_REQUIRE(out_size[2] == in_size[2] && out_size[3] == in_size[3], "upfirdn2d: channel/batch mismatch between x and y");
    UpfirArgs a;
    a.x = x; a.f = f; a.y = y;
    a.in_w = in_size[0]; a.in_h = in_size[1]; a.C = in_size[2]; a.N = in_size[3];
    a.isx = in_stride[0]; a.isy = in_stride[1]; a.isc = in_stride[2]; a.isn = in_stride[3];
    a.fw = f_size[0]; a.fh = f_size[1]; a.fsx = f_stride[0]; a.fsy = f_stride[1];
    a.out_w = out_size[0]; a.out_h = out_size[1];
    a.osx = out_stride[0]; a.osy = out_stride[1]; a.osc = out_stride[2]; a.osn = out_stride[3];
    a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y;
    a.pad_x0 = pad_x0; a.pad_y0 = pad_y0; a.flip = flip ? 1 : 0; a.gain = gain;
    a.lanes_on_c = (a.osc == 1 && a.C > 1 && a.osx != 1) ? 1 : 0;
    a.accumulate = accumulate;
    a.total = (int64_t)a.out_w * a.out_h * a.C * a.N;
    if (a.total <= 0) return P3D_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case P3D_F32: return launch_upfirdn2d<float>(a, s);
        case P3D_F16: return launch_upfirdn2d<__half>(a, s);
        case P3D_F64: return launch_upfirdn2d<double>(a, s);
    }
    return fail(P3D_ERR_ARGUMENT, "upfirdn2d: unknown dtype %d", dtype);
}

extern "C" int p3d_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                             const int32_t in_size[4], const int64_t in_stride[4],
                             const int32_t f_size[2], const int64_t f_stride[2],
                             const int32_t out_size[4], const int64_t out_stride[4],
                             int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                             int32_t pad_x0, int32_t pad_y0, int32_t flip, float gain, p3d_stream_t stream)
{
    return upfirdn2d_run(x, f, y, dtype, in_size, in_stride, f_size, f_stride, out_size, out_stride, up_x, up_y, down_x, down_y, pad_x0, pad_y0, flip, gain, 0, stream);
}

extern "C" int p3d_upfirdn2d_acc(const void* x, const float* f, void* y, int dtype,
                                 const int32_t in_size[4], const int64_t in_stride[4],
                                 const int32_t f_size[2], const int64_t f_stride[2],
                                 const int32_t out_size[4], const int64_t out_stride[4],
                                 int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                                 int32_t pad_x0, int32_t pad_y0, int32_t flip, float gain, p3d_stream_t stream)
{
    return upfirdn2d_run(x, f, y, dtype, in_size, in_stride, f_size, f_stride, out_size, out_stride, up_x, up_y, down_x, down_y, pad_x0, pad_y0, flip, gain, 1, stream);
}

static int fir4_bias_act_run(const void* x, const float* f, void* y, int dtype, int32_t n_img, int32_t c, int32_t in_h, int32_t in_w,
                             int32_t pad_x0, int32_t pad_y0, int32_t out_h, int32_t out_w, int32_t flip, float gain,
                             const float* bias, const float* noise, const float* noise_strength, int32_t act, float alpha, float act_gain,
                             float clamp, int32_t y_split, p3d_stream_t stream);

extern "C" int p3d_fir4_bias_act_nhwc(const void* x, const float* f, void* y, int dtype, int32_t n_img, int32_t c, int32_t in_h, int32_t in_w,
                                      int32_t pad_x0, int32_t pad_y0, int32_t out_h, int32_t out_w, int32_t flip, float gain,
                                      const float* bias, const float* noise, const float* noise_strength, int32_t act, float alpha, float act_gain,
                                      float clamp, p3d_stream_t stream)
{
    return fir4_bias_act_run(x, f, y, dtype, n_img, c, in_h, in_w, pad_x0, pad_y0, out_h, out_w, flip, gain, bias, noise, noise_strength, act, alpha, act_gain, clamp, 0, stream);
}

extern "C" int p3d_fir4_bias_act_nhwc_split(const void* x, const float* f, void* y, int32_t n_img, int32_t c, int32_t in_h, int32_t in_w,
                                            int32_t pad_x0, int32_t pad_y0, int32_t out_h, int32_t out_w, int32_t flip, float gain,
                                            const float* bias, const float* noise, const float* noise_strength, int32_t act, float alpha, float act_gain,
                                            float clamp, p3d_stream_t stream)
{
    return fir4_bias_act_run(x, f, y, P3D_F32, n_img, c, in_h, in_w, pad_x0, pad_y0, out_h, out_w, flip, gain, bias, noise, noise_strength, act, alpha, act_gain, clamp, 1, stream);
}

static int fir4_bias_act_run(const void* x, const float* f, void* y, int dtype, int32_t n_img, int32_t c, int32_t in_h, int32_t in_w,
                             int32_t pad_x0, int32_t pad_y0, int32_t out_h, int32_t out_w, int32_t flip, float gain,
                             const float* bias, const float* noise, const float* noise_strength, int32_t act, float alpha, float act_gain,
                             float clamp, int32_t y_split, p3d_stream_t stream)
{
    using namespace p3d;
    P3D_REQUIRE(x && f && y, "fir4_bias_act_nhwc: null pointer");
    P3D_REQUIRE(dtype == P3D_F16 || dtype == P3D_F32, "fir4_bias_act_nhwc: dtype must be fp16 or fp32");
    P3D_REQUIRE(act == 1 || act == 3, "fir4_bias_act_nhwc: act must be linear (1) or lrelu (3)");
    const int cb = dtype == P3D_F16 ? 64 : 32;
    if (c % cb != 0) return fail(P3D_ERR_UNSUPPORTED, "fir4_bias_act_nhwc: C=%d must be a multiple of %d", c, cb);
    P3D_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0, "fir4_bias_act_nhwc: x and y must be 16-byte aligned");
    UpfirArgs a{};
    a.x = x; a.f = f; a.y = y; a.in_w = in_w; a.in_h = in_h; a.C = c; a.N = n_img;
    a.isc = 1; a.isx = c; a.isy = (int64_t)in_w * c; a.isn = (int64_t)in_h * in_w * c;
    a.osc = 1; a.osx = c; a.osy = (int64_t)out_w * c; a.osn = (int64_t)out_h * out_w * c;
    a.fw = a.fh = 4; a.fsx = 1; a.fsy = 4; a.out_w = out_w; a.out_h = out_h;
    a.up_x = a.up_y = a.down_x = a.down_y = 1; a.pad_x0 = pad_x0; a.pad_y0 = pad_y0; a.flip = flip ? 1 : 0; a.gain = gain;
    FirEpilogue ep{bias, noise, noise_strength, act, alpha, act_gain, clamp, y_split};
    const int64_t blocks = (int64_t)n_img * ((out_h + 15) / 16) * ((out_w + 15) / 16) * (c / cb);
    P3D_REQUIRE(blocks > 0 && blocks < (1ll << 31), "fir4_bias_act_nhwc: bad launch size");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P3D_F16) hipLaunchKernelGGL(fir4_cl_fused_kernel<__half>, dim3((unsigned)blocks), dim3(256), 0, s, a, ep);
    else                  hipLaunchKernelGGL(fir4_cl_fused_kernel<float>,  dim3((unsigned)blocks), dim3(256), 0, s, a, ep);
    count_launch(FAM_UPFIRDN);
    return check_launch("fir4_bias_act_nhwc");
}
