// Backward of the fused tri-plane ray-marcher for gfx950 (training configs).
//
// The reference differentiates ImportanceRenderer.forward (renderer.py:88-140) by keeping ~120 ops' worth of per-sample tensors
// (17 GB at 4 images x 128^2 rays x 96 samples); here nothing per-sample is kept by the forward pass and the backward is two
// launches that recompute:
//   1. render_forward_kernel<.., TAPE = true> (render_device.h): the forward sweep again, driven by dL/dfeat.  Compositing is
//      linear in the colours, so dL/dw of an interval is a dot product with the colours the sweep produces anyway; a per-ray
//      back-to-front walk over the recorded (alpha, T, dL/dw, sigma_mid) then gives every SAMPLE its two scalars: the colour
//      weight (w[k-1] + w[k]) / 2 and dL/dsigma_k.  Importance depths are constants (renderer.py:198, 211: no_grad + detach).
//   2. render_backward_kernel: point-wise and order-free.  Per sample: gather + both MLPs again, then their backward on the matrix
//      cores, all in the forward's "lane = ray" layout:
//        dh = W2^T do, df = W1^T da          weights as the MFMA A operand (pre-permuted streams, p3d_pack_decoder_bwd),
//                                            the lane's own registers as B — no cross-lane traffic;
//        dW2 += do h^T, dW1 += da f^T        K = the wave's 32 rays: both operands go through a 32 x 32 LDS transpose
//                                            ([channel][ray], pitch 36) and come back as 16 consecutive rays per lane;
//                                            bias and density-row sums are row sums of the same transposed tiles;
//        d planes                            12 taps x 16 channels of atomic adds per lane (the reference's grid_sample
//                                            backward does the same scatter).  Measured: this scatter IS the kernel — 1.2 G
//                                            float atomics at ~20 G/s (32 of them serialise on every 128-byte texel line),
//                                            61 of the 64 ms at 4 x 128^2 rays x 96 samples; everything else takes 3 ms.
//      The 128 accumulator registers of the four weight-gradient tiles per net pin the kernel at one wave per SIMD (512-register
//      budget); weight gradients leave through atomics once per wave.
#include "render_device.h"

namespace p3d {
// Ablation builds (measurement only, never the shipped library: tests/gpu_probe_rbwd.sh compiles this file with -DP3D_RBWD_DEBUG=<bits>):
// 1 no plane atomics (the compiler then drops the whole scatter walk), 2 no weight-gradient products, 4 no gather, 8 no data-gradient MFMAs,
// 16 the scatter walk with its atomics replaced by a register sink (walk kept, no memory-side work).  Compile-time so that the product build is untouched.
#ifndef P3D_RBWD_DEBUG
#define P3D_RBWD_DEBUG 0
#endif
constexpr int kRbwdDbg = P3D_RBWD_DEBUG;

constexpr int kBwdNetStride = 4096;              // per net: 64 MFMA steps x 64 lanes (steps 0..31: dh, 32..63: df)
constexpr int kBwdFloats = 2 * kBwdNetStride;
constexpr int kBwdWaves = 4;
constexpr int TP = 36;                           // pitch (floats) of the transposed [row][ray] tiles: conflict-free b32 writes / b128 reads
constexpr int kBwdWaveLds = (32 + 64 + 32) * TP + 32 + kTapTile;     // T_f, T_x (h, then da), T_do, T_dsigma, tap table
// gradient record of one net (floats), effective (gain-scaled) weights: W1 [64][32], b1 [64], W2 [33][64], b2 [33]
constexpr int D_W1 = 0, D_B1 = 2048, D_W2 = 2112, D_B2 = 4224, kGradNetStride = 4260;

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }   // row of accumulator register r

struct PackBwdArgs { const float* w1[2]; const float* w2[2]; int n_nets; float wg1, wg2; };

__global__ void __launch_bounds__(256) pack_decoder_bwd_kernel(PackBwdArgs p, float* out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kBwdFloats; i += gridDim.x * blockDim.x) {
        const int n = i / kBwdNetStride, e = i % kBwdNetStride;
        const int q4 = e / 256, lane = (e % 256) / 4, q = q4 * 4 + (e & 3);
        const int row = lane & 31, h = lane >> 5;
        float v = 0.f;
        if (n < p.n_nets) {
            if (q < 32) {                                   // dh: hidden 32t + row  <-  colour channel acc_row(s, h)
                const int t = q >> 4, s = q & 15;
                v = p.w2[n][(1 + acc_row(s, h)) * 64 + 32 * t + row] * p.wg2;
            } else {                                        // df: feature row  <-  hidden 32t + acc_row(r, h)
                const int s = q - 32, t = s >> 4, r = s & 15;
                v = p.w1[n][(32 * t + acc_row(r, h)) * 32 + row] * p.wg1;
            }
        }
        out[i] = v;
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16 consecutive floats of a transposed tile row -> registers (4 x ds_read_b128)
__device__ __forceinline__ void read_row16(const float* p, float (&v)[16])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *(const f32x4*)(p + 4 * q);
        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
}

// d planes: the transpose of gather_features, with COALESCED atomics.  Scattering straight from the accumulator layout (lane = ray)
// makes every atomic instruction touch 32 texel lines with 2 dwords each, and the memory side serialises them: 1.2 G atomics ran at
// 20 G/s (61 of the kernel's then 64 ms).  Instead the wave parks dL/dfeature as [ray][feature] in LDS together with each ray's 12 tap
// offsets and weights, and then walks the rays two at a time with lane = CHANNEL: an atomic instruction is two full 128-byte texel
// lines, one request each.
__device__ __forceinline__ void scatter_features(const RenderArgs& a, float* __restrict__ d_planes, unsigned img_off, int lane, bool live,
                                                 float px, float py, float pz, const f32x16& df, float* Tdf, unsigned* Toff, float* Tw)
{
    const int W = a.W, H = a.H;
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                  // accumulator rows 8q + 4h + {0..3}: one 16-byte store
        f32x4 v; v[0] = df[4 * q]; v[1] = df[4 * q + 1]; v[2] = df[4 * q + 2]; v[3] = df[4 * q + 3];
        *(f32x4*)(Tdf + j * TP + 8 * q + 4 * h) = v;
    }
    if (h == 0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float gx = (p == 2) ? pz : px;
            const float gy = (p == 0) ? py : (p == 1 ? pz : px);
            float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
            float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
            ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);
            iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
            const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int x = x0 + (t & 1), y = y0 + (t >> 1);
                const bool ok = live & (x >= 0) & (x < W) & (y >= 0) & (y < H);     // zero padding: such a tap never contributed
                const float w = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0) * (1.f / 3.f);
                Tw[j * 12 + p * 4 + t] = ok ? w : 0.f;
                Toff[j * 12 + p * 4 + t] = ok ? img_off + (unsigned)((p * H + y) * W + x) * 32u : 0u;
            }
        }
    }
    wave_sync();
    // Entry-major walk with lane = CHANNEL: half-wave `half` takes entries 16 half .. 16 half + 15 in order — per entry one read of the lane's channel
    // value and six 16-byte reads of its 12 weights / offsets, all independent (a tap-major walk with three dependent LDS reads per atomic spent
    // more time in LDS latency than in the atomics: at one wave per SIMD nothing hides it).
    // RUN COMBINING (round 3).  The atomics were half of this kernel (profiles/round3_i_render_bwd_ablation.log: 10.6 ms, 5.4 without them): 2 x 1 536 B
    // of read-modify-write per sample at the memory side.  In ray mode a wave's 32 entries are 32 CONSECUTIVE SAMPLES OF ONE RAY, and consecutive
    // samples keep hitting the same texels — always on the plane that faces the camera (a ray barely moves across it), and wherever the importance
    // samples bunch up on the other two.  So each tap slot keeps a pending (texel, value): a sample whose tap is the pending texel adds to it in a
    // register, anything else sends the pending sum as ONE atomic and starts a new run; 12 flushes close the walk.
    // (one plane at a time: 4 pending slots + 4 weights + 4 offsets live instead of 12 + 12 + 12 — the kernel has no register to spare)
    const int c = lane & 31, half = lane >> 5;
    [[maybe_unused]] float sink = 0.f;
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
        unsigned p_off[4]; float p_val[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { p_off[t] = 0xffffffffu; p_val[t] = 0.f; }
        // entries one ahead in registers (the walk is a chain of LDS reads at one wave per SIMD) and select-based bookkeeping: the only
        // control flow per tap is the exec-masked atomic of a flush (the first cut branched three levels deep per tap: 2.4 ms of walk
        // beside 2.6 ms of atomics, profiles/round3_k_*)
        const float* const tv = Tdf + half * 16 * TP + c;
        const float* const tw = Tw + half * 16 * 12 + 4 * p;
        const unsigned* const to = Toff + half * 16 * 12 + 4 * p;
        float v_n = tv[0];
        f32x4 w_n = *(const f32x4*)tw;
        u32x4 o_n = *(const u32x4*)to;
#pragma unroll
        for (int pr = 0; pr < 16; ++pr) {
            const float v = v_n;
            const f32x4 w = w_n;
            const u32x4 off = o_n;
            if (pr + 1 < 16) {
                v_n = tv[(pr + 1) * TP];
                w_n = *(const f32x4*)(tw + (pr + 1) * 12);
                o_n = *(const u32x4*)(to + (pr + 1) * 12);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float contrib = w[t] * v;
                const bool valid = w[t] != 0.f;                   // dead lane / out-of-image tap: never contributed (uniform over the half-wave)
                const bool same = off[t] == p_off[t];
                if (valid && !same && p_off[t] != 0xffffffffu && !(kRbwdDbg & 1)) {
                    if constexpr ((kRbwdDbg & 16) != 0) sink += p_val[t] * (float)p_off[t];
                    else unsafeAtomicAdd(d_planes + p_off[t] + c, p_val[t]);
                }
                p_val[t] = valid ? (same ? p_val[t] + contrib : contrib) : p_val[t];
                p_off[t] = valid ? off[t] : p_off[t];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (p_off[t] != 0xffffffffu && !(kRbwdDbg & 1)) {
                if constexpr ((kRbwdDbg & 16) != 0) sink += p_val[t] * (float)p_off[t];
                else unsafeAtomicAdd(d_planes + p_off[t] + c, p_val[t]);
            }
    }
    if constexpr ((kRbwdDbg & 16) != 0) { if (sink == 12345.678f) d_planes[0] = sink; }
}

// POINTS = true: the same point-wise pass for free-standing point queries (renderer.py:142-148 run_model, G.sample_mixed — the density
// regularisation of loss.py:681-706): a wave walks tiles of 32 POINTS instead of the samples of 32 rays, the upstream gradients are
// dL/drgb [P][NNETS*32] (post-squash outputs, natural channel order) and dL/dsigma [P] instead of the tape's (colour weight, dL/dsigma).
struct PointArgs { const float* coords; const float* g_rgb; const float* g_sigma; int pts_per_img, total_pts;
                   int tiles_per_wave; };      // ray mode: consecutive (ray, 32-sample chunk) tiles one wave takes

template <int NNETS, bool POINTS = false>
__global__ void __launch_bounds__(kBwdWaves * 64, 1)
render_backward_kernel(RenderArgs a, const float* __restrict__ bwd_stream, float* __restrict__ d_planes, float* __restrict__ d_dec, PointArgs pa)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;                       // forward layout: ray j, half h
    const int ti = lane & 31, tk = lane >> 5;                     // transposed layout: tile row ti, ray half tk
    float* const bwd = lds + kDecoderFloats;
    {
        const f32x4* s0 = (const f32x4*)a.decoder;  f32x4* d0 = (f32x4*)lds;
        for (int i = tid; i < kDecoderFloats / 4; i += blockDim.x) d0[i] = s0[i];
        const f32x4* s1 = (const f32x4*)bwd_stream; f32x4* d1 = (f32x4*)bwd;
        for (int i = tid; i < kBwdFloats / 4; i += blockDim.x) d1[i] = s1[i];
    }
    __syncthreads();
    float* const Tf  = bwd + kBwdFloats + wave * kBwdWaveLds;     // [32 features][TP]
    float* const Tx  = Tf + 32 * TP;                              // [64 hidden][TP]: h, then da
    float* const Tdo = Tx + 64 * TP;                              // [32 colour channels][TP]
    float* const Tds = Tdo + 32 * TP;                             // [32 rays] dL/dsigma
    float* const Tt  = Tds + 32;                                  // the cooperative gather's tap table (rays 16..31)
    const int SN = NNETS - 1;
    const int S = a.Sc + a.Sf;

    // Work units ("tiles" of 32 lanes j).  Points: 32 consecutive points, tiles dealt grid-stride.  Rays: 32 consecutive SAMPLES OF ONE RAY — tile
    // t = ray t / tiles_per_ray, samples 32 (t % tiles_per_ray) + j — a wave takes pa.tiles_per_wave consecutive tiles.  (Round 2 gave a wave 32 rays
    // and walked their samples; the pass is order-free, and one ray per tile is what lets the scatter combine runs of equal texels.)
    const int tiles_per_ray = (S + 31) >> 5;
    const int n_tiles = POINTS ? (pa.total_pts + 31) / 32 : a.total_rays * tiles_per_ray;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int k_begin = POINTS ? (int)blockIdx.x * kBwdWaves + wave_u : ((int)blockIdx.x * kBwdWaves + wave_u) * pa.tiles_per_wave;
    const int k_end = POINTS ? n_tiles : min(n_tiles, k_begin + pa.tiles_per_wave);
    const int k_step = POINTS ? (int)gridDim.x * kBwdWaves : 1;
    if (k_begin >= k_end) return;
    int g = 0, n_img = 0;
    bool live = false;
    const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.planes, 0, a.planes_total_bytes, 0x00020000);
    unsigned img = 0, dimg = 0;                                   // d_planes is always the compact [N][3][H][W][32] (< 2^29 floats, checked on the host)
    const float cs = a.coord_scale;
    static_assert(TP == kFeatPitch, "T_f is the cooperative gather's tile");

    float dC[NNETS][16];                                          // rays: dL/dC (= 2 dL/dfeat) of this lane's channels; points: dL/drgb of the tile's point

    f32x16 aW2[NNETS][2], aW1[NNETS][2];                          // weight-gradient tiles (accumulate over all samples of the wave)
    float ab2[NNETS], ab1[NNETS][2], aW2s[2] = {0.f, 0.f}, ab2s = 0.f;
#pragma unroll
    for (int n = 0; n < NNETS; ++n) {
        ab2[n] = 0.f; ab1[n][0] = ab1[n][1] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { aW2[n][t][r] = 0.f; aW1[n][t][r] = 0.f; }
    }

    for (int k = k_begin; k < k_end; k += k_step) {
        float wgt = 1.f, dsig = 0.f, px, py, pz;
        if constexpr (POINTS) {
            g = min(k * 32 + j, pa.total_pts - 1);
            live = (k * 32 + j) < pa.total_pts;
            n_img = g / pa.pts_per_img;
            img = (unsigned)n_img * a.img_bytes;
            dimg = (unsigned)n_img * 3u * (unsigned)(a.H * a.W) * 32u;
            px = cs * pa.coords[(size_t)g * 3]; py = cs * pa.coords[(size_t)g * 3 + 1]; pz = cs * pa.coords[(size_t)g * 3 + 2];
            dsig = (live && pa.g_sigma) ? pa.g_sigma[g] : 0.f;
#pragma unroll
            for (int n = 0; n < NNETS; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) dC[n][r] = (live && pa.g_rgb) ? pa.g_rgb[(size_t)g * (NNETS * 32) + n * 32 + acc_row(r, h)] : 0.f;
        } else {
            g = k / tiles_per_ray;                                // wave-uniform
            const int ks = (k - g * tiles_per_ray) * 32 + j;      // this lane's sample
            live = ks < S;
            n_img = g / a.rays_per_img;
            img = (unsigned)n_img * a.img_bytes;
            dimg = (unsigned)n_img * 3u * (unsigned)(a.H * a.W) * 32u;
            const float ox = a.ray_o[g * 3 + 0], oy = a.ray_o[g * 3 + 1], oz = a.ray_o[g * 3 + 2];
            const float dx = a.ray_d[g * 3 + 0], dy = a.ray_d[g * 3 + 1], dz = a.ray_d[g * 3 + 2];
#pragma unroll
            for (int n = 0; n < NNETS; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) dC[n][r] = 2.f * a.g_feat[(size_t)g * (NNETS * 32) + n * 32 + acc_row(r, h)];
            const float4 rec = ((const float4*)a.tape_s)[(size_t)g * S + min(ks, S - 1)];      // z, colour weight, dL/dsigma
            const float z = rec.x;
            wgt = live ? rec.y : 0.f; dsig = live ? rec.z : 0.f;
            px = cs * fmaf(z, dx, ox); py = cs * fmaf(z, dy, oy); pz = cs * fmaf(z, dz, oz);
        }
        float feat[16];
        wave_sync();                                              // the previous sample's readers of T_f are done
        if constexpr ((kRbwdDbg & 4) != 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) feat[c] = 0.1f * (float)(c + h);
        } else
        gather_features_coop<true>(a, rsrc, img, lane, px, py, pz, Tf, Tt, feat);      // eight lanes to a texel; lands in T_f as [channel][ray] and in the lane's registers
        f32x16 df;
#pragma unroll
        for (int r = 0; r < 16; ++r) df[r] = 0.f;

#pragma unroll
        for (int n = 0; n < NNETS; ++n) {
            f32x16 h0, h1, o;
            mlp_layer1(lds, n, lane, h, feat, h0, h1);
            mlp_layer2(lds, n, lane, h, h0, h1, o);
            const bool squash = (n == 0) || (NNETS == 1) || a.sem_sigmoid;
            float dout[16];                                       // dL/d(decoder output 1 + channel)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float gsc = dC[n][r] * wgt;
                if (squash) { const float s = __builtin_amdgcn_rcpf(1.f + fast_exp(-o[r])); gsc *= 1.002f * s * (1.f - s); }
                dout[r] = gsc;
            }
            const float ds_n = (n == SN) ? dsig : 0.f;
            // ---- transposes: [channel][ray]
            wave_sync();                                          // previous net's readers of T_x / T_do are done
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, h);
                Tdo[row * TP + j] = dout[r];
                Tx[row * TP + j] = h0[r];
                Tx[(32 + row) * TP + j] = h1[r];
            }
            if (n == SN && h == 0) Tds[j] = ds_n;
            wave_sync();
            // ---- dW2 (colour rows) += do h^T ; db2 ; density row
            if constexpr (!(kRbwdDbg & 2)) {
                float fa[16], dsv[16];
                read_row16(Tdo + ti * TP + 16 * tk, fa);
                if (n == SN) read_row16(Tds + 16 * tk, dsv);
                float sb = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) sb += fa[s];
                ab2[n] += sb;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float fb[16];
                    read_row16(Tx + (32 * t + ti) * TP + 16 * tk, fb);
#pragma unroll
                    for (int s = 0; s < 16; ++s) aW2[n][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], aW2[n][t], 0, 0, 0);
                    if (n == SN) {
                        float sw = 0.f;
#pragma unroll
                        for (int s = 0; s < 16; ++s) sw = fmaf(dsv[s], fb[s], sw);
                        aW2s[t] += sw;
                    }
                }
                if (n == SN && h == 0) ab2s += ds_n;
            }
            // ---- dh = W2^T do (+ density row), da = dh * softplus'(pre-activation) = dh * (1 - exp(-h))
            f32x16 dh0, dh1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dh0[r] = 0.f; dh1[r] = 0.f; }
            if constexpr (!(kRbwdDbg & 8)) {
                const f32x4* wv = (const f32x4*)(bwd + n * kBwdNetStride) + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a0 = wv[q * 64], a1 = wv[(4 + q) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dh0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], dout[q * 4 + e], dh0, 0, 0, 0);
                        dh1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], dout[q * 4 + e], dh1, 0, 0, 0);
                    }
                }
            }
            if (n == SN) {
                const f32x4* w = (const f32x4*)(lds + OFF_W2S + h * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v0 = w[q], v1 = w[4 + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dh0[q * 4 + e] = fmaf(v0[e], ds_n, dh0[q * 4 + e]); dh1[q * 4 + e] = fmaf(v1[e], ds_n, dh1[q * 4 + e]); }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { dh0[r] *= (1.f - fast_exp(-h0[r])); dh1[r] *= (1.f - fast_exp(-h1[r])); }
            wave_sync();                                          // every lane has read h out of T_x
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, h);
                Tx[row * TP + j] = dh0[r];
                Tx[(32 + row) * TP + j] = dh1[r];
            }
            wave_sync();
            // ---- dW1 += da f^T ; db1
            if constexpr (!(kRbwdDbg & 2)) {
                float fbf[16];
                read_row16(Tf + ti * TP + 16 * tk, fbf);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float fa[16];
                    read_row16(Tx + (32 * t + ti) * TP + 16 * tk, fa);
                    float sb = 0.f;
#pragma unroll
                    for (int s = 0; s < 16; ++s) sb += fa[s];
                    ab1[n][t] += sb;
#pragma unroll
                    for (int s = 0; s < 16; ++s) aW1[n][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fbf[s], aW1[n][t], 0, 0, 0);
                }
            }
            // ---- df += W1^T da
            if constexpr (!(kRbwdDbg & 8)) {
                const f32x4* wv = (const f32x4*)(bwd + n * kBwdNetStride) + 8 * 64 + lane;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 av = wv[q * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int s = q * 4 + e;
                        const float b = (s < 16) ? dh0[s] : dh1[s - 16];
                        df = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b, df, 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wave_sync();                                              // T_x / T_do are free again: the scatter reuses them
        scatter_features(a, d_planes, dimg, lane, live, px, py, pz, df, Tx, (unsigned*)Tdo, Tdo + 384);
    }

    // ---- weight gradients leave through atomics (effective-weight gradients; the host applies the layer gains)
#pragma unroll
    for (int n = 0; n < NNETS; ++n) {
        float* gn = d_dec + n * kGradNetStride;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsafeAtomicAdd(gn + D_W2 + (1 + acc_row(r, tk)) * 64 + 32 * t + ti, aW2[n][t][r]);
                unsafeAtomicAdd(gn + D_W1 + (32 * t + acc_row(r, tk)) * 32 + ti, aW1[n][t][r]);
            }
        unsafeAtomicAdd(gn + D_B2 + 1 + ti, ab2[n]);
        unsafeAtomicAdd(gn + D_B1 + ti, ab1[n][0]);
        unsafeAtomicAdd(gn + D_B1 + 32 + ti, ab1[n][1]);
    }
    {
        float* gs = d_dec + SN * kGradNetStride;
        unsafeAtomicAdd(gs + D_W2 + ti, aW2s[0]);
        unsafeAtomicAdd(gs + D_W2 + 32 + ti, aW2s[1]);
        float s = ab2s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) unsafeAtomicAdd(gs + D_B2, s);
    }
}

} // namespace p3d

using namespace p3d;

extern "C" int p3d_render_bwd_decoder_floats(void) { return kBwdFloats; }
extern "C" int p3d_render_grad_decoder_floats(void) { return 2 * kGradNetStride; }

extern "C" int p3d_pack_decoder_bwd(const float* w1_a, const float* w2_a, const float* w1_b, const float* w2_b, int32_t n_nets, float lr_mul,
                                    float* packed_bwd, p3d_stream_t stream)
{
    P3D_REQUIRE(w1_a && w2_a && packed_bwd, "pack_decoder_bwd: null pointer");
    P3D_REQUIRE(n_nets == 1 || n_nets == 2, "pack_decoder_bwd: n_nets must be 1 or 2");
    P3D_REQUIRE(n_nets == 1 || (w1_b && w2_b), "pack_decoder_bwd: second net missing");
    PackBwdArgs p;
    p.w1[0] = w1_a; p.w2[0] = w2_a; p.w1[1] = w1_b; p.w2[1] = w2_b; p.n_nets = n_nets;
    p.wg1 = lr_mul / sqrtf(32.f); p.wg2 = lr_mul / sqrtf(64.f);
    hipLaunchKernelGGL(pack_decoder_bwd_kernel, dim3((kBwdFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, packed_bwd);
    count_launch(FAM_AUX);
    return check_launch("pack_decoder_bwd");
}

extern "C" int p3d_render_backward(const float* planes_cl, const float* decoder, const float* decoder_bwd, const float* ray_o, const float* ray_d,
                                   const float* u_coarse, const float* u_fine, const float* t_start, const float* t_end,
                                   const p3d_render_desc* d, const float* g_feat, const float* g_wsum, float* tape_intervals, float* tape_samples,
                                   float* d_planes_cl, float* d_decoder, p3d_stream_t stream)
{
    P3D_REQUIRE(d, "render_backward: null descriptor");
    P3D_REQUIRE(d->n_nets == 1 || d->n_nets == 2, "render_backward: n_nets must be 1 or 2");
    P3D_REQUIRE(planes_cl && decoder && decoder_bwd && ray_o && ray_d && u_coarse && u_fine && g_feat && tape_intervals && tape_samples && d_planes_cl && d_decoder,
                "render_backward: null pointer");
    P3D_REQUIRE((t_start == nullptr) == (t_end == nullptr), "render_backward: t_start/t_end must be given together");
    P3D_REQUIRE(d->plane_h >= 1 && d->plane_w >= 1 && d->box_warp != 0.f && d->rays_per_img >= 1 && d->n_img >= 0, "render_backward: bad sizes");
    if (d->depth_resolution < 4 || d->depth_resolution > kMaxS || d->depth_resolution_importance < 1 || d->depth_resolution_importance > kMaxS)
        return fail(P3D_ERR_UNSUPPORTED, "render_backward: needs 4 <= depth_resolution <= %d and 1 <= depth_resolution_importance <= %d", kMaxS, kMaxS);
    const int64_t total = (int64_t)d->n_img * d->rays_per_img;
    P3D_REQUIRE(total <= INT32_MAX / 64, "render_backward: too many rays");
    hipStream_t s = (hipStream_t)stream;
    const size_t plane_floats = (size_t)d->n_img * 3 * d->plane_h * d->plane_w * 32;
    if (hipMemsetAsync(d_planes_cl, 0, plane_floats * sizeof(float), s) != hipSuccess || hipMemsetAsync(d_decoder, 0, 2 * kGradNetStride * sizeof(float), s) != hipSuccess)
        return fail(P3D_ERR_LAUNCH, "render_backward: memset failed");
    if (total == 0) return P3D_OK;
    RenderArgs a{};
    a.H = d->plane_h; a.W = d->plane_w; a.Sc = d->depth_resolution; a.Sf = d->depth_resolution_importance;
    a.ray_start = d->ray_start; a.ray_end = d->ray_end; a.coord_scale = 2.f / d->box_warp;
    a.lin_step = a.Sc > 1 ? (d->ray_end - d->ray_start) / (float)(a.Sc - 1) : 0.f;
    a.disparity = d->disparity_space_sampling; a.white_back = d->white_back; a.sem_sigmoid = d->semantic_sigmoid;
    if (d->pixel_stride > 0) { a.plane_stride = d->plane_stride; a.pix_stride = d->pixel_stride; a.img_stride = d->image_stride; }
    else { a.plane_stride = (int64_t)a.H * a.W * 32; a.pix_stride = 32; a.img_stride = 3 * a.plane_stride; }
    a.plane_bytes = (unsigned)(a.plane_stride * 4); a.pix_bytes = (unsigned)(a.pix_stride * 4); a.img_bytes = (unsigned)(a.img_stride * 4);
    if ((int64_t)d->n_img * a.img_stride * 4 >= ((int64_t)1 << 31)) return fail(P3D_ERR_UNSUPPORTED, "render_backward: plane tensor too large for 32-bit buffer addressing");
    a.planes_total_bytes = (unsigned)((int64_t)d->n_img * a.img_stride * 4);
    a.planes = planes_cl; a.decoder = decoder; a.ray_o = ray_o; a.ray_d = ray_d; a.u_coarse = u_coarse; a.u_fine = u_fine;
    a.t_start = t_start; a.t_end = t_end; a.g_feat = g_feat; a.g_wsum = g_wsum; a.tape_i = tape_intervals; a.tape_s = tape_samples;
    a.total_rays = (int)total; a.rays_per_img = d->rays_per_img;
    { int r = 1; while (r * r < d->rays_per_img) ++r; a.res = (r * r == d->rays_per_img && d->raster_order) ? r : 0; }

    // 1. the forward sweep with tape
    {
        // a block is one-per-CU (LDS): small launches take fewer waves per block so that every CU still gets one
        int wpb = kWavesPerBlock;
        while (wpb > 2 && (total + wpb * 32 - 1) / (wpb * 32) < kNumCU) wpb >>= 1;
        const size_t lds_bytes = (size_t)(kDecoderFloats + kWavesPerBlock * (kWaveTile + kFeatTile + kTapTile)) * sizeof(float);
        const int blocks = (int)((total + wpb * 32 - 1) / (wpb * 32));
        if (d->n_nets == 1) {
            static std::atomic<uint64_t> once1_devs{0}; const hipError_t once1 = reserve_lds_once((const void*)render_forward_kernel<1, true>, (int)lds_bytes, once1_devs);
            if (once1 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_backward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once1));
            hipLaunchKernelGGL((render_forward_kernel<1, true>), dim3(blocks), dim3(wpb * 64), lds_bytes, s, a);
        } else {
            static std::atomic<uint64_t> once2_devs{0}; const hipError_t once2 = reserve_lds_once((const void*)render_forward_kernel<2, true>, (int)lds_bytes, once2_devs);
            if (once2 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_backward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once2));
            hipLaunchKernelGGL((render_forward_kernel<2, true>), dim3(blocks), dim3(wpb * 64), lds_bytes, s, a);
        }
        int rc = check_launch("render_backward (tape sweep)");
        if (rc != P3D_OK) return rc;
    }
    // 2. point-wise backward
    {
        const size_t lds_bytes = (size_t)(kDecoderFloats + kBwdFloats + kBwdWaves * kBwdWaveLds) * sizeof(float);
        // one block per CU at a time (LDS); a wave takes `tpw` consecutive (ray, 32-sample chunk) tiles: as many as keeps >= 2 rounds of blocks on
        // the chip, at most 96 (the weight-gradient tiles leave through atomics once per wave)
        const int S_all = a.Sc + a.Sf;
        const int64_t n_tiles = total * ((S_all + 31) / 32);
        int tpw = (int)(n_tiles / ((int64_t)kBwdWaves * 2 * kNumCU));
        tpw = tpw < 1 ? 1 : (tpw > 96 ? 96 : tpw);
        const int blocks = (int)((n_tiles + (int64_t)kBwdWaves * tpw - 1) / ((int64_t)kBwdWaves * tpw));
        PointArgs pa{}; pa.tiles_per_wave = tpw;
        if (d->n_nets == 1) {
            static std::atomic<uint64_t> once1_devs{0}; const hipError_t once1 = reserve_lds_once((const void*)render_backward_kernel<1>, (int)lds_bytes, once1_devs);
            if (once1 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_backward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once1));
            hipLaunchKernelGGL(render_backward_kernel<1>, dim3(blocks), dim3(kBwdWaves * 64), lds_bytes, s, a, decoder_bwd, d_planes_cl, d_decoder, pa);
        } else {
            static std::atomic<uint64_t> once2_devs{0}; const hipError_t once2 = reserve_lds_once((const void*)render_backward_kernel<2>, (int)lds_bytes, once2_devs);
            if (once2 != hipSuccess) return fail(P3D_ERR_LAUNCH, "render_backward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once2));
            hipLaunchKernelGGL(render_backward_kernel<2>, dim3(blocks), dim3(kBwdWaves * 64), lds_bytes, s, a, decoder_bwd, d_planes_cl, d_decoder, pa);
        }
    }
    count_launch(FAM_RENDER);
    return check_launch("render_backward");
}

// Backward of the point queries (p3d_sample_points): dL/dplanes (compact channels-last [N][3][H][W][32], zeroed here) and the
// effective-weight decoder gradients (the layout of p3d_render_backward's d_decoder) from dL/drgb [N*P][n_nets*32] and dL/dsigma [N*P]
// (either may be null = zero).  Reference: autograd through ImportanceRenderer.run_model (renderer.py:142-148), i.e. grid_sample's
// backward + the decoder's — what the density regularisation differentiates (loss.py:681-706).  Coordinates get no gradient.
extern "C" int p3d_sample_points_backward(const float* planes_cl, const float* decoder, const float* decoder_bwd, const float* coords,
                                          const p3d_render_desc* d, int32_t pts_per_img, const float* g_rgb, const float* g_sigma,
                                          float* d_planes_cl, float* d_decoder, p3d_stream_t stream)
{
    P3D_REQUIRE(d, "sample_points_backward: null descriptor");
    P3D_REQUIRE(d->n_nets == 1 || d->n_nets == 2, "sample_points_backward: n_nets must be 1 or 2");
    P3D_REQUIRE(planes_cl && decoder && decoder_bwd && coords && d_planes_cl && d_decoder, "sample_points_backward: null pointer");
    P3D_REQUIRE(d->plane_h >= 1 && d->plane_w >= 1 && d->box_warp != 0.f && pts_per_img >= 1 && d->n_img >= 0, "sample_points_backward: bad sizes");
    const int64_t total = (int64_t)d->n_img * pts_per_img;
    P3D_REQUIRE(total <= INT32_MAX / 64, "sample_points_backward: too many points");
    hipStream_t s = (hipStream_t)stream;
    const size_t plane_floats = (size_t)d->n_img * 3 * d->plane_h * d->plane_w * 32;
    if (hipMemsetAsync(d_planes_cl, 0, plane_floats * sizeof(float), s) != hipSuccess || hipMemsetAsync(d_decoder, 0, 2 * kGradNetStride * sizeof(float), s) != hipSuccess)
        return fail(P3D_ERR_LAUNCH, "sample_points_backward: memset failed");
    if (total == 0 || (!g_rgb && !g_sigma)) return P3D_OK;
    RenderArgs a{};
    a.H = d->plane_h; a.W = d->plane_w; a.coord_scale = 2.f / d->box_warp; a.sem_sigmoid = d->semantic_sigmoid;
    if (d->pixel_stride > 0) { a.plane_stride = d->plane_stride; a.pix_stride = d->pixel_stride; a.img_stride = d->image_stride; }
    else { a.plane_stride = (int64_t)a.H * a.W * 32; a.pix_stride = 32; a.img_stride = 3 * a.plane_stride; }
    a.plane_bytes = (unsigned)(a.plane_stride * 4); a.pix_bytes = (unsigned)(a.pix_stride * 4); a.img_bytes = (unsigned)(a.img_stride * 4);
    if ((int64_t)d->n_img * a.img_stride * 4 >= ((int64_t)1 << 31) || (int64_t)d->plane_h * d->plane_w >= (1 << 24) || d->pixel_stride * 4 >= (1 << 16))
        return fail(P3D_ERR_UNSUPPORTED, "sample_points_backward: plane tensor too large for 32-bit buffer addressing");
    a.planes_total_bytes = (unsigned)((int64_t)d->n_img * a.img_stride * 4);
    a.planes = planes_cl; a.decoder = decoder;
    PointArgs pa{coords, g_rgb, g_sigma, pts_per_img, (int)total, 0};
    const size_t lds_bytes = (size_t)(kDecoderFloats + kBwdFloats + kBwdWaves * kBwdWaveLds) * sizeof(float);
    const int64_t tiles = (total + 31) / 32;
    int blocks = (int)((tiles + kBwdWaves - 1) / kBwdWaves);
    if (blocks > kNumCU) blocks = kNumCU;                                 // one block per CU (LDS); waves take further tiles grid-stride
    if (d->n_nets == 1) {
        static std::atomic<uint64_t> once1_devs{0}; const hipError_t once1 = reserve_lds_once((const void*)render_backward_kernel<1, true>, (int)lds_bytes, once1_devs);
        if (once1 != hipSuccess) return fail(P3D_ERR_LAUNCH, "sample_points_backward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once1));
        hipLaunchKernelGGL((render_backward_kernel<1, true>), dim3(blocks), dim3(kBwdWaves * 64), lds_bytes, s, a, decoder_bwd, d_planes_cl, d_decoder, pa);
    } else {
        static std::atomic<uint64_t> once2_devs{0}; const hipError_t once2 = reserve_lds_once((const void*)render_backward_kernel<2, true>, (int)lds_bytes, once2_devs);
        if (once2 != hipSuccess) return fail(P3D_ERR_LAUNCH, "sample_points_backward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(once2));
        hipLaunchKernelGGL((render_backward_kernel<2, true>), dim3(blocks), dim3(kBwdWaves * 64), lds_bytes, s, a, decoder_bwd, d_planes_cl, d_decoder, pa);
    }
    count_launch(FAM_RENDER);
    return check_launch("sample_points_backward");
}
