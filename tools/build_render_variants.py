"""Ablation builds of the fused ray-marcher (measurement only — results are wrong by construction, only time is read).  The shipped sources stay untouched (their
hash pins the committed counter passes): csrc is copied to /tmp, render_device.h is patched there, and each variant is linked against the product's other objects into
pix2pix3d_amd/libp3d_hip_rv<bits>.so (git-ignored; travels to the GPU box; select with P3D_LIB_PATH).
    python tools/build_render_variants.py 1 2 3 4 8 16
bits: 32 kRaysB = 8, 64 softplus without the threshold select, 128 sigmoid without the exponent's pre-multiply, 1 no gather (features from the lane id), 2 no transcendentals in the decoder (softplus / sigmoid -> a multiply), 4 no MFMAs (operands kept alive),
      8 no sched_barrier between the two nets of a sample, 16 exact-fp32 layer 2 on two accumulators, 1024 split8 written on pairs (correct results)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pix2pix3d_amd')
sys.path.insert(0, ROOT)
from pix2pix3d_amd import build as B  # noqa: E402

B.build_library()
work = '/tmp/p3d_rv'
shutil.rmtree(work, ignore_errors=True)
shutil.copytree(os.path.join(PKG, 'csrc'), os.path.join(work, 'pix2pix3d_amd', 'csrc'), ignore=shutil.ignore_patterns('_obj'))
shutil.copytree(os.path.join(ROOT, 'include'), os.path.join(work, 'include'))
hdr = os.path.join(work, 'pix2pix3d_amd', 'csrc', 'render_device.h')
s = open(hdr).read()


def rep(old, new, count=1):
    global s
    assert s.count(old) >= 1, old[:60]
    s = s.replace(old, new) if count == 0 else s.replace(old, new, count)


rep('namespace p3d {\n', 'namespace p3d {\n#ifndef P3D_RENDER_DEBUG\n#define P3D_RENDER_DEBUG 0\n#endif\n'
    '#if P3D_RENDER_DEBUG & 4\n#define P3D_MFMA(fn, a, b, c) ({ asm volatile("" :: "v"(a), "v"(b)); (c); })\n#else\n#define P3D_MFMA(fn, a, b, c) fn(a, b, c, 0, 0, 0)\n#endif\n')
# 2: transcendentals
rep('__device__ __forceinline__ float softplus20_log2(float xs) {', '__device__ __forceinline__ float softplus20_log2(float xs) {\n    if (P3D_RENDER_DEBUG & 2) return xs * 0.5f;')
rep('__device__ __forceinline__ float sigmoid_clamped(float x) {  // sigmoid(x) * (1 + 2*0.001) - 0.001', '__device__ __forceinline__ float sigmoid_clamped(float x) {\n    if (P3D_RENDER_DEBUG & 2) return x * 0.1f;')
# 1: gather
rep('    const int sub = lane >> 3, chunk = lane & 7, j = lane & 31, h = lane >> 5;\n',
    '    if (P3D_RENDER_DEBUG & 1) {\n#pragma unroll\n        for (int c = 0; c < 16; ++c) feat[c] = px * (float)(c + 1) + py + pz * (float)lane * 1e-3f;\n        return;\n    }\n'
    '    const int sub = lane >> 3, chunk = lane & 7, j = lane & 31, h = lane >> 5;\n')
# 4: every MFMA of the decoder through the macro
import re
s = re.sub(r'__builtin_amdgcn_mfma_f32_32x32x2f32\(([^;]*?), 0, 0, 0\)', lambda m: 'P3D_MFMA(__builtin_amdgcn_mfma_f32_32x32x2f32, ' + m.group(1) + ')', s)
s = re.sub(r'__builtin_amdgcn_mfma_f32_32x32x16_bf16\(([^;]*?), 0, 0, 0\)', lambda m: 'P3D_MFMA(__builtin_amdgcn_mfma_f32_32x32x16_bf16, ' + m.group(1) + ')', s)
# 8: the barrier that closes a net's share of a sample
rep('                prev[n][r] = c;\n            }\n            __builtin_amdgcn_sched_barrier(0);\n', '                prev[n][r] = c;\n            }\n            if (!(P3D_RENDER_DEBUG & 8)) __builtin_amdgcn_sched_barrier(0);\n')
# 32: eight rays at a time through the importance sampler; 64: softplus without its threshold select; 128: sigmoid without the pre-multiply of its exponent
rep('constexpr int kRaysB = 4; ', 'constexpr int kRaysB = (P3D_RENDER_DEBUG & 32) ? 8 : 4; ')
rep('    return xs > 28.853900817779268f ? xs : __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(xs));', '    if (P3D_RENDER_DEBUG & 64) return __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(xs));\n    if (P3D_RENDER_DEBUG & 256) return fmaxf(xs, __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(fminf(xs, 127.f))));\n    if (P3D_RENDER_DEBUG & 512) return __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(fminf(xs, 127.f)));\n    return xs > 28.853900817779268f ? xs : __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(xs));')
rep('    return fmaf(__builtin_amdgcn_rcpf(1.f + fast_exp(-x)), 1.002f, -0.001f);', '    if (P3D_RENDER_DEBUG & 128) return fmaf(__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x)), 1.002f, -0.001f);\n    return fmaf(__builtin_amdgcn_rcpf(1.f + fast_exp(-x)), 1.002f, -0.001f);')
# 16: exact layer 2 on two accumulators (even / odd k-steps), summed at the end — does the single dependent chain of 32 MFMAs stall?
rep("""            const float b = (s < 16) ? h0[s] : h1[s - 16];
            out = P3D_MFMA(__builtin_amdgcn_mfma_f32_32x32x2f32, a[e], b, out);""",
    """            const float b = (s < 16) ? h0[s] : h1[s - 16];
            if ((P3D_RENDER_DEBUG & 16) && (e & 1)) out2 = P3D_MFMA(__builtin_amdgcn_mfma_f32_32x32x2f32, a[e], b, out2);
            else out = P3D_MFMA(__builtin_amdgcn_mfma_f32_32x32x2f32, a[e], b, out);""")
rep("""    const f32x4* wv = (const f32x4*)(lds + n * kNetStride) + 8 * 64 + lane;      // steps 32..63
""", """    const f32x4* wv = (const f32x4*)(lds + n * kNetStride) + 8 * 64 + lane;      // steps 32..63
    f32x16 out2;
#pragma unroll
    for (int r = 0; r < 16; ++r) out2[r] = 0.f;
""")
rep("""            else out = P3D_MFMA(__builtin_amdgcn_mfma_f32_32x32x2f32, a[e], b, out);
        }
    }
}
""", """            else out = P3D_MFMA(__builtin_amdgcn_mfma_f32_32x32x2f32, a[e], b, out);
        }
    }
    if (P3D_RENDER_DEBUG & 16) out = out + out2;
}
""")
# 1024: split8 on pairs (one v_cvt_pk_bf16_f32 per two pieces, the piece back to fp32 by shift / mask) — same bits out; does the decoder's vector share shrink?
rep("""#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e];
        const __bf16 hx = (__bf16)x;
        hi[e] = hx;
        lo[e] = (__bf16)(x - (float)hx);
    }
    asm volatile("s_nop 4" : "+v"(hi), "+v"(lo));""", """    if (P3D_RENDER_DEBUG & 1024) {
        typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t hh, ll;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x2_t x = {v[2 * p], v[2 * p + 1]};
            hh[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf2_t));
            const f32x2_t hf = {__builtin_bit_cast(float, hh[p] << 16), __builtin_bit_cast(float, hh[p] & 0xffff0000u)};
            ll[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x - hf, bf2_t));
        }
        hi = __builtin_bit_cast(bf8, hh); lo = __builtin_bit_cast(bf8, ll);
    } else
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e];
        const __bf16 hx = (__bf16)x;
        hi[e] = hx;
        lo[e] = (__bf16)(x - (float)hx);
    }
    asm volatile("s_nop 4" : "+v"(hi), "+v"(lo));""")
# 2048 / 4096: the compiler's MFMA-interleaving strategies (iglp_opt 0 / 1) for the scheduling region of each net of a sample (fine pass)
rep("""            const int n = (idx == 0) ? SN : idx - 1;
            f32x16 h0, h1, o;
            if constexpr (BF3)  mlp_layer1_bf3(lds, n, lane, h, fh, fl, h0, h1);""", """            const int n = (idx == 0) ? SN : idx - 1;
            f32x16 h0, h1, o;
            if (P3D_RENDER_DEBUG & 2048) __builtin_amdgcn_iglp_opt(0);
            if (P3D_RENDER_DEBUG & 4096) __builtin_amdgcn_iglp_opt(1);
            if constexpr (BF3)  mlp_layer1_bf3(lds, n, lane, h, fh, fl, h0, h1);""")
open(hdr, 'w').write(s)

objs = [os.path.join(B.OBJ_DIR, f) for f in os.listdir(B.OBJ_DIR) if f.endswith('.o') and f not in ('render.o', 'hazard_probe.o', 'mfma_rate_probe.o')]
for bits in [int(v) for v in sys.argv[1:]]:
    obj = f'/tmp/p3d_rv/render_{bits}.o'
    cmd = [B._hipcc()] + B.CXXFLAGS + [f'-DP3D_RENDER_DEBUG={bits}', '-c', os.path.join(work, 'pix2pix3d_amd', 'csrc', 'render.hip'), '-o', obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-3000:]); sys.exit(1)
    out = os.path.join(PKG, f'libp3d_hip_rv{bits}.so')
    r = subprocess.run([B._hipcc(), '-shared', '-fPIC', f'--offload-arch={B.ARCH}', '-o', out] + objs + [obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-3000:]); sys.exit(1)
    print('built', out, flush=True)
