"""What-if builds of the full-line render kernel (nerf_fused_line_kernel): the kernel source is patched so that one ingredient at a time
is left out (results are WRONG by construction) and each variant is linked into its own libcd360_nw<bits>.so; time them on the GPU box with
    for nw in 1 2 4 8 16 31; do CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_nw$nw.so python tools/bench_kernels.py nerf; done
bits: 1 no row gathers (bpermute, loads, LDS staging), 2 no sin / cos, 4 no SiLU transcendentals, 8 no MFMAs, 16 view 0's geometry reused for
every view (no projection / corner arithmetic, and every gather hits the same lines).  Run HERE (cross-compiles): python tools/probe/nerf_whatif.py
Round-3 measurement (640 channels, r 64, 50 views, b 3; us): full 8574 | 1: 6046 | 2: 8059 | 4: 7803 | 8: 7550 | 16: 6572 | 31: 2449."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "custom-diffusion360_amd")
src = open(os.path.join(PKG, "csrc", "nerf_fused.hip")).read()
i0 = src.index("__global__ __launch_bounds__(256, 2) void nerf_fused_line_kernel")
i1 = src.index("#ifdef CD360_WHATIF  // probe builds only")
k = src[i0:i1]
EDITS = [
    ("    auto load_rows = [&](const Geo& G, int half) {\n#pragma unroll", "    auto load_rows = [&](const Geo& G, int half) {\n      if (NW & 1) return;\n#pragma unroll"),
    ("    auto store_rows = [&](int half) {\n#pragma unroll", "    auto store_rows = [&](int half) {\n      if (NW & 1) return;\n#pragma unroll"),
    ("            y[mb][t] = *reinterpret_cast<const u32x4*>(slots + c * SLOT_BYTES + l31 * 128 + (((4 * mb + 2 * hh + t) ^ rswz) << 4));",
     "            y[mb][t] = (NW & 1) ? u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : *reinterpret_cast<const u32x4*>(slots + c * SLOT_BYTES + l31 * 128 + (((4 * mb + 2 * hh + t) ^ rswz) << 4));"),
    ("            fw[pr] = pack_bf16x2(__builtin_amdgcn_sinf(rev), __builtin_amdgcn_cosf(rev));",
     "            fw[pr] = (NW & 2) ? pack_bf16x2(rev, rev) : pack_bf16x2(__builtin_amdgcn_sinf(rev), __builtin_amdgcn_cosf(rev));"),
    ("            const f32x2 den = one + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};\n            const f32x2 sv = zz * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};",
     "            const f32x2 den = (NW & 4) ? one + t : one + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};\n            const f32x2 sv = (NW & 4) ? zz * den : zz * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};"),
    ("          z[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb, z[mb], 0, 0, 0);\n        }\n        __builtin_amdgcn_sched_barrier(0);  // one k-step per scheduling region",
     "          if (NW & 8) { z[mb][ks] += __builtin_bit_cast(float, fv[mb]) + __builtin_bit_cast(float, __builtin_bit_cast(u32x4, a)[0]); } else\n          z[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb, z[mb], 0, 0, 0);\n        }\n        __builtin_amdgcn_sched_barrier(0);  // one k-step per scheduling region"),
    ("        geometry(iv + 1, nxt);\n        load_rows(nxt, 0);", "        if (NW & 16) nxt = cur; else geometry(iv + 1, nxt);\n        load_rows(nxt, 0);"),
]
for old, new in EDITS:
    assert old in k, old[:60]
    k = k.replace(old, new)
os.makedirs("/tmp/cd360_nw", exist_ok=True)
for h in os.listdir(os.path.join(PKG, "csrc")):
    if h.endswith(".h"):
        open(os.path.join("/tmp/cd360_nw", h), "w").write(open(os.path.join(PKG, "csrc", h)).read())
open("/tmp/cd360_nw/nerf_whatif.hip", "w").write(src[:i0] + "#ifndef NW\n#define NW 0\n#endif\n" + k + src[i1:])
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
objs = [os.path.join(PKG, "lib", "obj", f[:-4] + ".o") for f in sorted(os.listdir(os.path.join(PKG, "csrc"))) if f.endswith(".hip") and f != "nerf_fused.hip"]
for nw in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 31]:
    o = f"/tmp/cd360_nw/nfw_{nw}.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-DNW={nw}", "-c", "/tmp/cd360_nw/nerf_whatif.hip", "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(PKG, "lib", f"libcd360_nw{nw}.so"), *objs, o])
    print("built", f"libcd360_nw{nw}.so")
