"""Diagnostic (GPU box): where a wave of the split-precision data gradient (nerf_bwd_data_h_kernel) spends an
item.  Builds a throw-away copy of csrc/ with s_memtime stamps (scalar-only: s_memtime + s_store_dwordx2, no
vector register and no vmcnt entry is added, so the counted hand-over waits see the product's queue) and prints,
per hidden layer, the durations of: LayerNorm backward (pass 1 + 2, dY stores issued) | x_hat loads issued +
row maximum + f16 split | loop entry -> hand-over 1 | -> 2 | -> 3 (the first that must see the 33 memory
operations retired) | -> 8 (half) | -> loop end.
usage: python scripts/experiments/stamps_dgrad_h.py [rays]"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

STAMP_MACRO = r'''
#define STAMP()                                                                                           \
    do {                                                                                                  \
        if (stamp_on && stamp_off < 2040u) {                                                              \
            uint64_t t_;                                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, %2"           \
                         : "=&s"(t_) : "s"(stamp_buf), "s"(stamp_off) : "memory");                        \
            stamp_off += 8u;                                                                              \
        }                                                                                                 \
    } while (0)
struct StampHook {
    GammaBetaTurn& turn;
    const bool stamp_on;
    uint64_t* const stamp_buf;
    uint32_t& stamp_off;
    __device__ __forceinline__ void operator()(int t) const {
        turn(t);
        if (t == 1 || t == 2 || t == 3 || t == 8) STAMP();
    }
};
'''


VARIANTS = {
    # timing-only variants (wrong results): what the phases wait for
    "nostores": [("        *(f32x4*)(dy_row + T * kTileT) = dy;", "        if (kScaled) asm volatile(\"\" :: \"v\"(dy)); else *(f32x4*)(dy_row + T * kTileT) = dy;")],
    "noloads": [("                    rstd = (ws_stat + ba.L.rstd[L - 1])[lane_word(j)];", "                    rstd = 1.0f; asm volatile(\"\" : \"+v\"(rstd));"),
                ("                    for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xrow_n + ro + T * kTileT);",
                 "                    for (int T = 0; T < 16; ++T) { xh[T] = f32x4{0.5f, -0.25f, 0.125f, 1.0f} * (float)ro; asm volatile(\"\" : \"+v\"(xh[T])); }")],
    "nonote": [("                note_max(wmax + L, amax, lane);", "")],
    # no weight stream: the LDS-DMA of every stage is skipped (the ring keeps whatever it held; hand-over waits and
    # barriers stay) — what the data gradient's own loads and saves cost when they do not share the CU's vector-memory
    # path with 20 GB/s of weights
    "nodma": [("                \"global_load_lds_dwordx4 %1, %3\\n\\t\"\n                \"global_load_lds_dwordx4 %1, %3 offset:1024\\n\\t\"\n                \"global_load_lds_dwordx4 %1, %3 offset:2048\\n\\t\"\n                \"global_load_lds_dwordx4 %1, %3 offset:3072\\n\\t\"\n", "")],
}


def patch(src, variant=None):
    s = open(src).read()
    s = s.replace("constexpr int kYoungerL5 = 17, kYoungerHidden = 33;",
                  "constexpr int kYoungerL5 = 17, kYoungerHidden = 33;" + STAMP_MACRO)
    head, kern = s.split("__global__ __launch_bounds__(256, 2) void nerf_bwd_data_h_kernel", 1)
    kern, tail = kern.split("// All six layers in ONE launch", 1)
    kern = kern.replace('''    pipe.issue();
    pipe.issue();
    __syncthreads();
''', '''    pipe.issue();
    pipe.issue();
    __syncthreads();
    // workgroups b and b + gridDim/2 share a CU (round-robin dispatch): stamp WG 0 and its partner
    const bool stamp_on = (blockIdx.x % (gridDim.x / 2)) == 0;
    uint64_t* const stamp_buf = (uint64_t*)(ba.dymax + (size_t)kMaxDataGrid * 8) + ((blockIdx.x / (gridDim.x / 2)) * 4 + wave) * 256;
    uint32_t stamp_off = 0;
''', 1)
    assert kern.count("            f32x4 dout[4];\n") == 1
    kern = kern.replace("            f32x4 dout[4];\n", "            STAMP();\n            f32x4 dout[4];\n", 1)
    # finer: the 17 loads issued | row maximum + note_max | split
    anchor = "                    for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xrow_n + ro + T * kTileT);\n                }\n"
    assert kern.count(anchor) == 1
    kern = kern.replace(anchor, anchor + "                STAMP();\n")
    anchor = "                note_max(wmax + L, amax, lane);\n"
    assert kern.count(anchor) == 1
    kern = kern.replace(anchor, anchor + "                STAMP();\n")
    kern = kern.replace("                layer_wide_h<2, kYoungerL5>(pipe, acc, bh, bl, TurnHook{turn});",
                        "                STAMP();\n                layer_wide_h<2, kYoungerL5>(pipe, acc, bh, bl, TurnHook{turn});\n                STAMP();")
    assert kern.count("                if (L == 0) {\n") == 1
    kern = kern.replace("                if (L == 0) {\n", "                STAMP();\n                if (L == 0) {\n")
    kern = kern.replace("                layer_wide_h<8, kYoungerHidden>(pipe, acc, bh, bl, TurnHook{turn});",
                        "                STAMP();\n                layer_wide_h<8, kYoungerHidden>(pipe, acc, bh, bl, StampHook{turn, stamp_on, stamp_buf, stamp_off});\n                STAMP();")
    kern = kern.replace("    if (threadIdx.x < 8) ba.dymax[", "    asm volatile(\"s_dcache_wb\" ::: \"memory\");\n    if (threadIdx.x < 8) ba.dymax[")
    s = head + "__global__ __launch_bounds__(256, 2) void nerf_bwd_data_h_kernel" + kern + "// All six layers in ONE launch" + tail
    s = s.replace("(size_t)kMaxDataGrid * (kGbFloats + 8)) * sizeof(float);",
                  "(size_t)kMaxDataGrid * (kGbFloats + 8)) * sizeof(float) + 8 * 256 * 8;")
    assert s.count("STAMP()") >= 8
    if variant:
        common = os.path.join(os.path.dirname(src), "nerf_backward_common.h")
        device = os.path.join(os.path.dirname(src), "nerf_device.h")
        c, dv = open(common).read(), open(device).read()
        for name in variant.split("+"):
            for old, new in VARIANTS[name]:
                if old in s:
                    s = s.replace(old, new)
                elif old in c:
                    c = c.replace(old, new)
                else:
                    assert old in dv, (name, old)
                    dv = dv.replace(old, new)
        open(device, "w").write(dv)
        open(common, "w").write(c)
    if variant and ("noloads" in variant or "nostores" in variant):
        n = 33 - (17 if "noloads" in variant else 0) - (16 if "nostores" in variant else 0)
        s = s.replace("kYoungerHidden = 33;", f"kYoungerHidden = {n};")
    open(src, "w").write(s)


def main():
    if "NERF_HIP_LIB" not in os.environ:
        from nerf_amd import build as B
        work = tempfile.mkdtemp(prefix="stamps_")
        csrc = os.path.join(work, "csrc")
        shutil.copytree(B.CSRC, csrc, ignore=shutil.ignore_patterns("*.so*"))
        variant = os.environ.get("STAMP_VARIANT") or None
        patch(os.path.join(csrc, "nerf_backward.hip"), variant)
        B.CSRC = csrc
        out = os.path.join(ROOT, "gpurun_out", f"libnerf_hip_stamps_{variant or 'base'}.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        B.build(out=out, defines=["NERF_HIP_EXPERIMENT=stamps_dgrad_h"], force=True)
        sys.exit(subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, NERF_HIP_LIB=out)).returncode)
    import torch
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = NeRF().to(dev)
    model.train_precision = "f16x3"
    n, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64
    o, d, tgt = torch.randn(n, 3, device=dev), torch.randn(n, 3, device=dev), torch.rand(n, 3, device=dev)
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0)
        ((rgb - tgt.unsqueeze(1)) ** 2).mean().backward()
    torch.cuda.synchronize()
    raw = model._scratch_buf.view(torch.int64)[-8 * 256:].cpu().view(8, 256)
    # item | L5: loop start, loop end | L = 4..1: LN end, loop start, ho1, ho2, ho3, ho8, loop end (+1: see below) | L0: LN end
    names = ["item", "L5 start", "L5 end"]
    for L in (4, 3, 2, 1):
        names += [f"L{L} LN", f"L{L} loads", f"L{L} max", f"L{L} split", f"L{L} ho1", f"L{L} ho2", f"L{L} ho3", f"L{L} ho8", f"L{L} end", None]
    names = [x for x in names if x is not None] + ["L0 LN"]
    per_item = len(names)
    import statistics
    phase = {}
    items = []
    for w in range(8):
        ts = [int(v) for v in raw[w] if int(v) != 0]
        for it in range(len(ts) // per_item - 1):
            seg = ts[it * per_item:(it + 1) * per_item + 1]
            items.append(seg[-1] - seg[0])
            for i in range(1, len(seg)):
                nm = names[i] if i < per_item else "next item"
                key = nm.split(" ", 1)[1] if nm[0] == "L" and nm[1] in "1234" else nm
                phase.setdefault(key, []).append(seg[i] - seg[i - 1])
    print(f"variant {os.environ.get('STAMP_VARIANT') or 'base'}: {len(items)} items of 8 waves (2 workgroups of one CU); "
          f"median item {statistics.median(items):.0f} cycles")
    for key in ("LN", "loads", "max", "split", "ho1", "ho2", "ho3", "ho8", "end", "L5 start", "L5 end", "L0 LN", "next item"):
        v = phase.get(key, [0])
        print(f"  {key:>9}: median {statistics.median(v):7.0f}  min {min(v):7.0f}  max {max(v):7.0f} cycles")
    print("(s_memtime ticks = shader cycles; hidden layers L4..L1 pooled: LN = LayerNorm backward incl. the 16 dY stores;"
          " loads = the 17 x_hat loads issued; max = row maximum + note_max; split = the f16 split; ho1/2/3/8 = hand-over of stage 1/2/3/8; end = loop end)")


if __name__ == "__main__":
    main()
