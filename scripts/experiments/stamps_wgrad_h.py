"""Diagnostic (GPU box): the timeline of the split-precision weight gradient's workgroups (nerf_wgrad_h_kernel).
Builds a throw-away copy of csrc/ in which wave 0 of every workgroup stamps (s_memrealtime, 100 MHz; scalar stores
only) its entry, the end of its batch-maximum reduction, the first hand-over of the ring (three k-steps of DMA
issued, two landed), the end of its k-step loop and its exit, with the XCD / CU it ran on.  Prints per job kind the
median duration of each phase and, per CU, how the rounds of workgroups follow each other.
usage: python scripts/experiments/stamps_wgrad_h.py [rays]"""
import os
import shutil
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

STAMP = r'''
#define WSTAMP(i)                                                                                         \
    do {                                                                                                  \
        if (wave == 0 && ba.stamps != nullptr) {                                                          \
            uint64_t t_;                                                                                  \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, %2"       \
                         : "=&s"(t_) : "s"(ba.stamps), "n"(8 * (i)) : "memory");                          \
        }                                                                                                 \
    } while (0)
'''


INNER = r'''
// s_memtime stamps INSIDE the k-steps of workgroup 0, wave 0 (shader cycles): slots 6144.. of the stamp area
#define ISTAMP()                                                                                          \
    do {                                                                                                  \
        if (inner_on && inner_off < 8u * 2040u) {                                                         \
            uint64_t t_;                                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, %2"       \
                         : "=&s"(t_) : "s"(inner_buf), "s"(inner_off) : "memory");                        \
            inner_off += 8u;                                                                              \
        }                                                                                                 \
    } while (0)
'''


def patch(csrc):
    p = os.path.join(csrc, "nerf_backward_common.h")
    s = open(p).read()

    def rep(a, b):
        nonlocal s
        assert s.count(a) == 1, (a, s.count(a))
        s = s.replace(a, b)
    rep("struct WgradJob {\n", STAMP + INNER + "struct WgradJob {\n    uint64_t* stamps;                             // EXPERIMENT: 8 slots of this workgroup\n")
    rep("    int out0, in0;                                // first 32-wide tile of this wave\n",
        "    WSTAMP(0);\n    if (wave == 0 && ba.stamps != nullptr) {\n        uint32_t hw, xcc;\n"
        "        asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\\n\\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)\" : \"=s\"(hw), \"=s\"(xcc));\n"
        "        const uint64_t id = ((uint64_t)xcc << 32) | hw;\n"
        "        asm volatile(\"s_store_dwordx2 %0, %1, 56\" :: \"s\"(id), \"s\"(ba.stamps) : \"memory\");\n    }\n"
        "    int out0, in0;                                // first 32-wide tile of this wave\n")
    rep("    f32x16 acc[Sh::kTo][Sh::kTi];\n", "    WSTAMP(1);\n    const bool inner_on = wave == 0 && blockIdx.x == 0 && ba.stamps != nullptr;\n"
        "    uint64_t* const inner_buf = ba.stamps + 6144;\n    uint32_t inner_off = 0;\n    f32x16 acc[Sh::kTo][Sh::kTi];\n")
    # around the dY half of the DMA issue (in front of slot 0)
    rep("                if (a == 0) ring_issue_part<Sh, 0>(dy, x, sample_begin + t_fill * kRingStep, fill, wave, lane);\n                if (a == 1)",
        "                if (a == 0) {\n                    ISTAMP();\n                    ring_issue_part<Sh, 0>(dy, x, sample_begin + t_fill * kRingStep, fill, wave, lane);\n                    ISTAMP();\n                }\n                if (a == 1)")
    # one stamp in front of every slot of a k-step, in front of the hand-over wait, behind it, behind the barrier
    rep("            const Operand& ac = at[a & 1];\n            Operand& an = at[(a & 1) ^ 1];\n            __builtin_amdgcn_sched_barrier(0);\n",
        "            const Operand& ac = at[a & 1];\n            Operand& an = at[(a & 1) ^ 1];\n            __builtin_amdgcn_sched_barrier(0);\n            ISTAMP();\n            __builtin_amdgcn_sched_barrier(0);\n")
    rep("        asm volatile(\"s_waitcnt vmcnt(%c0) lgkmcnt(0)\" ::\"n\"(P::kPerWave) : \"memory\");\n        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n    };",
        "        ISTAMP();\n        asm volatile(\"s_waitcnt vmcnt(%c0) lgkmcnt(0)\" ::\"n\"(P::kPerWave) : \"memory\");\n        ISTAMP();\n        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        ISTAMP();\n    };")
    rep("        Operand b0[Sh::kTi], b1[Sh::kTi];\n", "        WSTAMP(2);\n        Operand b0[Sh::kTi], b1[Sh::kTi];\n")
    rep("    float* slab = ba.slab;\n    const int col = lane & 31, half = lane >> 5;\n",
        "    WSTAMP(3);\n    float* slab = ba.slab;\n    const int col = lane & 31, half = lane >> 5;\n")
    # the exit stamp: after the slab stores have been issued AND retired
    rep("        if (in0 == 0 && half == 0) slab[b_off + 32 * (out0 + a) + col] = both;\n    }\n}\n",
        "        if (in0 == 0 && half == 0) slab[b_off + 32 * (out0 + a) + col] = both;\n    }\n"
        "    WSTAMP(4);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    WSTAMP(5);\n"
        "    asm volatile(\"s_dcache_wb\" ::: \"memory\");\n}\n")
    open(p, "w").write(s)
    for name, args in (("nerf_backward.hip", "BwdArgs"), ("nerf_legacy_backward.hip", "LBwdArgs")):
        p = os.path.join(csrc, name)
        s = open(p).read()
        s = s.replace("const WgradJob jb{ba.tiles_per_split,", "const WgradJob jb{STAMPS_PTR, ba.tiles_per_split,")
        if name == "nerf_backward.hip":
            # stamps live behind the data gradient's per-workgroup maxima in the backward scratch
            s = s.replace("const WgradJob jb{STAMPS_PTR,", "const WgradJob jb{(uint64_t*)(ba.dymax + (size_t)kMaxDataGrid * 8) + (size_t)blockIdx.x * 8,")
            a = "(size_t)kMaxDataGrid * (kGbFloats + 8)) * sizeof(float);"
            assert s.count(a) == 1
            s = s.replace(a, "(size_t)kMaxDataGrid * (kGbFloats + 8)) * sizeof(float) + 1024 * 64;")
        else:
            s = s.replace("const WgradJob jb{STAMPS_PTR,", "const WgradJob jb{nullptr,")
        open(p, "w").write(s)


def main():
    if "NERF_HIP_LIB" not in os.environ:
        from nerf_amd import build as B
        work = tempfile.mkdtemp(prefix="wstamps_")
        csrc = os.path.join(work, "csrc")
        shutil.copytree(B.CSRC, csrc, ignore=shutil.ignore_patterns("*.so*"))
        patch(csrc)
        B.CSRC = csrc
        out = os.path.join(ROOT, "nerf_amd", "csrc", "libnerf_hip_wstamps.so")
        B.build(out=out, defines=["NERF_HIP_EXPERIMENT=stamps_wgrad_h"], force=True)
        if "--build-only" in sys.argv:
            return
        sys.exit(subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, NERF_HIP_LIB=out)).returncode)
    import torch
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = NeRF().to(dev)
    model.train_precision = "f16x3"
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n, S = int(args[0]) if args else 4096, 64
    o, d, tgt = torch.randn(n, 3, device=dev), torch.randn(n, 3, device=dev), torch.rand(n, 3, device=dev)
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0)
        ((rgb - tgt.unsqueeze(1)) ** 2).mean().backward()
    torch.cuda.synchronize()
    raw = model._scratch_buf.view(torch.int64)[-1024 * 8:].cpu().view(1024, 8)
    mp = (n + 3) // 4 * 4 * ((S - 1 + 15) // 16) * 16
    splits = max(1, min(128, mp // 32 // 24))                 # nerf_backward.hip: choose_splits
    wgs = [r.tolist() for r in raw[:6 * splits]]
    t0 = min(w[0] for w in wgs)
    us = lambda ticks: ticks / 100.0                      # s_memrealtime: 100 MHz
    print(f"{len(wgs)} workgroups = 6 jobs x {splits} splits; kernel span {us(max(w[5] for w in wgs) - t0):.1f} us")
    kinds = {"hidden": range(0, 4 * splits), "layer 0": range(4 * splits, 5 * splits), "layer 5": range(5 * splits, 6 * splits)}
    for kind, idx in kinds.items():
        sel = [wgs[i] for i in idx]
        med = lambda f: statistics.median(f(w) for w in sel)
        print(f"  {kind:8s}: start {med(lambda w: us(w[0] - t0)):7.1f} us | batch max {med(lambda w: us(w[1] - w[0])):5.1f} | ring fill "
              f"{med(lambda w: us(w[2] - w[1])):5.1f} | k-steps {med(lambda w: us(w[3] - w[2])):6.1f} (min {min(us(w[3] - w[2]) for w in sel):6.1f}, max "
              f"{max(us(w[3] - w[2]) for w in sel):6.1f}) | slab stores issued {med(lambda w: us(w[4] - w[3])):5.1f} | retired "
              f"{med(lambda w: us(w[5] - w[4])):5.1f} | whole {med(lambda w: us(w[5] - w[0])):6.1f} us")
    # per CU: the order of its workgroups and the gaps between them
    cus = {}
    for i, w in enumerate(wgs):
        hw, xcc = w[7] & 0xffffffff, (w[7] >> 32) & 0xf
        cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)          # XCD, SE, SH, CU
        cus.setdefault(cu, []).append((w[0], w[5], i))
    counts = {}
    gaps = []
    for cu, lst in cus.items():
        lst.sort()
        counts[len(lst)] = counts.get(len(lst), 0) + 1
        gaps += [us(b[0] - a[1]) for a, b in zip(lst, lst[1:])]
    print(f"  {len(cus)} CUs; workgroups per CU: {sorted(counts.items())}; gap between a workgroup's exit and the next one's entry on "
          f"its CU: median {statistics.median(gaps):.1f} us, max {max(gaps):.1f} us")
    ends = sorted(us(max(e for _, e, _ in lst) - t0) for lst in cus.values())
    print(f"  last exit per CU: min {ends[0]:.1f}, median {ends[len(ends) // 2]:.1f}, max {ends[-1]:.1f} us")
    inner = [int(v) for v in raw.reshape(-1)[6144:6144 + 2040].tolist() if int(v) != 0]
    per = 2 + 4 + 3                                      # DMA issue (2), slots a = 0..3, wait, barrier, after
    steps = [inner[k * per:(k + 1) * per + 1] for k in range(8, min(120, len(inner) // per - 1))]
    if steps:
        names = ["dY half of the DMA issue (4 instructions)", "raw LDS reads of slot 0 issued", "slot 0 (12 MFMAs; + the X half of the DMA issue at its end)", "slot 1",
                 "slot 2", "slot 3", "hand-over wait (vmcnt / lgkmcnt)", "barrier", "barrier -> DMA issue of the next step (addresses)"]
        print(f"  inside the k-steps of workgroup 0, wave 0 ({len(steps)} steady-state steps; s_memtime ticks):")
        for k, nm in enumerate(names):
            v = [st[k + 1] - st[k] for st in steps]
            print(f"    {nm:60s} median {statistics.median(v):7.0f}  min {min(v):6.0f}  max {max(v):6.0f}")
        v = [st[-1] - st[0] for st in steps]
        print(f"    {'whole step':36s} median {statistics.median(v):7.0f}  min {min(v):6.0f}  max {max(v):6.0f}")
    order = sorted(cus.items(), key=lambda kv: kv[1][0][0])[:3]
    for cu, lst in order:
        print("   CU", cu, " ".join(f"[wg {i} ({'hidden' if i < 4 * splits else 'L0' if i < 5 * splits else 'L5'}) {us(s - t0):.1f}..{us(e - t0):.1f}]" for s, e, i in lst))


if __name__ == "__main__":
    main()
