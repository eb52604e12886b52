"""Diagnostic (GPU box): phase time stamps of the fp32 data gradient for workgroup 0 and the workgroup that
shares its CU (-DNERF_EXP_STAMPS build): per wave, item start / LayerNorm-backward start / loop start / item end.
usage: python scripts/stamps_dgrad.py"""
import os, sys, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "NERF_HIP_LIB" not in os.environ:
    from nerf_amd import build as B
    out = os.path.join(ROOT, "nerf_amd", "csrc", "libnerf_hip_stamps.so")
    B.build(out=out, defines=["NERF_EXP_STAMPS"] + sys.argv[1:])
    sys.exit(subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, NERF_HIP_LIB=out)).returncode)
from nerf_amd import NeRF
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = NeRF().to(dev)
n, S = 4096, 64
o, d, tgt = torch.randn(n, 3, device=dev), torch.randn(n, 3, device=dev), torch.rand(n, 3, device=dev)
for _ in range(3):
    model.zero_grad(set_to_none=True)
    rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0)
    ((rgb - tgt.unsqueeze(1)) ** 2).mean().backward()
torch.cuda.synchronize()
raw = model._scratch_buf.view(torch.int64)[-8 * 64:].cpu().view(8, 64)
# per item: start, then (LN start, loop start) x 4, LN start (layer 0), end = 11 stamps
t0 = None
rows = []
for w in range(8):
    vals = [int(v) for v in raw[w]]
    hw, ts = vals[0] & 0xffffffff, [v for v in vals[1:] if v != 0]
    rows.append((w, hw, ts))
    if ts:
        t0 = min(ts) if t0 is None else min(t0, min(ts))
names = ["item"] + ["LN", "loop"] * 4 + ["LN", "end"]
for w, hw, ts in rows:
    if w % 4:
        continue                                 # the four waves of a workgroup move together: show wave 0
    print(f"WG{w // 4} wave{w % 4} hw_id {hw:#x} (cu {(hw >> 8) & 0xf}, simd {(hw >> 4) & 3}, se {(hw >> 13) & 7}):")
    out = []
    for i, t in enumerate(ts[:33]):
        out.append(f"{names[i % 11]}@{(t - t0) / 100:.0f}")
    print("   " + " ".join(out))
print("(time in units of 100 s_memtime ticks = 1 us at 100 MHz; per item: item, 4 x (LN, loop), LN, end)")
