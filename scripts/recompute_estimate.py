"""What would "recompute instead of save" buy the training step?  (GPU box)

The training step saves the normalised activations x_hat of five layers in the forward (1.34 GB written per
4096 x 64 batch) and reads them back twice (data gradient: LayerNorm backward; weight gradient: the GEMM's X
operand).  The alternative recomputes them from the encoded inputs wherever they are needed.  This script
measures the PRICE of one recomputation and the SAVING of not writing, on the training batch itself:

  forward without saves  = the inference kernel on the same 4096 x 64 rays (same MLP loops, no x_hat / rstd /
                           output stores, compositing fused in)
  forward with saves     = the training forward kernel (+ its compositing kernel)

and prints them next to the backward kernels' times from the same run, so that
  recompute-in-the-data-gradient  =  data gradient - (its x_hat reads) + forward without saves
can be read off against the measured cost of the x_hat reads (the data gradient without its
x_hat reads runs 0.165 ms faster at f16 pairs).   python scripts/recompute_estimate.py [rays] [samples]
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF, _lib

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
model = NeRF().to(dev)
o, d = torch.randn(n, 3, device=dev), torch.randn(n, 3, device=dev)
u = torch.rand(n, S, device=dev)


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for precision in ("fp32", "f16x3"):
    model.precision = model.train_precision = precision

    def inference():
        with torch.no_grad():
            model.render_rays(o, d, S, randomly_sample=True, u=u)

    def train_forward():
        model.render_rays(o, d, S, randomly_sample=True, u=u)

    def train_step():
        model.zero_grad(set_to_none=True)
        rgb, _ = model.render_rays(o, d, S, randomly_sample=True, u=u)
        (rgb ** 2).mean().backward()

    a, b, c = timed(inference), timed(train_forward), timed(train_step)
    print(f"[{precision}] {n} rays x {S}: forward without saves {a:.3f} ms, forward with saves {b:.3f} ms "
          f"(saves cost {b - a:+.3f}), forward + backward {c:.3f} ms -> backward {c - b:.3f} ms; one extra "
          f"recomputation costs {a:.3f} ms against the {b - a:.3f} ms the saves cost", flush=True)
