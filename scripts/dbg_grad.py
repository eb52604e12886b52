import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import golden_params, load_golden
from oracle import nerf_oracle as O
from nerf_amd import NeRF
dev = torch.device('cuda:0')
for name, scale in (("g6_train_step", 1.0), ("g6_train_step_x3", 3.0)):
    g = load_golden(name)
    model = NeRF(); model.load_state_dict(golden_params(scale)); model = model.to(dev)
    pixels, _ = model.render_rays(g["rays_o"].to(dev), g["rays_d"].to(dev), 64, randomly_sample=True,
                                  density_noise_std=1.0, u=g["u"].to(dev), noise=g["noise"].to(dev))
    loss = ((pixels - g["target"].to(dev).unsqueeze(1)) ** 2).mean()
    loss.backward()
    # fp64 oracle
    p64 = {k: v.double().requires_grad_(k.startswith("prediction")) for k, v in golden_params(scale).items()}
    l64 = O.training_loss(p64, O.default_config(), g["rays_o"].double(), g["rays_d"].double(), 64, g["target"].double(), g["u"].double(), g["noise"].double(), 1.0)
    l64.backward()
    print(name, float(loss), float(g["loss"]), float(l64))
    for k, p in model.named_parameters():
        ref = g["grad." + k]; r64 = p64[k].grad.float()
        mx = ref.abs().max()
        print(f"  {k:32s} hip-ref {float((p.grad.cpu()-ref).abs().max()/mx):.2e}  hip-f64 {float((p.grad.cpu()-r64).abs().max()/mx):.2e}  ref-f64 {float((ref-r64).abs().max()/mx):.2e}")
