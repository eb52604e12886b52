import sys, torch, hashlib
sys.path.insert(0, '/root/repo')
from nerf_amd import trainer as T
dev = torch.device('cuda:0')
def digest(t): return hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]
images, poses, focal = T.synthetic_scene(num_views=9, size=24, num_samples=32, device=dev)
print("scene", digest(images))
for trial in range(3):
    run = T.Trainer(images, poses, focal, logging_dir=None, batch_size=512, learning_rate=5e-4,
                    num_samples_per_ray=32, density_noise_std=0.0, log_interval=10**9, seed=1)
    losses = []
    it = 0
    for ep in range(100):
      for batch in run.dataset.batches(512, generator=run.sampler):
        l = run.train_step(batch); losses.append(float(l)); it += 1
        if it in (1, 9, 100, 300, 600, 900):
            flat = torch.cat([p.grad.reshape(-1) for p in run.model.parameters()])
            print(f" trial {trial} it {it} loss {losses[-1]:.8f} batch {digest(batch['pixels'])} grad {digest(flat)} params {digest(torch.cat([p.reshape(-1) for p in run.model.parameters()]))}")
