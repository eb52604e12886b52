"""``Adam`` — torch.optim.Adam's update as one HIP launch (include/nerf_hip.h: nerf_hip_adam_step).

The reference's training loops build ``optim.Adam(nerf.parameters(), lr=...)`` with the default betas / eps, no
weight decay and no amsgrad (train_conditional_nerf.py:106-107, examples/example.ipynb cell 7).  This class is that
optimiser for parameters on a ROCm device: same constructor arguments (the unsupported options raise), same
update rule, ``torch.optim.Optimizer`` protocol (``zero_grad``, ``param_groups``, ``state_dict`` with the flat
moments), and graph-capturable by construction: the step count lives on the device, the kernel itself increments
it, and every step is ONE kernel launch.  (``lr``, ``betas`` and ``eps`` are launch arguments: a captured
step replays with the values it was captured with — the reference's loops keep them constant; re-capture after
changing them.)  torch's fused multi-tensor kernel needs 43 us for this model's 22 small tensors; at 512
rays per GPU (BASELINE config 5 on 8 GPUs) that is a tenth of the training step.
"""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("nerf_amd.optim.Adam: weight_decay / amsgrad are not implemented (the "
                                      "reference's scripts use neither)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise NotImplementedError("nerf_amd.optim.Adam takes one parameter group")
        self._params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not 1 <= len(self._params) <= _lib.ADAM_MAX_TENSORS:
            raise ValueError(f"nerf_amd.optim.Adam takes 1 .. {_lib.ADAM_MAX_TENSORS} parameter tensors")
        self._flat = None
        self._steps = None

    def _state(self):
        if self._flat is None:
            dev = self._params[0].device
            for p in self._params:
                if not p.is_cuda or p.dtype != torch.float32 or p.device != dev or not p.is_contiguous():
                    raise RuntimeError("nerf_amd.optim.Adam: parameters must be contiguous float32 tensors on one "
                                       "ROCm device")
            total = sum(p.numel() for p in self._params)
            # the step count: one copy per workgroup of the launch (include/nerf_hip.h), all equal; state shows [0]
            self._steps = torch.zeros(_lib.ADAM_STEP_SLOTS, dtype=torch.float32, device=dev)
            self._flat = dict(exp_avg=torch.zeros(total, dtype=torch.float32, device=dev),
                              exp_avg_sq=torch.zeros(total, dtype=torch.float32, device=dev),
                              step=self._steps[:1])
            self.state["flat"] = self._flat          # (visible through state_dict())
        return self._flat

    def load_state_dict(self, state_dict):
        """torch's loader replaces ``state["flat"]`` by copies; the kernel's buffers take their VALUES and keep
        their addresses: a HIP graph captured before the load (``Trainer(graph=True)``) has those addresses in
        its Adam launch and goes on replaying against them, so the loaded moments and step count must land
        there, not in fresh tensors."""
        super().load_state_dict(state_dict)
        loaded = self.state.get("flat")
        if loaded is not None:
            st = self._state()                       # existing buffers if there are any, else new ones
            st["exp_avg"].copy_(loaded["exp_avg"].reshape(-1))
            st["exp_avg_sq"].copy_(loaded["exp_avg_sq"].reshape(-1))
            self._steps.fill_(float(loaded["step"].reshape(-1)[0]))
            self.state["flat"] = st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = self._state()
        group = self.param_groups[0]
        args = _lib.AdamArgs()
        args.num_tensors, off = len(self._params), 0
        keep = []
        for i, p in enumerate(self._params):
            g = p.grad
            if g is None:
                raise RuntimeError("nerf_amd.optim.Adam.step(): a parameter has no gradient")
            g = g if g.is_contiguous() else g.contiguous()
            keep.append(g)
            args.offsets[i] = off
            args.params[i], args.grads[i] = p.data_ptr(), g.data_ptr()
            off += p.numel()
        args.offsets[len(self._params)] = off
        args.total = off
        # (the step count is read AND incremented on the device, by the kernel itself)
        args.exp_avg, args.exp_avg_sq, args.step = _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), _lib.ptr(self._steps)
        args.lr, (args.beta1, args.beta2), args.eps = float(group["lr"]), group["betas"], float(group["eps"])
        dev = self._params[0].device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.lib().nerf_hip_adam_step(ctypes.byref(args), ctypes.c_void_p(stream)), "nerf_hip_adam_step")
        # The kernel wrote the parameters through raw pointers: tell autograd (its in-place-modification guard)
        # and everything keyed on the version counters (the models' f16 range-check cache) that they changed,
        # as torch's own optimisers do.  Host-side only, so legal during a capture; a REPLAY runs no host code
        # at all, which is why the trainer forgets the range check itself between replays.
        for p in self._params:
            torch.autograd.graph.increment_version(p)
        return loss
