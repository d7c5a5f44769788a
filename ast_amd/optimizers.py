"""Optimizer drop-ins for nn.py:81-119: Adam(alpha, .9, .999, 1e-8, amsgrad=True) / SGD(lr) with the hooks
WeightDecay -> GradientClipping applied in insertion order, fused into two kernel launches
over the flat arena: astk_grad_sqnorm and astk_decay_clip_amsgrad_step."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import check


class WeightDecay:
    def __init__(self, rate):
        self.rate = rate


class GradientClipping:
    def __init__(self, threshold):
        self.threshold = threshold


class GradientNoise:
    """chainer.optimizer.GradientNoise(eta) (nn.py:108-110): g += N(0, sigma^2), sigma^2 = eta / (1 + t)^0.55 with t = the optimizer's
    update count BEFORE this update (Chainer's GradientMethod.update calls the 'pre' hooks, then increments t; exponential_decay_noise
    reads opt.t): sigma^2 = eta at the first update.  Applied behind WeightDecay and GradientClipping, the reference's insertion order."""

    def __init__(self, eta, seed=0x6E015E):
        self.eta = float(eta)
        self.seed = seed
        self.offset = 0


class _Optimizer:
    def __init__(self):
        self.target = None
        self.hooks = []
        self.t = 0
        self.sqnorm = None
        self.grad_sync = None      # set by ast_amd.dist for data-parallel runs: called before the hooks
        self.grad_scale = 1.0

    def setup(self, model):
        self.target = model
        return self

    def add_hook(self, hook):
        if isinstance(hook, WeightDecay) and any(isinstance(h, (GradientClipping, GradientNoise)) for h in self.hooks):
            raise NotImplementedError("the fused update implements the reference's order WeightDecay -> GradientClipping -> GradientNoise")
        if isinstance(hook, GradientClipping) and any(isinstance(h, GradientNoise) for h in self.hooks):
            raise NotImplementedError("the fused update implements the reference's order WeightDecay -> GradientClipping -> GradientNoise")
        self.hooks.append(hook)

    def _apply_noise(self, lib, a, l2, clip, s):
        """If a GradientNoise hook is installed: finish the gradient in place (decay, clip, noise) and return the (l2, clip, scale) the
        update kernel must use afterwards (none of them again); else pass the values through."""
        noise = [h for h in self.hooks if isinstance(h, GradientNoise)]
        if not noise:
            return l2, clip, self.grad_scale
        h = noise[0]
        sigma = math.sqrt(h.eta / (1.0 + (self.t - 1)) ** 0.55)      # self.t was incremented just above; the hook saw t - 1
        frozen = getattr(self.target, "_frozen", set())
        # parameter by parameter over the exact extents: the arena's alignment pads must keep their zero gradient (and value)
        for name, shp in a.shapes.items():
            if name.split("/")[0] in frozen:
                continue
            off, n = a.offsets[name], 1
            for d in shp:
                n *= int(d)
            check(lib.astk_decay_clip_noise(C.c_void_p(a.grad.data_ptr() + 4 * off), C.c_void_p(a.data.data_ptr() + 4 * off), n, self.grad_scale,
                                            l2, clip, C.c_void_p(self.sqnorm.data_ptr()), sigma, h.seed, h.offset, s))
            h.offset += (n + 1) // 2
        return 0.0, 3.0e38, 1.0

    def _hook_values(self):
        l2 = sum(h.rate for h in self.hooks if isinstance(h, WeightDecay))
        clips = [h.threshold for h in self.hooks if isinstance(h, GradientClipping)]
        clip = float(clips[0]) if clips else float("inf")
        return float(l2), clip

    def _prepare(self):
        m = self.target
        lib = _lib.load()
        a = m.arena
        if self.sqnorm is None:
            self.sqnorm = torch.zeros(1, dtype=torch.float64, device=a.device)
        # grad_sync may leave the SUM over the replicas in the arena and return the factor that turns it into the mean:
        # the kernels below apply it on the fly (one pass over 53 MB less per step than scaling in place)
        self.grad_scale = 1.0
        if self.grad_sync is not None:
            r = self.grad_sync(a)
            if isinstance(r, float):
                self.grad_scale = r
        l2, clip = self._hook_values()
        s = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        # the clip norm runs over every parameter (chainer's GradientClipping sums over target.params())
        check(lib.astk_grad_sqnorm_scaled(C.c_void_p(a.grad.data_ptr()), C.c_void_p(a.data.data_ptr()), self.grad_scale, l2, a.size,
                                          C.c_void_p(self.sqnorm.data_ptr()), s))
        if clip == float("inf"):
            clip = 3.0e38
        return lib, a, l2, clip, s

    @property
    def last_grad_norm(self):
        """sqrt(sum (g + l2 p)^2): the norm GradientClipping compares with its threshold (parity observable)."""
        return math.sqrt(float(self.sqnorm.item()))


class Adam(_Optimizer):
    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=False):
        super().__init__()
        self.alpha, self.beta1, self.beta2, self.eps, self.amsgrad = alpha, beta1, beta2, eps, amsgrad
        self.m = self.v = self.vhat = None

    def update(self):
        lib, a, l2, clip, s = self._prepare()
        if self.m is None:
            self.m = torch.zeros_like(a.data)
            self.v = torch.zeros_like(a.data)
            self.vhat = torch.zeros_like(a.data)
        self.t += 1
        lr_t = self.alpha * math.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        l2, clip, gscale = self._apply_noise(lib, a, l2, clip, s)
        for off, n in self.target.enabled_ranges():
            def p(t):
                return C.c_void_p(t.data_ptr() + 4 * off)
            check(lib.astk_decay_clip_amsgrad_step_scaled(p(a.data), p(a.grad), p(self.m), p(self.v), p(self.vhat), n, gscale, l2,
                                                          clip, C.c_void_p(self.sqnorm.data_ptr()), lr_t, self.beta1, self.beta2,
                                                          self.eps, 1 if self.amsgrad else 0, s))


class SGD(_Optimizer):
    def __init__(self, lr=0.01):
        super().__init__()
        self.lr = lr

    def update(self):
        lib, a, l2, clip, s = self._prepare()
        self.t += 1
        l2, clip, gscale = self._apply_noise(lib, a, l2, clip, s)
        for off, n in self.target.enabled_ranges():
            check(lib.astk_decay_clip_sgd_step_scaled(C.c_void_p(a.data.data_ptr() + 4 * off), C.c_void_p(a.grad.data_ptr() + 4 * off), n,
                                                      gscale, l2, clip, C.c_void_p(self.sqnorm.data_ptr()), self.lr, s))
