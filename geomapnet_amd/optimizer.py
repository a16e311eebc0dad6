"""Optimizer facade: /root/reference/common/optimizer.py:8-47 (`Optimizer(params, method, base_lr,
weight_decay, **kwargs)` with `.learner`, `.adjust_lr(epoch)`, `.mult_lr(f)`).

Every shipped config uses Adam (scripts/configs/*.ini `opt = adam`); that is the method the fused
HIP step implements (geomapnet_amd/csrc/optim.h: clip + L2 weight decay + Adam in one kernel over
the flat parameter arena).  `learner` is a light object carrying param_groups/hyper-parameters in
torch.optim's shape; the update itself runs inside train.step_feedfwd's fused call.
"""


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            self.param_groups.append(g)
        hp = [(g["lr"], g["weight_decay"], tuple(g["betas"]), g["eps"]) for g in self.param_groups]
        if len(set(hp)) > 1:
            raise NotImplementedError("the fused step applies one (lr, wd, betas, eps) to all groups, as the "
                                      "reference's scripts do (scripts/train.py:104-112)")
        self._engine = None

    def hyper(self):
        g = self.param_groups[0]
        return g["lr"], g["weight_decay"], tuple(g["betas"]), g["eps"]

    def zero_grad(self):
        """gradients are zeroed inside the fused step (hipMemsetAsync of the gradient arena)"""

    def step(self):
        raise RuntimeError("FusedAdam.step() runs inside geomapnet_amd.train.step_feedfwd (one fused HIP call per "
                           "training step); call step_feedfwd(..., train=True)")

    def state_dict(self):
        sd = {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}
        if self._engine is not None and self._engine.opt_state is not None:
            n = self._engine.n_params
            sd["step"] = self._engine.step_count
            sd["exp_avg"] = self._engine.opt_state[n:2 * n].clone()
            sd["exp_avg_sq"] = self._engine.opt_state[2 * n:3 * n].clone()
        return sd

    def load_state_dict(self, sd):
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in s.items() if k != "params"})
        self._pending = {k: sd[k] for k in ("step", "exp_avg", "exp_avg_sq") if k in sd}

    def _attach(self, engine):
        self._engine = engine
        pend = getattr(self, "_pending", None)
        if pend and engine.opt_state is not None:
            n = engine.n_params
            engine.opt_state[n:2 * n].copy_(pend["exp_avg"])
            engine.opt_state[2 * n:3 * n].copy_(pend["exp_avg_sq"])
            engine.step_count = int(pend["step"])
            self._pending = None


class Optimizer:
    def __init__(self, params, method, base_lr, weight_decay, **kwargs):
        self.method = method
        self.base_lr = base_lr
        if method != "adam":
            raise NotImplementedError("the HIP training step implements method='adam' (every shipped config); got %r" % method)
        self.learner = FusedAdam(params, lr=base_lr, weight_decay=weight_decay, **kwargs)

    def adjust_lr(self, epoch):
        return self.base_lr  # step-LR applies to SGD only in the reference (common/optimizer.py:29-30)

    def mult_lr(self, f):
        for g in self.learner.param_groups:
            g["lr"] *= f
