"""Optimizer facade: /root/reference/common/optimizer.py:8-47 (`Optimizer(params, method, base_lr,
weight_decay, **kwargs)` with `.learner`, `.adjust_lr(epoch)`, `.mult_lr(f)`).

Every shipped config uses Adam (scripts/configs/*.ini `opt = adam`); the fused HIP step
(geomapnet_amd/csrc/optim.h: clip + L2 weight decay + update in one kernel over the flat parameter
arena) implements all three methods of the reference's wrapper: Adam, SGD (momentum / dampening /
Nesterov, with the wrapper's step schedule) and RMSprop (not centered).  `learner` is a light object carrying param_groups/hyper-parameters in
torch.optim's shape; the update itself runs inside train.step_feedfwd's fused call.
"""


class _FusedLearner:
    """torch.optim-shaped holder of param_groups / hyper-parameters / state for the fused HIP update.  Subclasses name
    the method (csrc/optim.h AdamArgs::method), the hyper-parameters the kernel reads as (beta1, beta2, eps), and the
    torch state keys of the two moment arenas."""
    METHOD = 0
    NAME = "Adam"

    def __init__(self, params, defaults):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.defaults = dict(defaults)
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            self.param_groups.append(g)
        def norm(v):  # a config file / JSON gives betas as a list, the default is a tuple: the same hyper-parameter
            return tuple(v) if isinstance(v, (list, tuple)) else v
        hp = [tuple(repr(norm(g[k])) for k in sorted(self.defaults)) for g in self.param_groups]
        if len(set(hp)) > 1:
            raise NotImplementedError("the fused step applies one set of hyper-parameters to all groups, as the "
                                      "reference's scripts do (scripts/train.py:104-112)")
        self._engine = None
        self._pending = None

    # (lr, weight_decay, (beta1, beta2), eps) as csrc/optim.h reads them, and (method, nesterov)
    def hyper(self):
        raise NotImplementedError

    def method(self):
        return self.METHOD, 0

    def _state_keys(self):
        """[(torch state key, arena index 1 | 2)] of the moments this configuration keeps"""
        raise NotImplementedError

    def _has_step(self):
        return True

    def zero_grad(self):
        """gradients are zeroed inside the fused step (own fill kernel over the gradient arena)"""

    def step(self):
        raise RuntimeError("Fused%s.step() runs inside geomapnet_amd.train.step_feedfwd (one fused HIP call per "
                           "training step); call step_feedfwd(..., train=True)" % self.NAME)

    # -- torch.optim's state_dict format (checkpoints interchange with the reference's
    #    `optimizer.learner.state_dict()`, common/train.py:202 / :170): per-parameter moments in the parameter's own
    #    (OIHW) shape, indexed in param_groups order
    def _moment_views(self, p):
        eng = self._engine
        n, off = eng.n_params, p.storage_offset()
        if p.untyped_storage().data_ptr() != eng.params.untyped_storage().data_ptr():
            raise RuntimeError("parameter is not a view of the engine's parameter arena")
        return {1: eng.opt_state.as_strided(p.size(), p.stride(), n + off),
                2: eng.opt_state.as_strided(p.size(), p.stride(), 2 * n + off)}

    def state_dict(self):
        state, groups, idx = {}, [], 0
        eng = self._engine
        # the optimiser's own count (torch keeps no state before the first APPLIED step: fp16 steps skipped on overflow are
        # not steps), read back from the device
        step = eng.effective_step() if eng is not None and eng.opt_state is not None else 0
        have = step > 0
        # a state that was loaded but has not reached the device yet (the optimiser meets its engine in the first
        # step_feedfwd) is returned as loaded: save -> load -> save without a step in between keeps the moments
        pend = getattr(self, "_pending", None)
        keys = self._state_keys()
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                if pend:
                    st = pend.get(idx, pend.get(str(idx)))
                    if st is not None:
                        state[idx] = {k: (int(v) if k == "step" else v.detach().clone().cpu()) for k, v in st.items()}
                elif have and (keys or self._has_step()):
                    views = self._moment_views(p)
                    entry = {"step": int(step)} if self._has_step() else {}
                    for key, arena in keys:
                        entry[key] = views[arena].clone().contiguous()
                    state[idx] = entry
                ids.append(idx)
                idx += 1
            packed = {k: v for k, v in g.items() if k != "params"}
            packed["params"] = ids
            groups.append(packed)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        groups = sd.get("param_groups", [])
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for g, s in zip(self.param_groups, groups):
            if len(s.get("params", g["params"])) != len(g["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of "
                                 "optimizer's group")
            g.update({k: v for k, v in s.items() if k != "params"})
        self._pending = dict(sd.get("state", {}))
        if self._engine is not None:
            self._apply_pending()

    def _apply_pending(self):
        pend = getattr(self, "_pending", None)
        if not pend:
            return
        eng = self._engine
        eng.ensure_opt_state()
        step, idx = 0, 0
        keys = self._state_keys()
        for g in self.param_groups:
            for p in g["params"]:
                st = pend.get(idx, pend.get(str(idx)))
                if st is not None:
                    views = self._moment_views(p)
                    for key, arena in keys:
                        if st.get(key) is not None:  # (torch's SGD stores momentum_buffer = None before a first gradient)
                            views[arena].copy_(st[key].to(views[arena].device).reshape(views[arena].shape))
                    # SGD keeps no step: a loaded momentum buffer means "not the first step"
                    step = max(step, int(st.get("step", 1 if st else 0)))
                idx += 1
        eng.set_step(step)
        self._pending = None

    def _attach(self, engine):
        self._engine = engine
        self._apply_pending()


class FusedAdam(_FusedLearner):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    def hyper(self):
        g = self.param_groups[0]
        return g["lr"], g["weight_decay"], tuple(g["betas"]), g["eps"]

    def _state_keys(self):
        return [("exp_avg", 1), ("exp_avg_sq", 2)]


class FusedSGD(_FusedLearner):
    """torch.optim.SGD(params, lr, momentum=0, dampening=0, weight_decay=0, nesterov=False): state = momentum_buffer"""
    METHOD = 1
    NAME = "SGD"

    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=bool(nesterov)))

    def hyper(self):
        g = self.param_groups[0]
        return g["lr"], g["weight_decay"], (g["momentum"], g["dampening"]), 0.0

    def method(self):
        return self.METHOD, int(bool(self.param_groups[0]["nesterov"]))

    def _state_keys(self):
        return [("momentum_buffer", 1)] if self.param_groups[0]["momentum"] != 0 else []

    def _has_step(self):
        return False


class FusedRMSprop(_FusedLearner):
    """torch.optim.RMSprop(params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False)"""
    METHOD = 2
    NAME = "RMSprop"

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0, centered=False):
        if centered:
            raise NotImplementedError("centered RMSprop keeps a third moment; the fused step has two moment arenas")
        super().__init__(params, dict(lr=lr, momentum=momentum, alpha=alpha, eps=eps, centered=False, weight_decay=weight_decay))

    def hyper(self):
        g = self.param_groups[0]
        return g["lr"], g["weight_decay"], (g["momentum"], g["alpha"]), g["eps"]

    def _state_keys(self):
        keys = [("square_avg", 2)]
        if self.param_groups[0]["momentum"] > 0:
            keys.append(("momentum_buffer", 1))
        return keys


class Optimizer:
    def __init__(self, params, method, base_lr, weight_decay, **kwargs):
        self.method = method
        self.base_lr = base_lr
        if method == "sgd":  # common/optimizer.py:16-20: step schedule + torch.optim.SGD
            self.lr_decay = kwargs.pop("lr_decay")
            self.lr_stepvalues = sorted(kwargs.pop("lr_stepvalues"))
            self.learner = FusedSGD(params, lr=base_lr, weight_decay=weight_decay, **kwargs)
        elif method == "adam":
            self.learner = FusedAdam(params, lr=base_lr, weight_decay=weight_decay, **kwargs)
        elif method == "rmsprop":
            self.learner = FusedRMSprop(params, lr=base_lr, weight_decay=weight_decay, **kwargs)
        else:
            raise ValueError("method must be 'sgd', 'adam' or 'rmsprop', got %r" % (method,))

    def adjust_lr(self, epoch):
        if self.method != "sgd":
            return self.base_lr  # the step schedule applies to SGD only (common/optimizer.py:29-30)
        lr = self.base_lr * self.lr_decay ** sum(1 for s in self.lr_stepvalues if epoch >= s)
        for g in self.learner.param_groups:
            g["lr"] = lr
        return lr

    def mult_lr(self, f):
        for g in self.learner.param_groups:
            g["lr"] *= f
