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
        self._pending = None

    def hyper(self):
        g = self.param_groups[0]
        return g["lr"], g["weight_decay"], tuple(g["betas"]), g["eps"]

    def zero_grad(self):
        """gradients are zeroed inside the fused step (hipMemsetAsync of the gradient arena)"""

    def step(self):
        raise RuntimeError("FusedAdam.step() runs inside geomapnet_amd.train.step_feedfwd (one fused HIP call per "
                           "training step); call step_feedfwd(..., train=True)")

    # -- torch.optim.Adam's state_dict format (checkpoints interchange with the reference's
    #    `optimizer.learner.state_dict()`, common/train.py:202 / :170): per-parameter `exp_avg` / `exp_avg_sq`
    #    in the parameter's own (OIHW) shape, indexed in param_groups order
    def _moment_views(self, p):
        eng = self._engine
        n, off = eng.n_params, p.storage_offset()
        if p.untyped_storage().data_ptr() != eng.params.untyped_storage().data_ptr():
            raise RuntimeError("parameter is not a view of the engine's parameter arena")
        return (eng.opt_state.as_strided(p.size(), p.stride(), n + off),
                eng.opt_state.as_strided(p.size(), p.stride(), 2 * n + off))

    def state_dict(self):
        state, groups, idx = {}, [], 0
        eng = self._engine
        have = eng is not None and eng.opt_state is not None and eng.step_count > 0
        # a state that was loaded but has not reached the device yet (the optimiser meets its engine in the first
        # step_feedfwd) is returned as loaded: save -> load -> save without a step in between keeps the moments
        pend = getattr(self, "_pending", None)
        step = eng.effective_step() if have else 0  # Adam's own count: fp16 steps skipped on overflow are not steps
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                if pend:
                    st = pend.get(idx, pend.get(str(idx)))
                    if st is not None:
                        state[idx] = {"step": int(st["step"]), "exp_avg": st["exp_avg"].detach().clone().cpu(),
                                      "exp_avg_sq": st["exp_avg_sq"].detach().clone().cpu()}
                elif have:
                    m, v = self._moment_views(p)
                    state[idx] = {"step": int(step), "exp_avg": m.clone().contiguous(),
                                  "exp_avg_sq": v.clone().contiguous()}
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
        for g in self.param_groups:
            for p in g["params"]:
                st = pend.get(idx, pend.get(str(idx)))
                if st is not None:
                    m, v = self._moment_views(p)
                    m.copy_(st["exp_avg"].to(m.device).reshape(m.shape))
                    v.copy_(st["exp_avg_sq"].to(v.device).reshape(v.shape))
                    step = max(step, int(st["step"]))
                idx += 1
        eng.step_count = step
        self._pending = None

    def _attach(self, engine):
        self._engine = engine
        self._apply_pending()


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
