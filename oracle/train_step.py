"""Oracle: one training step (TEST INFRASTRUCTURE).

Restates /root/reference/common/train.py:322-363 (`step_feedfwd`: forward, criterion,
zero_grad, backward, optional clip_grad_norm over model.parameters() only, optimizer step,
loss.item()) and /root/reference/common/optimizer.py:8-47 (`Optimizer`: picks
torch.optim.{SGD,Adam,RMSprop}; step-LR only for SGD; every shipped config uses Adam).
CUDA transfer lines (:340-341,346-347) are dropped: the oracle is CPU-only.

`adam_reference_step` spells out Adam with L2 weight decay added to the gradient (not AdamW)
as the torch.optim.Adam installed here (2.10) evaluates it -- denom = sqrt(v)/sqrt(1-b2^t) + eps,
step = lr/(1-b1^t) -- and the clip_grad_norm rule (scale by max_norm / (total_norm + 1e-6)
when that coefficient is < 1), so the fused HIP Adam kernel can be checked against explicit
arithmetic as well as against torch.optim.Adam.  (PyTorch 0.4.1, the reference's pin, placed
eps before the bias correction: denom = sqrt(v) + eps, step = lr*sqrt(1-b2^t)/(1-b1^t); the
two differ only where sqrt(v_hat) is comparable to 1e-8..1e-6.  The CPU reference path the
north star names is the reference run on this host's torch, so the installed form is the
oracle; the HIP kernel exposes the 0.4.1 form as `eps_mode=1`.)
"""
import math

import torch
import torch.optim as optim


class Optimizer:
    def __init__(self, params, method, base_lr, weight_decay, **kwargs):
        self.method, self.base_lr = method, base_lr
        if method == "sgd":
            self.lr_decay = kwargs.pop("lr_decay")
            self.lr_stepvalues = sorted(kwargs.pop("lr_stepvalues"))
            self.learner = optim.SGD(params, lr=base_lr, weight_decay=weight_decay, **kwargs)
        elif method == "adam":
            self.learner = optim.Adam(params, lr=base_lr, weight_decay=weight_decay, **kwargs)
        elif method == "rmsprop":
            self.learner = optim.RMSprop(params, lr=base_lr, weight_decay=weight_decay, **kwargs)
        else:
            raise ValueError(method)

    def adjust_lr(self, epoch):
        if self.method != "sgd":
            return self.base_lr
        lr = self.base_lr * (self.lr_decay ** sum(1 for s in self.lr_stepvalues if epoch >= s))
        for g in self.learner.param_groups:
            g["lr"] = lr
        return lr

    def mult_lr(self, f):
        for g in self.learner.param_groups:
            g["lr"] *= f


def step_feedfwd(data, model, cuda=False, target=None, criterion=None, optim=None, train=True, max_grad_norm=0.0):
    if train:
        assert criterion is not None
    x = data.clone().requires_grad_(train)  # reference asks for the input gradient (:339)
    with torch.set_grad_enabled(train):
        output = model(x)
    if criterion is None:
        return 0, output
    with torch.set_grad_enabled(train):
        loss = criterion(output, target)
    if train:
        optim.learner.zero_grad()
        loss.backward()
        if max_grad_norm > 0.0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
        optim.learner.step()
    return loss.item(), output


def adam_reference_step(params, grads, exp_avg, exp_avg_sq, step, lr, weight_decay, beta1=0.9, beta2=0.999,
                        eps=1e-8, max_grad_norm=0.0, clip_mask=None):
    """Explicit Adam (torch 0.4.1 semantics) over lists of tensors, in place.  `step` is the
    1-based step count after this update.  `clip_mask[i]` says whether tensor i belongs to
    model.parameters() (clipped) or to the criterion (never clipped, common/train.py:357-358)."""
    if max_grad_norm > 0.0:
        sel = [g for i, g in enumerate(grads) if clip_mask is None or clip_mask[i]]
        total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in sel))
        coef = max_grad_norm / (total + 1e-6)
        if coef < 1:
            for g in sel:
                g.mul_(coef)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        if weight_decay != 0:
            g = g + weight_decay * p
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
