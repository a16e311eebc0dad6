"""Oracle: PoseNet / MapNet heads (TEST INFRASTRUCTURE).

Follows /root/reference/models/posenet.py:
  * PoseNet.__init__  (:37-63)  -- avgpool -> AdaptiveAvgPool2d(1), fc -> Linear(512, feat_dim),
    two Linear(feat_dim, 3) heads, He-normal (fan_in) init of the new linears when
    `pretrained`, of every Conv2d/Linear otherwise, zero biases; optional NaN filter hook.
  * PoseNet.forward   (:65-73)  -- relu(fe(x)) -> dropout -> cat(fc_xyz, fc_wpqr).
  * MapNet.forward    (:87-97)  -- fold the T frames of each window into the batch.
  * filter_hook       (:28-34)  -- zero NaNs in the gradients flowing out of fc_wpqr.

Dropout: the reference calls F.dropout(x, p) without `training=` (:68-69).  (`dropout_mask`, an attribute set by the
parity tests, replaces the random draw by a given mask.)  Under its pinned
PyTorch 0.4.1 that default is training=False, i.e. identity; under torch>=1.0 it is always
active.  `dropout_active=False` (default) reproduces the pinned behaviour; parity runs use
droprate=0 where both readings coincide.
"""
import torch
from torch import nn
import torch.nn.functional as F


def _zero_nans(t):
    return torch.where(t != t, torch.zeros_like(t), t)


class PoseNet(nn.Module):
    def __init__(self, feature_extractor, droprate=0.5, pretrained=True, feat_dim=2048, filter_nans=False,
                 dropout_active=False):
        super().__init__()
        self.droprate = droprate
        self.dropout_active = dropout_active
        self.filter_nans = filter_nans
        fe = feature_extractor
        fe.avgpool = nn.AdaptiveAvgPool2d(1)
        fe.fc = nn.Linear(fe.fc.in_features, feat_dim)
        self.feature_extractor = fe
        self.fc_xyz = nn.Linear(feat_dim, 3)
        self.fc_wpqr = nn.Linear(feat_dim, 3)
        targets = [fe.fc, self.fc_xyz, self.fc_wpqr] if pretrained else list(self.modules())
        for m in targets:
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    nn.init.constant_(m.bias.data, 0)

    def forward(self, x):
        feat = F.relu(self.feature_extractor(x))
        if getattr(self, "dropout_mask", None) is not None:
            # explicit mask (0 or 1/(1-p) per element): what F.dropout computes for ONE draw -- the parity tests hand the
            # oracle the mask the device drew (mn_debug_tensor "dropmask"), since two generators cannot agree on a draw
            feat = feat * self.dropout_mask
        elif self.droprate > 0 and self.dropout_active:
            feat = F.dropout(feat, p=self.droprate, training=True)
        if self.filter_nans:
            # reference: register_backward_hook(filter_hook) on fc_wpqr (:50-51).  The legacy
            # module hook filtered the module's grad_inputs (input, weight, bias grads of the
            # addmm).  Restated with tensor hooks on the same three quantities.
            h_in = feat + 0
            if h_in.requires_grad:
                h_in.register_hook(_zero_nans)
            wpqr = F.linear(h_in, self._hooked(self.fc_wpqr.weight), self._hooked(self.fc_wpqr.bias))
        else:
            wpqr = self.fc_wpqr(feat)
        xyz = self.fc_xyz(feat)
        return torch.cat((xyz, wpqr), 1)

    @staticmethod
    def _hooked(p):
        q = p + 0
        if q.requires_grad:
            q.register_hook(_zero_nans)
        return q


class MapNet(nn.Module):
    def __init__(self, mapnet):
        super().__init__()
        self.mapnet = mapnet

    def forward(self, x):
        n, t = x.shape[0], x.shape[1]
        return self.mapnet(x.reshape(n * t, *x.shape[2:])).view(n, t, -1)
