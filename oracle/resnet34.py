"""Oracle: torchvision-compatible ResNet-34 (TEST INFRASTRUCTURE).

The reference builds its feature extractor with `torchvision.models.resnet34`
(scripts/train.py:76, scripts/eval.py:71); torchvision is a third-party dependency that is
not vendored under /root/reference and is not installed here, so the published architecture
is restated: conv7x7/2 -> BN -> ReLU -> maxpool3x3/2 -> BasicBlock x [3,4,6,3] at widths
(64,128,256,512), stride 2 on the first 3x3 conv of the first block of layers 2-4 with a
1x1/2 conv+BN projection shortcut, global average pool, fc.  Attribute names reproduce the
torchvision state-dict keys listed in SURVEY.md Appendix B so checkpoints interchange.
"""
import torch
from torch import nn
import torch.nn.functional as F

_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


class _Block(nn.Module):
    """torchvision BasicBlock: two 3x3 convs, identity/projection shortcut, post-add ReLU."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        sc = x if self.downsample is None else self.downsample(x)
        return F.relu(y + sc)


class ResNet34(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (width, nblk, stride) in enumerate(_STAGES, start=1):
            blocks = []
            for b in range(nblk):
                blocks.append(_Block(cin, width, stride if b == 0 else 1))
                cin = width
            setattr(self, "layer%d" % li, nn.Sequential(*blocks))
        self.avgpool = nn.AvgPool2d(7, stride=1)  # PoseNet swaps this for AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, num_classes)
        # torchvision's own init: He (fan_out) for convs, unit/zero for BN
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.fc(x.view(x.size(0), -1))


def resnet34(pretrained=False):
    if pretrained:
        raise RuntimeError("oracle resnet34: model-zoo weights are not available offline")
    return ResNet34()
