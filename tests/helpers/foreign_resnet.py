"""A ResNet-18 written against the reference's module LAYOUT (attribute names of bnn/models/resnet.py:93-101 and
bnn/models/layers/res_block.py:30-37) in a package that is not bnn_amd: what a user of the reference's `bnn.models`
brings to `prepare_binary_model`.  `residual=False` keeps the names and changes the arithmetic (no shortcut add) — the
case AutoFusion's first-call check has to catch."""
import torch
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1
    residual = True

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.act2 = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.bn2(self.conv2(self.act1(self.bn1(self.conv1(x)))))
        if self.residual:
            out = out + (x if self.downsample is None else self.downsample(x))
        return self.act2(out)


class ResNet(nn.Module):
    def __init__(self, residual: bool = True):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        stages, inp = [], 64
        for planes, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
            ds = None
            if stride != 1 or inp != planes:
                ds = nn.Sequential(nn.AvgPool2d(stride, stride, ceil_mode=True, count_include_pad=False),
                                   nn.Conv2d(inp, planes, 1, bias=False), nn.BatchNorm2d(planes))
            stages.append(nn.Sequential(BasicBlock(inp, planes, stride, ds), BasicBlock(planes, planes)))
            inp = planes
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
        for m in self.modules():
            if isinstance(m, BasicBlock):
                m.residual = residual

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))
