from .blocks import BasicBlock, Bottleneck, HBlock, PreBasicBlock, PreBottleneck, conv1x1, conv3x3
from .resnet import DaBNNStem, ResNet, resnet18, resnet34, resnet50

__all__ = ["BasicBlock", "Bottleneck", "HBlock", "PreBasicBlock", "PreBottleneck", "conv1x1",
           "conv3x3", "DaBNNStem", "ResNet", "resnet18", "resnet34", "resnet50"]
