"""ResNet-18 blur-type estimator used to route images through the detector ensemble at evaluation time
(reference evaluate.py:186-205 builds `torchvision.models.resnet18()` and replaces `fc` with a 4- or
16-way layer; only its INFERENCE is on the built path, engine.py:359-366).  Stock PyTorch modules."""
import torch.nn.functional as F
from torch import nn


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(out)) + idt)


class ResNet18(nn.Module):
    graph_safe = True      # forward is a fixed sequence of launches: engine.evaluate may replay it as a HIP graph

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cfg, inpl, layers = [(64, 1), (128, 2), (256, 2), (512, 2)], 64, []
        for planes, stride in cfg:
            layers.append(nn.Sequential(BasicBlock(inpl, planes, stride), BasicBlock(planes, planes)))
            inpl = planes
        self.layer1, self.layer2, self.layer3, self.layer4 = layers
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(F.adaptive_avg_pool2d(x, 1).flatten(1))


def resnet18(num_classes=1000):
    return ResNet18(num_classes)
