"""Mirror of reference ``nr4seg/network/deeplabv3.py``.

The reference wraps ``torchvision.models.segmentation.deeplabv3_resnet101``
(torchvision 0.12, ``aux_loss=None``).  torchvision is not a dependency here:
the same architecture is defined below with **torchvision-compatible module
names**, so reference checkpoints (keys ``_model.backbone.*``,
``_model.classifier.*``; rewrite at scripts/train_joint.py:116-128) load with
``strict=True``.

Convolutions run through PyTorch-ROCm (MIOpen / rocBLAS) -- dense,
library-shaped work.  The hand-written HIP parts of the
segmentation path are the memory-bound passes around them: every
BatchNorm2d (+ residual add) (+ ReLU), forward and backward, is one fused
channels-last op (``FusedBatchNorm2d`` -> ``ucsa_bn_act_fwd/bwd``,
csrc/batchnorm.hip) instead of MIOpen's batch norm plus separate add / ReLU
kernels, and the tail (softmax / argmax / CE-on-softmax fwd+bwd,
``ucsa_seg_tail``).  On inputs the fused op does not take (CPU, NCHW) the same
modules run ``F.batch_norm`` + add + relu.  ``cfg_model["backbone"]`` ("resnet101" default,
"resnet50" for BASELINE cfg3's ResNet-50 wording) is an optional extra key.
Pretrained weights cannot be downloaded here (no network): the flags are
accepted and ignored with a warning unless ``cfg_model["weights_path"]`` points
to a local state_dict.
"""
from __future__ import annotations

import warnings
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fused_bn import FusedBatchNorm2d

__all__ = ["DeepLabV3"]


class PointwiseConv2d(nn.Conv2d):
    """1x1 convolution (same parameters and state_dict keys as nn.Conv2d).
    On a channels_last input it is one GEMM over all pixels of the batch
    (``F.linear`` on the NHWC view) instead of MIOpen's one GEMM launch per
    image: measured 33.9 -> 30.4 ms for the bf16 channels_last train step of
    DeepLabV3-R101 on 8 x 240x320 (tools/seg_pointwise_exp.py).  Any other
    layout, a stride or a 1x1 map goes through the ordinary convolution."""

    def forward(self, x):
        if (self.stride == (1, 1) and x.dim() == 4 and x.shape[1] > 1
                and x.shape[2] * x.shape[3] > 1
                and x.is_contiguous(memory_format=torch.channels_last)
                and not x.is_contiguous()):
            w = self.weight.view(self.out_channels, self.in_channels)
            y = F.linear(x.permute(0, 2, 3, 1), w, self.bias)
            return y.permute(0, 3, 1, 2)
        return super().forward(x)


def _conv1x1(cin, cout, stride=1, bias=False):
    return PointwiseConv2d(cin, cout, 1, stride=stride, bias=bias)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv1x1(inplanes, planes)
        self.bn1 = FusedBatchNorm2d(planes)
        # torchvision "v1.5": the stride sits on the 3x3 convolution
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride,
                               padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FusedBatchNorm2d(planes)
        self.conv3 = _conv1x1(planes, planes * 4)
        self.bn3 = FusedBatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        if self.downsample is not None:
            identity = self.downsample(x)
        # conv3 -> BN -> (+ identity) -> ReLU as one pass over the activation
        return self.bn3(self.conv3(out), residual=identity, relu=True)


class ResNetBackbone(nn.Module):
    """ResNet-50/101 trunk with replace_stride_with_dilation=[False, True,
    True] (output stride 8), truncated after layer4 -- what torchvision's
    IntermediateLayerGetter(return_layers={"layer4": "out"}) keeps."""

    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.dilation = 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FusedBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2, dilate=True)
        self.layer4 = self._make_layer(512, layers[3], stride=2, dilate=True)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out",
                                        nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                _conv1x1(self.inplanes, planes * 4, stride=stride),
                FusedBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample,
                             previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes,
                                     dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(x), relu=True))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return OrderedDict(out=x)


class _ConvBNReLU(nn.Sequential):
    """Sequential(conv, BatchNorm2d, ReLU[, ...]) (torchvision's module order
    and state_dict keys) whose forward fuses the BatchNorm with the ReLU;
    modules after index 2 (the projection's Dropout) run as usual."""

    def forward(self, x):
        x = self[1](self[0](x), relu=True)
        for m in list(self)[3:]:
            x = m(x)
        return x


class ASPPConv(_ConvBNReLU):

    def __init__(self, cin, cout, dilation):
        super().__init__(
            nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation,
                      bias=False), FusedBatchNorm2d(cout), nn.ReLU())


class ASPPPooling(nn.Sequential):

    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1),
                         nn.Conv2d(cin, cout, 1, bias=False),
                         FusedBatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        """torchvision's forward (pool -> 1x1 conv -> BN -> ReLU -> bilinear
        upsample), plus one zero-copy re-striding of the pooled map.

        Diagnosed on MI355X / ROCm 7.2 (tests/scripts/aspp_probe.py,
        aspp_probe2.py; each case in its own process): MIOpen's *NHWC,
        16-bit* batch-norm TRAINING kernel segfaults on maps of one or two
        pixels with small batches ([2,256,1,1], [3,256,1,1], [2,256,1,2] in
        bf16 or fp16 crash; [4,256,1,1], [2,256,2,2], fp32, eval mode, the
        NCHW kernel and the native (non-MIOpen) kernel are all fine).  A
        [B,C,1,1] tensor is NCHW- and NHWC-contiguous at once and PyTorch picks
        the kernel from its strides: a channels_last model hands BatchNorm the
        conv output with strides (C,1,C,C) -> the NHWC kernel -> the crash at
        batch 2 under bf16 autocast.  flatten/unflatten gives the same storage
        the canonical strides (C,1,1,1), so the NCHW kernel runs, in the
        input's own dtype -- no fp32 detour, no autocast switch."""
        size = x.shape[-2:]
        cl = (x.dim() == 4 and not x.is_contiguous()
              and x.is_contiguous(memory_format=torch.channels_last))
        y = self[1](self[0](x))                       # [B, cout, 1, 1]
        y = y.flatten(1).unflatten(1, (y.shape[1], 1, 1))
        y = self[3](self[2](y))
        y = F.interpolate(y, size=size, mode="bilinear", align_corners=False)
        return y.contiguous(memory_format=torch.channels_last) if cl else y


class ASPP(nn.Module):

    def __init__(self, cin, rates, cout=256):
        super().__init__()
        mods = [_ConvBNReLU(_conv1x1(cin, cout),
                            FusedBatchNorm2d(cout), nn.ReLU())]
        mods += [ASPPConv(cin, cout, r) for r in rates]
        mods.append(ASPPPooling(cin, cout))
        self.convs = nn.ModuleList(mods)
        self.project = _ConvBNReLU(
            _conv1x1(len(self.convs) * cout, cout),
            FusedBatchNorm2d(cout), nn.ReLU(), nn.Dropout(0.5))

    def forward(self, x):
        return self.project(torch.cat([c(x) for c in self.convs], dim=1))


class DeepLabHead(nn.Sequential):

    def __init__(self, cin, num_classes):
        super().__init__(ASPP(cin, [12, 24, 36]),
                         nn.Conv2d(256, 256, 3, padding=1, bias=False),
                         FusedBatchNorm2d(256), nn.ReLU(),
                         _conv1x1(256, num_classes, bias=True))

    def forward(self, x):
        x = self[0](x)
        x = self[2](self[1](x), relu=True)
        return self[4](x)


class _DeepLabV3Model(nn.Module):
    """torchvision.models.segmentation.DeepLabV3 with aux_classifier=None."""

    def __init__(self, backbone, classifier):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier

    def forward(self, x):
        size = x.shape[-2:]
        feats = self.backbone(x)
        out = self.classifier(feats["out"])
        # the logits leave in the reference's NCHW layout: re-striding the small
        # pre-upsample map (1/64 of the output) is free, transposing the
        # upsampled [B,C,H,W] afterwards (what `.contiguous()` on a
        # channels-last result does, forward and backward) is 2 x 98 MB
        out = out.contiguous()
        out = F.interpolate(out, size=size, mode="bilinear",
                            align_corners=False)
        return OrderedDict(out=out)


_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}


class DeepLabV3(nn.Module):
    """reference nr4seg/network/deeplabv3.py:6-19: ``DeepLabV3(cfg_model)``,
    ``forward(img) -> {"out": logits [B,num_classes,H,W]}``."""

    def __init__(self, cfg_model):
        super().__init__()
        from ._gemm_tuning import ensure as _tuned_gemms
        _tuned_gemms()   # look-up-only TunableOp table for the 1x1 convolutions
        name = cfg_model.get("backbone", "resnet101")
        self._model = _DeepLabV3Model(ResNetBackbone(_LAYERS[name]),
                                      DeepLabHead(2048, cfg_model["num_classes"]))
        path = cfg_model.get("weights_path")
        if path:
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd)
            # strict: a key mismatch must not pass silently
            self.load_state_dict(sd, strict=True)
        elif cfg_model.get("pretrained") or cfg_model.get("pretrained_backbone"):
            warnings.warn(
                "pretrained / pretrained_backbone requested but no network "
                "access and no cfg_model['weights_path']: random initialisation")

    def forward(self, data):
        from . import _gemm_tuning
        _gemm_tuning.use(not torch.is_autocast_enabled())
        return self._model(data)
