"""Feature extractor of the GNT static renderer (row A16 of SURVEY.md 8a): a ResNet-34-style
encoder (reflect padding, InstanceNorm) with a two-level decoder producing C-channel maps at
1/4 resolution -- architecture of ``pgdvs.models.gnt.models.feature_network.ResUNet``
(pgdvs/models/gnt/models/feature_network.py:182-333).  Dense convolutions stay on
PyTorch-ROCm/MIOpen (not a custom-kernel row).  Sub-module names and construction order
follow the reference so that (a) GNT release checkpoints (`feature_net.*` keys) load
unchanged and (b) a seeded random init reproduces the reference's weights."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _inorm(ch):
    return nn.InstanceNorm2d(ch, track_running_stats=False, affine=True)


def _conv(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2, bias=bias, padding_mode="reflect")


class _ResBlock(nn.Module):
    """two 3x3 convs + identity / projected shortcut"""

    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        self.conv1 = _conv(cin, cout, 3, stride)
        self.bn1 = _inorm(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(cout, cout, 3)
        self.bn2 = _inorm(cout)
        self.downsample = downsample

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class _ConvNormElu(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = _conv(cin, cout, k, 1, bias=True)
        self.bn = _inorm(cout)

    def forward(self, x):
        return F.elu(self.bn(self.conv(x)), inplace=True)


class _UpConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _ConvNormElu(cin, cout, 3)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, align_corners=True, mode="bilinear"))


class ResUNet(nn.Module):
    def __init__(self, encoder="resnet34", coarse_out_ch=32, fine_out_ch=32, norm_layer=None, single_net=True):
        super().__init__()
        assert encoder == "resnet34", "only the configuration used by GNT/PGDVS is built"
        self.single_net = single_net
        self.coarse_out_ch = coarse_out_ch
        self.fine_out_ch = coarse_out_ch if single_net else fine_out_ch
        out_ch = coarse_out_ch if single_net else coarse_out_ch + fine_out_ch
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False, padding_mode="reflect")
        self.bn1 = _inorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._stage(64, 64, 3)
        self.layer2 = self._stage(64, 128, 4)
        self.layer3 = self._stage(128, 256, 6)
        self.upconv3 = _UpConv(256, 128)
        self.iconv3 = _ConvNormElu(128 + 128, 128, 3)
        self.upconv2 = _UpConv(128, 64)
        self.iconv2 = _ConvNormElu(64 + 64, out_ch, 3)
        self.out_conv = nn.Conv2d(out_ch, out_ch, 1, 1)

    @staticmethod
    def _stage(cin, cout, n_blocks):
        # every stage downsamples by 2; the shortcut projection is created first (init order)
        down = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=2, bias=False, padding_mode="reflect"), _inorm(cout))
        blocks = [_ResBlock(cin, cout, 2, down)]
        blocks += [_ResBlock(cout, cout, 1, None) for _ in range(n_blocks - 1)]
        return nn.Sequential(*blocks)

    @staticmethod
    def _skip(enc, dec):
        dy, dx = dec.shape[2] - enc.shape[2], dec.shape[3] - enc.shape[3]
        enc = F.pad(enc, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        return torch.cat([dec, enc], dim=1)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x = self.iconv3(self._skip(x2, self.upconv3(x3)))
        x = self.iconv2(self._skip(x1, self.upconv2(x)))
        x = self.out_conv(x)
        if self.single_net:
            return x, x
        return x[:, : self.coarse_out_ch], x[:, -self.fine_out_ch:]
