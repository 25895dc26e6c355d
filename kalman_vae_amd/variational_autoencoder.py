"""Frame encoder and decoders of the Kalman VAE (reference:
kvae/variational_autoencoder.py:8-161).

These are stock PyTorch-ROCm modules (three stride-2 3x3 convolutions down,
one linear + three stride-2 transposed convolutions up; MIOpen runs them);
they are here so that the full model exists on this side of the drop-in
boundary, not because they are part of the accelerated path.  Sub-module names
and shapes equal the reference's, so its checkpoints load unchanged.
"""
import torch.distributions as D
import torch.nn as nn
import torch.nn.functional as F

CHANNELS = 32          # feature maps of every hidden convolution
STAGES = 3             # stride-2 stages on each side


def _halved(image_size):
    """Spatial size after STAGES convolutions with kernel 3, stride 2, padding 1."""
    h, w = image_size
    for _ in range(STAGES):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return h, w


def _down(cin, cout):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=2, padding=1)


def _up(cin, cout):
    return nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1)


class Encoder(nn.Module):
    """x (R, C, H, W) -> Normal(mean (R, a_dim), softplus std)."""

    def __init__(self, image_size, image_channels, a_dim):
        super().__init__()
        self.conv1 = _down(image_channels, CHANNELS)
        self.conv2 = _down(CHANNELS, CHANNELS)
        self.conv3 = _down(CHANNELS, CHANNELS)
        h, w = _halved(image_size)
        self.fc_mean = nn.Linear(CHANNELS * h * w, a_dim)
        self.fc_std = nn.Linear(CHANNELS * h * w, a_dim)

    def forward(self, x):
        for conv in (self.conv1, self.conv2, self.conv3):
            x = F.relu(conv(x))
        flat = x.flatten(1)
        return D.Normal(loc=self.fc_mean(flat), scale=F.softplus(self.fc_std(flat)))


class _DecoderTrunk(nn.Module):
    """a (R, a_dim) -> feature maps (R, 32, H/4, W/4) shared by both decoders."""

    def __init__(self, a_dim, image_size):
        super().__init__()
        self.conv_output_size = _halved(image_size)
        h, w = self.conv_output_size
        self.fc = nn.Linear(a_dim, CHANNELS * h * w)
        self.deconv1 = _up(CHANNELS, CHANNELS)
        self.deconv2 = _up(CHANNELS, CHANNELS)

    def features(self, a):
        x = F.relu(self.fc(a)).view(-1, CHANNELS, *self.conv_output_size)
        return F.relu(self.deconv2(F.relu(self.deconv1(x))))


class BernoulliDecoder(_DecoderTrunk):
    def __init__(self, a_dim, image_size, image_channels):
        super().__init__(a_dim, image_size)
        self.deconv3 = _up(CHANNELS, image_channels)

    def forward(self, a):
        return D.Bernoulli(logits=self.deconv3(self.features(a)))


class GaussianDecoder(_DecoderTrunk):
    def __init__(self, a_dim, image_size, image_channels):
        super().__init__(a_dim, image_size)
        self.deconv3_mean = _up(CHANNELS, image_channels)
        self.deconv3_std = _up(CHANNELS, image_channels)

    def forward(self, a):
        x = self.features(a)
        return D.Normal(loc=self.deconv3_mean(x), scale=F.softplus(self.deconv3_std(x)))
