"""Nature-DQN convolutional trunks (reference pfrl/nn/atari_cnn.py:17-80).
On the GPU with channels_last weights and ReLU the trunk runs as the hand-written f32
MFMA kernels of csrc/qnet.hip (pfrl_amd/nn/mfma_trunk.py); any other configuration
takes stock PyTorch-ROCm (MIOpen conv + hipBLASLt GEMM) with the fused bias + ReLU
launches below."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from pfrl_amd.initializers import init_chainer_default


def constant_bias_initializer(bias=0.0):
    @torch.no_grad()
    def init_bias(m):
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            m.bias.fill_(bias)

    return init_bias


def wants_channels_last(model):
    """True if the first convolution of ``model`` keeps its weights in
    torch.channels_last (``model.to(memory_format=torch.channels_last)``): its input is
    then best delivered in that format too."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            w = m.weight
            return (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last)
                    and not w.is_contiguous())
    return False


def _is_relu(activation):
    return activation is F.relu or activation is torch.relu


def conv_activation(layer, h, activation, planar_out=False):
    """``activation(layer(h))`` for a conv layer.  For ReLU on the GPU with
    channels_last weights: the conv runs without bias, then ONE launch does bias +
    ReLU (and one their backward) instead of PyTorch's separate add / clamp /
    threshold / reduction kernels.  ``planar_out`` (the last convolution before a
    flatten): the fused launch also writes the result in plain NCHW, so the flatten
    and its backward are views rather than layout copies."""
    if (_is_relu(activation) and h.is_cuda and h.dtype == torch.float32
            and isinstance(layer, nn.Conv2d) and layer.bias is not None
            and layer.padding_mode == "zeros"
            and layer.weight.is_contiguous(memory_format=torch.channels_last)):
        from pfrl_amd import ops

        z = F.conv2d(h, layer.weight, None, layer.stride, layer.padding, layer.dilation,
                     layer.groups)   # (the functional form: layer may be a _ConvSlot)
        if ops.bias_relu_supported(z, layer.bias):
            return ops.bias_relu(z, layer.bias, planar=planar_out)
        return activation(z + layer.bias.view(1, -1, 1, 1))
    if isinstance(layer, nn.Conv2d):
        return activation(nn.Conv2d.forward(layer, h))   # (layer may be a _ConvSlot)
    return activation(layer(h))


# Measured on the DQN bench: 16.4 k env-steps/s with this fusion vs 17.7 k without (the
# GEMM without its bias epilogue takes a slower hipBLASLt kernel than the one it
# replaces), so it is off unless PFRL_FUSE_LINEAR=1.
_FUSE_LINEAR = os.environ.get("PFRL_FUSE_LINEAR", "0") == "1"


def linear_activation(layer, h, activation):
    """``activation(layer(h))`` for a linear layer: at minibatch sizes on the GPU the
    bias add + ReLU (forward) and ReLU mask + bias gradient (backward) are one launch
    each next to the GEMMs."""
    if (_FUSE_LINEAR and _is_relu(activation) and h.is_cuda and h.dtype == torch.float32
            and h.dim() == 2 and isinstance(layer, nn.Linear) and layer.bias is not None
            and h.shape[0] <= 256 and layer.out_features % 4 == 0):
        from pfrl_amd import ops

        z = F.linear(h, layer.weight)
        if ops.bias_relu_supported(z, layer.bias):
            return ops.bias_relu(z, layer.bias)
        return activation(z + layer.bias)
    from pfrl_amd.nn.noisy_linear import FactorizedNoisyLinear

    if isinstance(layer, FactorizedNoisyLinear) and _is_relu(activation):
        return layer(h, relu=True)
    return activation(layer(h))   # large batches: the GEMM's own bias epilogue


class _Identity(nn.Module):
    def forward(self, x):
        return x


def fuse_conv_bias_relu(model):
    """Rewrite every ``Conv2d`` directly followed by ``ReLU`` inside an ``nn.Sequential``
    of ``model`` so that bias add + ReLU (and their backward) run as the fused HIP
    launches; the convolution stays MIOpen's.  Parameters are shared, not copied; the
    ReLU slot becomes an identity so that indices (and checkpoints' key prefixes,
    via the ``_ConvSlot`` wrapper's ``weight`` / ``bias`` attributes) are unchanged."""
    for seq in [m for m in model.modules() if isinstance(m, nn.Sequential)]:
        mods = list(seq._modules.items())
        for (ka, a), (kb, b) in zip(mods[:-1], mods[1:]):
            if isinstance(a, nn.Conv2d) and isinstance(b, nn.ReLU):
                seq._modules[ka] = _ConvSlot(a)
                seq._modules[kb] = _Identity()
    return model


class _ConvSlot(nn.Conv2d):
    """A Conv2d (same class, same parameter names) whose forward also applies bias +
    ReLU through the fused path; built around the tensors of an existing layer."""

    def __init__(self, conv):
        nn.Module.__init__(self)
        self.__dict__.update({k: v for k, v in conv.__dict__.items()
                              if k not in ("_parameters", "_buffers", "_modules")})
        self._parameters = conv._parameters
        self._buffers = conv._buffers
        self._modules = conv._modules

    def forward(self, x):
        return conv_activation(self, x, F.relu)


class _AtariCNN(nn.Module):
    def __init__(self, convs, flat, n_output_channels, activation, bias):
        super().__init__()
        self.activation = activation
        self.n_output_channels = n_output_channels
        self.layers = nn.ModuleList(convs)
        self.output = nn.Linear(flat, n_output_channels)
        self.apply(init_chainer_default)
        self.apply(constant_bias_initializer(bias=bias))

    def forward(self, state):
        if _is_relu(self.activation) and state.is_cuda:
            # the whole trunk as hand-written MFMA kernels (one autograd node) when the
            # shapes are inside what csrc/qnet.hip covers; stock route otherwise
            from pfrl_amd.nn import mfma_trunk

            specs = mfma_trunk.plan_for(self.layers, self.output, state)
            if specs is not None:
                return mfma_trunk.trunk_forward(state, specs, list(self.layers), self.output)
        h = state
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            h = conv_activation(layer, h, self.activation, planar_out=(i == last))
        return linear_activation(self.output, h.reshape(h.size(0), -1), self.activation)


class LargeAtariCNN(_AtariCNN):
    """Nature 2015: conv 32x8x8/4, 64x4x4/2, 64x3x3/1, fc 3136 -> 512."""

    def __init__(self, n_input_channels=4, n_output_channels=512, activation=F.relu, bias=0.1):
        self.n_input_channels = n_input_channels
        super().__init__([nn.Conv2d(n_input_channels, 32, 8, stride=4),
                          nn.Conv2d(32, 64, 4, stride=2),
                          nn.Conv2d(64, 64, 3, stride=1)], 3136, n_output_channels, activation,
                         bias)


class SmallAtariCNN(_AtariCNN):
    """NIPS-workshop 2013: conv 16x8x8/4, 32x4x4/2, fc 2592 -> 256."""

    def __init__(self, n_input_channels=4, n_output_channels=256, activation=F.relu, bias=0.1):
        self.n_input_channels = n_input_channels
        super().__init__([nn.Conv2d(n_input_channels, 16, 8, stride=4),
                          nn.Conv2d(16, 32, 4, stride=2)], 2592, n_output_channels, activation,
                         bias)
