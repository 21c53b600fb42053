"""Encoder stage block: [conv(stride) -> conv(1)] x num_blocks. Mirrors StackedBlock / StackedConvBlock2
(nndet/arch/blocks/basic.py:45-151); module names (`convs.{i}.{0,1}`) match the reference state dict."""
from typing import Callable, Sequence

import torch
import torch.nn as nn


class StackedConvBlock2(nn.Module):
    expansion = 2

    def __init__(self, conv: Callable, in_channels: int, conv_kernel, stride=None, out_channels: int = None,
                 max_out_channels: int = None, num_blocks: int = 1, **kwargs):
        super().__init__()
        if out_channels is not None and max_out_channels is not None and out_channels > max_out_channels:
            raise ValueError("Output channels can not be larger than max output channels")
        if out_channels is None:
            out_channels = in_channels * self.expansion
        if max_out_channels is not None and out_channels > max_out_channels:
            out_channels = max_out_channels
        stride = 1 if stride is None else stride
        if not isinstance(conv_kernel, Sequence):
            conv_kernel = [conv_kernel] * conv.dim
        padding = tuple((i - 1) // 2 for i in conv_kernel)
        blocks = [self.build_block(conv, in_channels, out_channels, conv_kernel, stride, padding, **kwargs)]
        for _ in range(num_blocks - 1):
            blocks.append(self.build_block(conv, out_channels, out_channels, conv_kernel, 1, padding, **kwargs))
        self.convs = nn.Sequential(*blocks)
        self.out_channels = out_channels
        # every conv output that is consumed by the NEXT conv of this block only: its norm + ReLU is applied by that consumer while
        # staging its input (arch/conv.py: deferred normalisation). The block's last output is decided by the encoder.
        mods = [m for b in blocks for m in b]
        for m in mods[:-1]:
            if hasattr(m, "defer_output"):
                m.defer_output = True

    @staticmethod
    def build_block(conv, in_channels, out_channels, kernel_size, stride, padding, **kwargs) -> nn.Module:
        return nn.Sequential(
            conv(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride, padding=padding, **kwargs),
            conv(in_channels=out_channels, out_channels=out_channels, kernel_size=kernel_size, stride=1, padding=padding, **kwargs),
        )

    def get_output_channels(self) -> int:
        return self.out_channels

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.convs(x)
