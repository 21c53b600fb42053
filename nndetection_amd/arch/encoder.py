"""Modular encoder. Mirrors nndet/arch/encoder/modular.py:28-157 (same ctor arguments and helper methods)."""
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn as nn

from .conv import ensure_materialized


class Encoder(nn.Module):
    def __init__(self, conv: Callable, conv_kernels, strides, block_cls, in_channels: int, start_channels: int,
                 stage_kwargs=None, out_stages: Sequence[int] = None, max_channels: int = None, first_block_cls=None):
        super().__init__()
        self.num_stages = len(conv_kernels)
        self.dim = conv.dim
        if stage_kwargs is None:
            stage_kwargs = [{}] * self.num_stages
        elif isinstance(stage_kwargs, dict):
            stage_kwargs = [stage_kwargs] * self.num_stages
        assert len(stage_kwargs) == len(conv_kernels)
        self.out_stages = list(range(self.num_stages)) if out_stages is None else out_stages
        first_block_cls = block_cls if first_block_cls is None else first_block_cls
        if isinstance(strides[0], int):
            strides = [tuple([s] * self.dim) for s in strides]
        self.strides = strides
        stages, self.out_channels = [], []
        for sid in range(self.num_stages):
            if sid == 0:
                blk = first_block_cls(conv=conv, in_channels=in_channels, out_channels=start_channels,
                                      conv_kernel=conv_kernels[sid], stride=None, max_out_channels=max_channels, **stage_kwargs[sid])
            else:
                blk = block_cls(conv=conv, in_channels=in_channels, out_channels=None, conv_kernel=conv_kernels[sid],
                                stride=strides[sid - 1], max_out_channels=max_channels, **stage_kwargs[sid])
            in_channels = blk.get_output_channels()
            self.out_channels.append(in_channels)
            stages.append(blk)
        self.stages = nn.ModuleList(stages)
        self.defer_outputs = False       # see set_defer_outputs
        self.fuse_grad_accum = False     # see set_fuse_grad_accum
        self.stage_hook = None           # optional callable(output index, tensor) run right after a stage output exists (decoder.early_lateral)

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        outputs = []
        from . import conv as _cv
        for sid, module in enumerate(self.stages):
            pre = getattr(x, "_nndet_pre", None)
            # (arch/conv.py NORM_INPUT_FUSE: stage 0 may hand an UNWRITTEN normalised tensor to stage 1 -- only here, where the next stage is
            # known to run, nobody observes the stage output through a hook, and whatever is left unwritten is written right after)
            _cv._fill_scope[0] = (sid == 0 and self.num_stages > 1 and not any(_cv._observed(m) for m in module.modules()))
            try:
                x_in, x = x, module(x)
            finally:
                _cv._fill_scope[0] = False
            if pre is not None:                  # (set_early_consumer: whatever the stage did with the tag, order this stream behind the
                if len(pre) > 4:                 #  materialising pass -- or run it: NORM_INPUT_FUSE -- before the tensor is handed to anybody else)
                    ensure_materialized(x_in)
                elif x_in.is_cuda:
                    torch.cuda.current_stream(x_in.device).wait_event(pre[3])
                del x_in._nndet_pre
            if sid in self.out_stages:
                outputs.append(x)
                if self.fuse_grad_accum and sid + 1 < self.num_stages and torch.is_grad_enabled() and x.requires_grad:
                    # (see set_fuse_grad_accum / arch/conv.py:_ConvFn.backward; "stream": where this activation's own backward
                    # node will run -- the second consumer orders that stream behind its in-place accumulation)
                    x._nndet_gacc = {"buf": None, "stream": torch.cuda.current_stream(x.device) if x.is_cuda else None}
                if self.stage_hook is not None:
                    if not getattr(getattr(self.stage_hook, "__func__", self.stage_hook), "_nndet_handles_pre", False):
                        ensure_materialized(x)       # (a hook we do not know may read the stage output now; decoder.early_lateral writes it itself)
                    self.stage_hook(len(outputs) - 1, x)
        return outputs

    def set_fuse_grad_accum(self, on: bool) -> None:
        """A stage output has two consumers, the next stage's first convolution and the decoder's lateral convolution; autograd adds
        their two input gradients in a third pass (2 reads + 1 write of the activation: 0.3 ms at full resolution). With this on, the
        consumer whose backward runs second ADDS its data gradient into the first one's buffer (nndet_conv3d_backward_data_acc) and
        returns no gradient of its own. Only the detector switches it on, and only when both consumers are our convolutions."""
        self.fuse_grad_accum = bool(on)

    def set_early_consumer(self, on: bool) -> None:
        """Stage 0's output is normalised on an auxiliary stream while the first convolution of stage 1 already reads the pre-norm
        tensor (arch/conv.py: EARLY_CONSUMER): the 1.26 GB norm pass leaves the serial chain of the forward pass. Only the detector
        switches it on."""
        if self.num_stages < 2:
            return
        last0 = [m for m in self.stages[0].modules() if hasattr(m, "early_output")]
        first1 = [m for m in self.stages[1].modules() if hasattr(m, "early_input")]
        if last0 and first1:
            last0[-1].early_output = bool(on)
            first1[0].early_input = bool(on)

    def set_defer_outputs(self, on: bool) -> None:
        """Hand the stage outputs on as DEFERRED activations (pre-norm tensor + coefficients, arch/conv.py). Only the detector
        switches this on, and only when every consumer of the encoder outputs is one of our own convolutions (next stage, decoder
        laterals); stand-alone the encoder returns ordinary tensors like the reference's."""
        self.defer_outputs = bool(on)
        for stage in self.stages:
            last = [m for m in stage.modules() if hasattr(m, "defer_output")][-1]
            last.defer_output = bool(on)

    def get_channels(self) -> List[int]:
        return [self.out_channels[s] for s in range(self.num_stages) if s in self.out_stages]

    def get_strides(self) -> List[List[int]]:
        out = []
        for sid in range(self.num_stages):
            out.append([1] * self.dim if sid == 0 else [a * b for a, b in zip(out[sid - 1], self.strides[sid - 1])])
        return out
