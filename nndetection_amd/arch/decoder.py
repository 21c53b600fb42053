"""U-shaped FPN decoder. Mirrors BaseUFPN / UFPNModular (nndet/arch/decoder/base.py:28-417) for the V001
configuration (transposed-conv upsampling, 1 lateral + 1 out conv per level, no norm / activation, no fusion convs).

One deliberate difference: `skip_unused_out` (default True) does not COMPUTE output convs
whose result nothing consumes (`out.P1` with decoder_levels (2,3,4,5): heads read levels 2-5, the segmenter
level 0). The parameters stay in the module (state-dict parity), the forward result for that level is `None`.
"""
import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib as L
from .conv import conv_kwargs_helper, ensure_materialized


class UFPNModular(nn.Module):
    def __init__(self, conv: Callable, strides, in_channels: Sequence[int], conv_kernels, decoder_levels,
                 fixed_out_channels: int, min_out_channels: int = 8, upsampling_mode: str = "nearest",
                 num_lateral: int = 1, norm_lateral: bool = False, activation_lateral: bool = False,
                 num_out: int = 1, norm_out: bool = False, activation_out: bool = False,
                 num_fusion: int = 0, norm_fusion: bool = False, activation_fusion: bool = False,
                 skip_unused_out: bool = True, used_levels: Optional[Sequence[int]] = None):
        super().__init__()
        if len(strides) != len(in_channels):
            raise ValueError("Strides must contain same number of elements as channels.")
        if upsampling_mode.lower() != "transpose" or num_fusion != 0 or num_lateral != 1 or num_out != 1:
            raise NotImplementedError("only the RetinaUNetV001 decoder configuration (v001.yaml:63-73) is on the hot path")
        self.dim = conv.dim
        self.num_level = len(in_channels)
        self.in_channels = list(in_channels)
        self.decoder_levels = decoder_levels
        strides = [s if isinstance(s, Sequence) else (s,) * self.dim for s in strides]
        self.strides = [tuple(int(s1 / s0) for s1, s0 in zip(strides[i], strides[i - 1])) for i in range(1, len(strides))]
        if isinstance(conv_kernels, int):
            conv_kernels = [conv_kernels] * self.num_level
        self.conv_kernels = [tuple([ck] * self.dim) if isinstance(ck, int) else tuple(ck) for ck in conv_kernels]
        self.conv_paddings = [tuple((i - 1) // 2 for i in ck) for ck in self.conv_kernels]
        self.fixed_out_channels, self.min_out_channels = fixed_out_channels, min_out_channels
        self.out_channels = self.compute_output_channels()
        lat_kw = conv_kwargs_helper(norm_lateral, activation_lateral)
        out_kw = conv_kwargs_helper(norm_out, activation_out)
        oc = self.out_channels
        self.lateral = nn.ModuleDict({f"P{l}": nn.Sequential(conv(self.in_channels[l], oc[l], kernel_size=1, padding=0, stride=1, **lat_kw))
                                      for l in range(self.num_level)})
        self.out = nn.ModuleDict({f"P{l}": nn.Sequential(conv(oc[l], oc[l], kernel_size=self.conv_kernels[l],
                                                              padding=self.conv_paddings[l], stride=1, **out_kw))
                                  for l in range(self.num_level)})
        self.up = nn.ModuleDict({f"P{l}": conv(oc[l], oc[l - 1], kernel_size=self.strides[l - 1], stride=self.strides[l - 1],
                                               transposed=True, add_norm=False, add_act=False)
                                 for l in range(1, self.num_level)})
        self.skip_unused_out = skip_unused_out
        self.used_levels = None if used_levels is None else set(used_levels)

    def compute_output_channels(self) -> List[int]:
        """decoder/base.py:182-199"""
        out = [self.fixed_out_channels] * self.num_level
        if self.decoder_levels is not None:
            for ol in [l for l in range(self.num_level) if l < min(self.decoder_levels)][::-1]:
                out[ol] = max(self.min_out_channels, out[ol + 1] // 2)
        return out

    def get_channels(self) -> List[int]:
        return self.out_channels

    # The full-resolution tail of the decoder -- lateral P0, the last top-down step up.P1 (+ its residual add) and out.P0 -- only feeds
    # the segmentation branch; the detection head reads the levels >= 1. On a side stream these three HBM-bound launches (0.2 + 0.43 +
    # 0.33 ms at 160x160x96, and their backward passes) run next to the MFMA-bound head instead of in front of it. `tail_event` tells
    # the consumer of level 0 when it is ready. NNDET_DECODER_TAIL=0: everything on the caller's stream.
    split_tail = os.environ.get("NNDET_DECODER_TAIL", "1") != "0"
    # ... and, when nothing but level 0 reads level 1 either (the detection head starts at level 2 and out.P1 is skipped), lateral P1 and
    # the top-down step up.P2 that forms x_1 as well: 0.07 + 0.23 ms of HBM-bound launches leave the chain in front of the head.
    # NNDET_DECODER_TAIL1=0: level 1 stays on the caller's stream.
    split_tail1 = os.environ.get("NNDET_DECODER_TAIL1", "1") != "0"
    _tail_streams: dict = {}
    # The lateral 1x1x1 convolutions only need their own encoder stage. Issued from a hook of the encoder as soon as that stage's
    # output exists (the detector installs it), the HBM-bound full- and half-resolution laterals (0.28 + 0.30 ms at 160x160x96) run on
    # side streams UNDER the deep encoder stages, whose launches of 100-400 workgroups leave most of the 256 CUs idle (timeline of
    # round 3: the GPU ran one kernel at a time from e2 to the end of the encoder). MEASURED SLOWER and therefore OFF by default
    # (A/B in one session, profiles/round3_ab_early_lateral.txt: 225.3 / 227.1 patches/s with, 229.8 / 228.9 without): the laterals
    # then compete for HBM with the stage-1 / stage-2 convolutions that follow their fork point, which costs more than the idle CUs
    # under e3-e5 give back. NNDET_EARLY_LATERAL=1 enables it.
    early_laterals = os.environ.get("NNDET_EARLY_LATERAL", "0") != "0"
    _early_streams: dict = {}

    def early_lateral(self, level: int, fm: torch.Tensor) -> None:
        """Encoder stage hook: lateral P<level> of `fm` on a side stream, forked from the caller's stream now. The result is picked
        up (and the stream joined) by the next forward(); the deepest level is not worth a fork (the decoder needs it first)."""
        if not (self.early_laterals and fm.is_cuda and level < self.num_level - 1):
            return
        ensure_materialized(fm)        # (arch/conv.py NORM_INPUT_FUSE: this reader comes before the convolution that would have written it)
        dev = fm.device
        main = torch.cuda.current_stream(dev)
        need0 = not (self.skip_unused_out and self.used_levels is not None and 0 not in self.used_levels)
        if level == 0 and self.split_tail and self.num_level >= 3 and need0:
            side = UFPNModular._tail_streams.get(dev.index or 0)        # level 0 stays on the tail stream of forward()
            if side is None:
                side = UFPNModular._tail_streams[dev.index or 0] = L.new_stream("tail", dev, int(os.environ.get("NNDET_PRIO_TAIL", "0")))
        else:
            side = UFPNModular._early_streams.get(dev.index or 0)
            if side is None:
                side = UFPNModular._early_streams[dev.index or 0] = L.new_stream("early", dev)
        side.wait_stream(main)                                   # this stage's output is ready
        fm.record_stream(side)
        with torch.cuda.stream(side):
            lat = self.lateral[f"P{level}"](fm)
            ev = torch.cuda.Event()
            ev.record(side)
        if not hasattr(self, "_early"):
            self._early = {}
        self._early[level] = (fm, lat, ev, side)

    def forward(self, inp_seq: Sequence[torch.Tensor]) -> List[Optional[torch.Tensor]]:
        self.tail_event = None
        need0 = not (self.skip_unused_out and self.used_levels is not None and 0 not in self.used_levels)
        split = self.split_tail and self.num_level >= 3 and inp_seq[0].is_cuda and need0
        fpn: List[Optional[torch.Tensor]] = [None] * self.num_level
        early = getattr(self, "_early", None) or {}
        self._early = {}
        for l, (src, lat, ev, st) in early.items():              # laterals the encoder hook already issued (same input tensor only)
            if l < len(inp_seq) and src is inp_seq[l]:
                fpn[l] = lat
                cons = UFPNModular._tail_streams.get(lat.device.index or 0) if (l == 0 and split) else None
                if cons is None or cons != st:                   # consumed on another stream than it was produced on: join
                    cur = torch.cuda.current_stream(lat.device) if cons is None else cons
                    cur.wait_event(ev)
                    lat.record_stream(cur)
        absorb = self._absorbing(inp_seq[0])
        tail1 = bool(split and self.split_tail1 and self.num_level >= 4 and self.skip_unused_out and self.used_levels is not None
                     and 1 not in self.used_levels and fpn[1] is None)
        if split:
            dev = inp_seq[0].device
            main = torch.cuda.current_stream(dev)
            side = UFPNModular._tail_streams.get(dev.index or 0)
            if side is None:
                side = UFPNModular._tail_streams[dev.index or 0] = L.new_stream("tail", dev, int(os.environ.get("NNDET_PRIO_TAIL", "0")))
            if fpn[0] is None and not absorb:
                side.wait_stream(main)                           # the encoder outputs are ready
                inp_seq[0].record_stream(side)
                with torch.cuda.stream(side):
                    fpn[0] = self.lateral["P0"](inp_seq[0])
        for l, fm in enumerate(inp_seq):
            if fpn[l] is None and not (l == 0 and absorb) and not (l == 1 and tail1):
                fpn[l] = self.lateral[f"P{l}"](fm)
        xs: List[Optional[torch.Tensor]] = [None] * self.num_level
        x = fpn[self.num_level - 1]
        xs[self.num_level - 1] = x
        for level in range(self.num_level - 2, (1 if tail1 else 0) if split else -1, -1):
            # x_l = lateral_l + up_{l+1}(x_{l+1})  (decoder/base.py:405-413): the add is the epilogue of the transposed conv
            x = self._top_down0(x, inp_seq[0]) if (level == 0 and absorb) else self.up[f"P{level + 1}"](x, residual=fpn[level])
            xs[level] = x
        outs: List[Optional[torch.Tensor]] = [None] * self.num_level
        if split:
            ev = torch.cuda.Event()
            ev.record(main)                                      # x_1 (tail1: x_2 and the encoder outputs) is ready
            side.wait_event(ev)
            if tail1:
                xs[2].record_stream(side)
                inp_seq[1].record_stream(side)
                with torch.cuda.stream(side):
                    fpn[1] = self.lateral["P1"](inp_seq[1])
                    xs[1] = self.up["P2"](xs[2], residual=fpn[1])
            else:
                xs[1].record_stream(side)
            with torch.cuda.stream(side):
                xs[0] = self._top_down0(xs[1], inp_seq[0]) if absorb else self.up["P1"](xs[1], residual=fpn[0])
                outs[0] = self._out0(xs[0])
                self.tail_event = torch.cuda.Event()
                self.tail_event.record(side)
        self._ragged_out(xs, 1 if split else 0)
        for level in range(1 if split else 0, self.num_level):
            if self.skip_unused_out and self.used_levels is not None and level not in self.used_levels:
                outs[level] = None
            else:
                outs[level] = self._out0(xs[0]) if level == 0 else self.out[f"P{level}"](xs[level])
        return outs

    # The detection head batches its levels as ONE ragged [rows, C_p] buffer, level-major (arch/pyramid.py: cat_levels). When the
    # levels it reads are a contiguous run of equally wide out convolutions, those write their outputs straight into consecutive
    # slices of one allocation and the concatenation becomes a view (45 MB copied per step otherwise). NNDET_RAGGED_OUT=0: off.
    ragged_out = os.environ.get("NNDET_RAGGED_OUT", "1") != "0"

    def _ragged_out(self, xs, first: int) -> None:
        from .conv import BaseConvNormAct
        from ..layout import cpad
        if not (self.ragged_out and self.skip_unused_out and self.used_levels is not None):
            return
        lv = sorted(l for l in self.used_levels if l >= max(first, 1))
        if len(lv) < 2 or lv != list(range(lv[0], lv[0] + len(lv))) or any(xs[l] is None or not xs[l].is_cuda for l in lv):
            return
        mods = []
        for l in lv:
            blk = self.out[f"P{l}"]
            m = blk[0] if len(list(blk.children())) == 1 else None
            if not isinstance(m, BaseConvNormAct) or m.transposed or m.s != (1, 1, 1) or m.k != (3, 3, 3) or m.p != (1, 1, 1) or m.norm_groups:
                return
            mods.append(m)
        cp = cpad(mods[0].out_channels)
        if any(cpad(m.out_channels) != cp for m in mods) or xs[lv[0]].dtype not in (torch.bfloat16, torch.float16, torch.float32):
            return
        rows = [xs[l].shape[0] * xs[l].shape[2] * xs[l].shape[3] * xs[l].shape[4] for l in lv]
        base = torch.empty((sum(rows), cp), dtype=xs[lv[0]].dtype, device=xs[lv[0]].device)
        r0 = 0
        for l, m, nr in zip(lv, mods, rows):
            x = xs[l]
            m._out_buf = base[r0:r0 + nr].view(x.shape[0], x.shape[2], x.shape[3], x.shape[4], cp)
            r0 += nr

    # Set by the detector for ONE forward pass (core/retina.py): a training step without prediction whose segmentation branch
    # computes decoder.out.P0 + output conv + loss as one composed 32 -> 1 convolution (arch/segmenter.py: _SegBranchFn) gets the
    # level-0 map BEFORE the output convolution, tagged with that module; nothing else reads level 0.
    defer_out0 = False
    # ... and, with it, may absorb the level-0 lateral: then lateral P0 is not run either, the last top-down step is the transposed
    # convolution ALONE with both biases (b_up + b_lat: the add of two [C] vectors is differentiable, so both get their gradient), and
    # the segmentation branch reads the encoder's level-0 map itself.
    absorb_lat0 = False
    # ... and the last top-down step itself (arch/segmenter.py: NNDET_SEG_UP): then up.P1 is not run either, level 0 of the returned
    # list is x_1 (the half-resolution map) tagged with the three modules, and the branch composes them.
    absorb_up0 = False

    def _out0(self, x0: torch.Tensor) -> torch.Tensor:
        if self.defer_out0:
            x0._nndet_pre_out = self.out["P0"][0]
            return x0
        return self.out["P0"](x0)

    def _absorbing(self, inp0: torch.Tensor) -> bool:
        from .conv import deferred
        return bool(self.defer_out0 and self.absorb_lat0 and inp0.is_cuda and deferred(inp0) is None and self.num_level >= 2)

    def _top_down0(self, x1: torch.Tensor, inp0: torch.Tensor) -> torch.Tensor:
        """x_0 of a pass that absorbs the lateral: up_1(x_1) + b_up + b_lat, tagged with (lateral module, its input)."""
        from .conv import _ConvFn
        up, lat = self.up["P1"], self.lateral["P0"][0]
        if self.absorb_up0:
            x1._nndet_pre_lat = (lat, inp0)
            x1._nndet_pre_up = up
            return x1
        biases = [b for b in (up.conv.bias, lat.conv.bias) if b is not None]
        bias = (biases[0] + biases[1]) if len(biases) == 2 else (biases[0] if biases else None)
        x0, _ = _ConvFn.apply(x1, None, False, up.conv.weight, bias, up, None, False)
        x0._nndet_pre_lat = (lat, inp0)
        return x0


UFPNModular.early_lateral._nndet_handles_pre = True      # (arch/encoder.py: this stage hook writes an unwritten stage output itself before it reads it)
