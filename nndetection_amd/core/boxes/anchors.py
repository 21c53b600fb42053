"""3D anchor generator on the GPU (csrc/boxes.hip:k_anchor_grid). Mirrors AnchorGenerator3DS
(nndet/core/boxes/anchors.py:459-559) + the forward/caching logic of AnchorGenerator2D/3D (:102-118,211-263,337-377).
"""
from itertools import product
from typing import List, Sequence

import torch

from ... import _lib as L


class AnchorGenerator3DS(torch.nn.Module):
    def __init__(self, width, height, depth, **kwargs):
        super().__init__()
        fix = lambda v: [tuple(x) if isinstance(x, (list, tuple)) else (x,) for x in v]
        self.width, self.height, self.depth = fix(width), fix(height), fix(depth)
        assert len(self.width) == len(self.height) == len(self.depth)
        self.cell_anchors = None
        self._cache = {}
        self.num_anchors_per_level: List[int] = None

    @staticmethod
    def generate_anchors(width, height, depth, dtype=torch.float, device="cpu") -> torch.Tensor:
        """[-w/2, -h/2, w/2, h/2, -d/2, d/2] for product(width, height, depth) (anchors.py:526-549)."""
        s = torch.tensor(list(product(width, height, depth)), dtype=dtype, device=device) / 2
        return torch.stack([-s[:, 0], -s[:, 1], s[:, 0], s[:, 1], -s[:, 2], s[:, 2]], dim=1)

    def set_cell_anchors(self, dtype, device):
        if self.cell_anchors is None:
            self.cell_anchors = [self.generate_anchors(w, h, d, torch.float32, device)
                                 for w, h, d in zip(self.width, self.height, self.depth)]

    def num_anchors_per_location(self) -> List[int]:
        return [len(w) * len(h) * len(d) for w, h, d in zip(self.width, self.height, self.depth)]

    def get_num_acnhors_per_level(self) -> List[int]:   # (sic) anchors.py:253-263
        if self.num_anchors_per_level is None:
            raise RuntimeError("Need to forward features maps before get_num_acnhors_per_level can be called")
        return self.num_anchors_per_level

    def grid_anchors(self, grid_sizes, strides):
        assert len(grid_sizes) == len(strides) == len(self.cell_anchors)
        anchors, per_level = [], []
        for size, stride, base in zip(grid_sizes, strides, self.cell_anchors):
            A = base.shape[0]
            out = torch.empty((int(size[0]) * int(size[1]) * int(size[2]) * A, 6), dtype=torch.float32, device=base.device)
            L.call("nndet_anchors3d_grid_f32", L.ptr(base), A, int(size[0]), int(size[1]), int(size[2]),
                   int(stride[0]), int(stride[1]), int(stride[2]), L.ptr(out), L.stream())
            anchors.append(out)
            per_level.append(out.shape[0])
        return anchors, per_level

    def cached_grid_anchors(self, grid_sizes, strides):
        key = str(grid_sizes + strides)
        if key not in self._cache:
            per, npl = self.grid_anchors(grid_sizes, strides)
            self._cache[key] = (per, npl, torch.cat(per))
        self.num_anchors_per_level = self._cache[key][1]
        return self._cache[key]

    def forward(self, image_list: torch.Tensor, feature_maps: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        grid_sizes = [list(fm.shape[2:]) for fm in feature_maps]
        image_size = image_list.shape[2:]
        strides = [[int(i / s) for i, s in zip(image_size, fm)] for fm in grid_sizes]
        self.set_cell_anchors(torch.float32, feature_maps[0].device)
        _, _, cat = self.cached_grid_anchors(grid_sizes, strides)
        return [cat for _ in range(image_list.shape[0])]    # the same tensor object for every image (anchors.py:230-237)


def get_anchor_generator(dim: int, s_param: bool = False):
    if dim != 3 or not s_param:
        raise L.NndetError("only the 3D 'S' parametrisation (width/height/depth plans) is on the MI355X hot path")
    return AnchorGenerator3DS
