import torch


class AnchorGenerator(torch.nn.Module):
    """Only used as a TypeVar bound by the reference (anchors.py:12,17)."""
