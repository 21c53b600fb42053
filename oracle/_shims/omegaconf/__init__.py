"""Stub of `omegaconf` (reference only uses it for type checks in utils). Test infrastructure only."""


class OmegaConf:  # noqa: D401
    pass


class DictConfig(dict):
    pass


class ListConfig(list):
    pass
