from omegaconf import OmegaConf, DictConfig, ListConfig  # noqa: F401
