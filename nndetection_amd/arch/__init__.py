from .conv import Generator, ConvInstanceRelu, ConvGroupRelu  # noqa: F401
from .blocks import StackedConvBlock2  # noqa: F401
from .encoder import Encoder  # noqa: F401
from .decoder import UFPNModular  # noqa: F401
from .heads import BCECLassifier, GIoURegressor, DetectionHeadHNMNative, Scale  # noqa: F401
from .segmenter import DiCESegmenterFgBg  # noqa: F401
