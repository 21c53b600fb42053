"""MI355X-native (gfx950) 3D Retina U-Net hot path for nnDetection: hand-written HIP kernels behind
the reference's module / function interfaces. See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
