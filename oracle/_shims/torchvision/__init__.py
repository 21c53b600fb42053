"""Minimal stand-in for the four torchvision symbols the reference's hot path touches
(SURVEY.md 8c). Only what nnDetection needs at import time + BoxCoder's constructor constant.
Test infrastructure only - never imported by the product."""
