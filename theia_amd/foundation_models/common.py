"""Teacher feature-size constants (mirror of the reference's foundation_models/common.py:18-50).

Only the constants/lookup that the hot path needs; the teacher models themselves are out of scope (their
features are pre-extracted offline)."""
import math

import torch

MODELS = [
    "facebook/dinov2-large",
    "facebook/sam-vit-huge",
    "google/vit-huge-patch14-224-in21k",
    "llava-hf/llava-1.5-7b-hf",
    "openai/clip-vit-large-patch14",
    "LiheYoung/depth-anything-large-hf",
]

# (latent_dim, height, width)
MODEL_FEATURE_SIZES = {
    "facebook/dinov2-large": (1024, 16, 16),
    "facebook/sam-vit-huge": (256, 64, 64),
    "google/vit-huge-patch14-224-in21k": (1280, 16, 16),
    "llava-hf/llava-1.5-7b-hf": (1024, 24, 24),
    "openai/clip-vit-large-patch14": (1024, 16, 16),
    "LiheYoung/depth-anything-large-hf": (32, 64, 64),
}


def get_model_feature_size(model_name: str, keep_spatial: bool = False, return_torch_size: bool = False):
    size = MODEL_FEATURE_SIZES[model_name]
    if not keep_spatial:
        size = (size[0], math.prod(size[1:]))
    if return_torch_size:
        size = torch.Size(size)
    return size
