"""Oracle shim: named colours (reflect_sampling_nerf_model.py:32,117)."""
import torch

WHITE = torch.tensor([1.0, 1.0, 1.0])
BLACK = torch.tensor([0.0, 0.0, 0.0])
COLORS_DICT = {"white": WHITE, "black": BLACK}
