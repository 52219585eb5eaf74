"""``grid_sample`` switchable replacement (reference: torch_utils/ops/grid_sample_gradfix.py:28-77).  The training
loop leaves it disabled (training_loop.py:282) and only the augmentation pipeline, which is out of scope, calls
it — so this keeps the module-level switch and forwards to torch."""
import torch

enabled = False


def grid_sample(input, grid):
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)
