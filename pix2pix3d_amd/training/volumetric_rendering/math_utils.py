"""Ray/box helpers (reference: training/volumetric_rendering/math_utils.py:23-118, MIT-licensed there).
Only ``get_ray_limits_box`` and ``linspace`` are on the renderer's path (the 'auto' ray-range branch)."""
import torch


def transform_vectors(matrix: torch.Tensor, vectors4: torch.Tensor) -> torch.Tensor:
    """Left-multiply MxM @ NxM (row vectors)."""
    return (matrix @ vectors4.T).T


def normalize_vecs(vectors: torch.Tensor) -> torch.Tensor:
    return vectors / torch.norm(vectors, dim=-1, keepdim=True)


def torch_dot(x: torch.Tensor, y: torch.Tensor):
    return (x * y).sum(-1)


def get_ray_limits_box(rays_o: torch.Tensor, rays_d: torch.Tensor, box_side_length):
    """Slab test of rays against the axis-aligned cube [-L/2, L/2]^3 (math_utils.py:46-98).
    Returns (t_near, t_far), each [..., 1]; rays that miss get (-1, -2)."""
    shape = rays_o.shape
    o = rays_o.detach().reshape(-1, 3)
    d = rays_d.detach().reshape(-1, 3)
    half = box_side_length / 2
    lo = torch.tensor([-half] * 3, device=o.device, dtype=o.dtype)
    hi = -lo
    inv = 1.0 / d
    neg = (inv < 0)
    # per axis: entry through the near slab face, exit through the far one (faces swap for negative directions)
    t_in = (torch.where(neg, hi, lo) - o) * inv
    t_out = (torch.where(neg, lo, hi) - o) * inv
    valid = torch.ones(o.shape[0], dtype=torch.bool, device=o.device)
    tmin, tmax = t_in[:, 0], t_out[:, 0]
    for ax in (1, 2):
        valid = valid & ~((tmin > t_out[:, ax]) | (t_in[:, ax] > tmax))
        tmin = torch.max(tmin, t_in[:, ax])
        tmax = torch.min(tmax, t_out[:, ax])
    tmin = torch.where(valid, tmin, torch.full_like(tmin, -1))
    tmax = torch.where(valid, tmax, torch.full_like(tmax, -2))
    return tmin.reshape(*shape[:-1], 1), tmax.reshape(*shape[:-1], 1)


def linspace(start: torch.Tensor, stop: torch.Tensor, num: int):
    """Tensor-valued linspace along a new leading axis (math_utils.py:101-118)."""
    steps = torch.arange(num, dtype=torch.float32, device=start.device) / (num - 1)
    steps = steps.reshape(-1, *([1] * start.ndim))
    return start[None] + steps * (stop - start)[None]
