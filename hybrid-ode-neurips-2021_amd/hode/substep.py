"""``options["step_size"]`` of torchdiffeq's fixed-grid solvers: integrate on the solver's own grid ``t[0] + k*step_size``
(last point clipped to ``t[-1]``) and read the requested output times off it by linear interpolation
(torchdiffeq 0.2.2 ``_grid_constructor_from_step_size`` / ``_linear_interp``; reached from ``DecoderReal`` with
``ode_step_size = step / ode_step_div``, run_real.py:51,63, model.py:826).

The kernels integrate over any increasing grid, so sub-stepping is: build that grid (in ``t``'s dtype, as torchdiffeq does),
run the SAME kernels over it, and gather.  An output time that coincides with a grid point returns that grid state
exactly; anything else is ``y0 + (t - t0) / (t1 - t0) * (y1 - y0)`` on the bracketing interval -- differentiable through
``index_select``, so the adjoint kernels see cotangents only at the rows that were read.
"""
import torch


def fixed_grid(t, step_size):
    """The solver grid torchdiffeq builds for ``step_size`` (same dtype and device as ``t``)."""
    t0, t_end = t[0], t[-1]
    n = int(torch.ceil((t_end - t0) / step_size + 1).item())
    grid = torch.arange(0, n, dtype=t.dtype, device=t.device) * step_size + t0
    grid[-1] = t_end
    return grid


def read_outputs(h_grid, grid, t):
    """h at the output times ``t`` from the states ``h_grid`` (G, B, D) on ``grid``."""
    g, tt = grid.detach().cpu(), t.detach().cpu()
    # output j (>= 1) belongs to the first interval (t0, t1] with t1 >= t[j]
    hi = torch.searchsorted(g, tt, right=False).clamp_(min=1, max=g.numel() - 1)
    hi[0] = 0
    exact = g[hi] == tt
    if bool(exact.all()):
        return h_grid.index_select(0, hi.to(h_grid.device))
    lo = (hi - 1).clamp_(min=0)
    slope = ((tt - g[lo]) / (g[hi] - g[lo])).to(h_grid.dtype)
    slope[exact] = 1.0
    slope[0] = 0.0
    y0 = h_grid.index_select(0, lo.to(h_grid.device))
    y1 = h_grid.index_select(0, hi.to(h_grid.device))
    s = slope.to(h_grid.device).view(-1, 1, 1)
    lerp = y0 + s * (y1 - y0)
    return torch.where(exact.to(h_grid.device).view(-1, 1, 1), y1, lerp)


def solve_with_step_size(solve_on, t, step_size):
    """``solve_on(grid) -> (G, B, D)`` is the fixed-grid kernel call; returns h on ``t``."""
    if step_size is None:
        return solve_on(t)
    grid = fixed_grid(t, step_size)
    return read_outputs(solve_on(grid), grid, t)
