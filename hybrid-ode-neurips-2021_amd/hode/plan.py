"""Pre-planned solver launches: descriptors and buffers built once, then re-launched with no host allocation.

``RocheRKPlan`` is the launch-bound inner loop's answer on the host side: the fixed-grid forward + discrete-adjoint
pair for one (batch, grid, latent-dim) shape with every device buffer resident, so a step costs two C-ABI calls and
one small memset and can be captured in a HIP graph (``capture()``), since libhode neither allocates nor syncs.
"""

from __future__ import annotations

import torch

from . import _lib as L


class RocheRKPlan:
    def __init__(self, y0, theta, w, b, t, dosage, dose_times, method="rk4", ablate=False, perturb=False,
                 lanes_per_patient=0, need_theta_grad=True, tape=True):
        for x in (y0, theta, t, dosage, dose_times):
            if not x.is_cuda:
                raise L.HodeConfigError("hode: RocheRKPlan needs HIP-device tensors (no CPU fallback)")
        self.lib = L.lib()
        self.dev = y0.device
        B, D = y0.shape
        T = t.numel()
        self.B, self.D, self.T = B, D, T
        f32 = dict(device=self.dev, dtype=torch.float32)
        self.y0 = y0.detach().to(torch.float32).contiguous()
        self.theta = theta.detach().to(torch.float32).contiguous()
        self.w = None if w is None else w.detach().to(torch.float32).contiguous()
        self.b = None if b is None else b.detach().to(torch.float32).contiguous()
        self.t = t.detach().to(torch.float32).contiguous()
        self.dosage = dosage.detach().to(torch.float32).contiguous()
        self.dose_times = dose_times.detach().to(torch.float32).reshape(B, -1).contiguous()
        self.h = torch.empty((T, B, D), **f32)
        self.grad_h = torch.zeros((T, B, D), **f32)
        self.grad_y0 = torch.empty((B, D), **f32)
        # one flat accumulator for all parameter gradients: [w | b | theta] -> a single memset / all-reduce bucket
        M = D - 4
        self.n_w, self.n_b = M * D, M
        self.grad_flat = torch.zeros(self.n_w + self.n_b + L.N_THETA, **f32)
        self.grad_w = self.grad_flat[: self.n_w].view(M, D) if M > 0 else None
        self.grad_b = self.grad_flat[self.n_w: self.n_w + self.n_b] if M > 0 else None
        self.grad_theta = self.grad_flat[self.n_w + self.n_b:]
        d = L.new_solve_desc()
        d.rhs_kind = L.RHS_ROCHE_ABLATE if ablate else L.RHS_ROCHE
        d.method, d.perturb, d.batch, d.latent_dim, d.n_times = L.METHODS[method], int(perturb), B, D, T
        d.n_dose = self.dose_times.shape[1]
        d.lanes_per_patient = int(lanes_per_patient)
        d.need_theta_grad = int(need_theta_grad)
        d.t, d.y0, d.dosage, d.theta = self.t.data_ptr(), self.y0.data_ptr(), self.dosage.data_ptr(), self.theta.data_ptr()
        d.dose_times = self.dose_times.data_ptr() if d.n_dose else 0
        d.w1 = 0 if self.w is None else self.w.data_ptr()
        d.b1 = 0 if self.b is None else self.b.data_ptr()
        d.h = self.h.data_ptr()
        d.grad_h, d.grad_y0 = self.grad_h.data_ptr(), self.grad_y0.data_ptr()
        d.grad_w1 = 0 if self.grad_w is None else self.grad_w.data_ptr()
        d.grad_b1 = 0 if self.grad_b is None else self.grad_b.data_ptr()
        d.grad_theta = self.grad_theta.data_ptr()
        # one buffer for both launches: the backward's partial sums and (tape=True) the forward's stage tape
        self.base_flags = L.FLAG_OVERWRITE_GRADS | (L.FLAG_TAPE if tape else 0)
        d.flags = self.base_flags
        nbytes = max(self.lib.hode_workspace_bytes(d, L.WS_RK_BWD), self.lib.hode_workspace_bytes(d, L.WS_RK_FWD))
        self.ws = torch.empty(max(nbytes, 4), device=self.dev, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = self.ws.data_ptr(), nbytes
        self.desc = d
        self._graph = None

    # ---- algorithmic HBM bytes per launch (DESIGN.md "Roofline"): fp32, state reuse across the sweep
    @property
    def fwd_bytes(self):
        return self.B * (4 * self.T * self.D + 4 * self.D + 8)

    @property
    def bwd_bytes(self):
        return self.B * (8 * self.T * self.D + 4 * self.D)

    def forward(self):
        L.check(self.lib.hode_rk_fwd(self.desc, torch.cuda.current_stream().cuda_stream), "hode_rk_fwd")
        return self.h

    def backward(self):
        """Discrete adjoint for the cotangent currently in ``self.grad_h``; overwrites ``grad_flat``
        (HODE_FLAG_OVERWRITE_GRADS: the fold stores, so no memset of the bucket is needed)."""
        L.check(self.lib.hode_rk_bwd(self.desc, torch.cuda.current_stream().cuda_stream), "hode_rk_bwd")
        return self.grad_y0, self.grad_flat

    def backward_kernel_only(self):
        """The adjoint kernel alone (HODE_FLAG_SKIP_FOLD, no accumulator memset): what a profiler reports as one kernel."""
        self.desc.flags = self.base_flags | L.FLAG_SKIP_FOLD
        try:
            L.check(self.lib.hode_rk_bwd(self.desc, torch.cuda.current_stream().cuda_stream), "hode_rk_bwd")
        finally:
            self.desc.flags = self.base_flags

    def step(self):
        self.forward()
        return self.backward()

    def _point_grads_at(self, flat):
        """Make ``flat`` the gradient bucket the backward writes ([w | b | theta], same layout as ``grad_flat``)."""
        d = self.desc
        d.grad_w1 = flat.data_ptr() if self.n_w else 0
        d.grad_b1 = flat[self.n_w:].data_ptr() if self.n_b else 0
        d.grad_theta = flat[self.n_w + self.n_b:].data_ptr()

    def capture(self, n_buckets=1):
        """Capture forward + backward into one HIP graph; ``replay()`` then costs a single graph launch.

        ``n_buckets > 1`` captures one graph per gradient bucket (``self.buckets[i]``, ``replay(i)``): a data-parallel
        caller alternates them so that the all-reduce of one bucket can run while the next step fills the other."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.step()  # warm-up outside capture (module load, allocator)
        torch.cuda.current_stream().wait_stream(s)
        self.buckets = [self.grad_flat] + [torch.zeros_like(self.grad_flat) for _ in range(n_buckets - 1)]
        self._graphs = []
        for flat in self.buckets:
            self._point_grads_at(flat)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step()
            self._graphs.append(g)
        self._point_grads_at(self.grad_flat)
        self._graph = self._graphs[0]
        return self._graph

    def replay(self, bucket=0):
        self._graphs[bucket].replay()
