"""Device-resident batch source with the reference data generators' interface.

The training / evaluation loops ask a generator for ``get_split(fold, batch_size, chunk)`` and
``get_mini_batch(fold, batch_size)`` (reference ``dataloader.py:301-341``, real-data variant ``:441-489``) and read
``train_size / val_size / test_size / expert_dim / latent_dim``.  ``DeviceFolds`` answers those calls from tensors that
already live in HBM: the three folds are split ONCE (``split_sample``), every batch is produced by one device gather per
field (``index_select`` along the patient axis) and comes out CONTIGUOUS in the time-major layout the kernels read
(the reference returns strided views of the whole fold, which every consumer then has to copy), and a request for a whole
fold returns the stored tensor itself.  Nothing crosses PCIe per step except -- in the default, reference-compatible
index mode -- the ``batch_size`` int64 indices of a random minibatch, drawn with ``np.random.choice`` exactly like the
reference (``dataloader.py:297-299``) so that a seeded run sees the same minibatches; ``index_rng="device"`` draws them
with ``torch.randperm`` on the device instead.
"""

from __future__ import annotations

import numpy as np
import torch

FIELDS = ("measurements", "actions", "latents", "masks")


class DeviceFolds:
    def __init__(self, measurements, actions, latents, masks, val_size, test_size, statics=None, device=None,
                 expert_dim=4, index_rng="numpy"):
        if index_rng not in ("numpy", "device"):
            raise ValueError("index_rng must be 'numpy' (the reference's draw) or 'device'")
        self.device = torch.device(device) if device is not None else measurements.device
        put = lambda x: None if x is None else torch.as_tensor(x).to(device=self.device, dtype=torch.float32).contiguous()
        self.measurements, self.actions, self.latents, self.masks = put(measurements), put(actions), put(latents), put(masks)
        self.statics = put(statics)
        self.fields = FIELDS + (("statics",) if statics is not None else ())
        self.time_dim, self.n_sample, self.obs_dim = self.measurements.shape
        self.action_dim, self.latent_dim = self.actions.shape[2], self.latents.shape[2]
        self.expert_dim = int(expert_dim)
        self.val_size, self.test_size = int(val_size), int(test_size)
        self.train_size = int(self.n_sample - self.val_size - self.test_size)
        self.index_rng = index_rng
        self.split_sample()

    # ---- construction helpers --------------------------------------------------------------------------------------
    @classmethod
    def from_generator(cls, dg, device, index_rng="numpy"):
        """Adopt the tensors of a reference-style generator (``DataGeneratorRoche`` / ``DataGeneratorReal``)."""
        return cls(dg.measurements, dg.actions, dg.latents, dg.masks, dg.val_size, dg.test_size,
                   statics=getattr(dg, "statics", None), device=device, expert_dim=getattr(dg, "expert_dim", 4),
                   index_rng=index_rng)

    @classmethod
    def synthetic(cls, n_sample, time_dim, obs_dim, latent_dim, val_size, test_size, device, seed=666, step=0.125,
                  dose_max=10.0, p_remove=0.5):
        """Folds of the benchmark's distribution generated ON the device: z-scored Gaussian measurements,
        Bernoulli(1 - p_remove) masks, one dose per patient at a grid index ~ U{0..T-2} with amount ~ U(0, dose_max),
        Exponential(rate 100) initial latents in ``latents[0]``."""
        dev = torch.device(device)
        gen = torch.Generator(device=dev).manual_seed(seed)
        x = torch.randn(time_dim, n_sample, obs_dim, device=dev, generator=gen)
        mask = (torch.rand(time_dim, n_sample, obs_dim, device=dev, generator=gen) > p_remove).float()
        idx = torch.randint(0, max(time_dim - 1, 1), (n_sample,), device=dev, generator=gen)
        amt = torch.rand(n_sample, device=dev, generator=gen) * dose_max + 1e-3
        a = torch.zeros(time_dim, n_sample, 1, device=dev)
        a[idx, torch.arange(n_sample, device=dev), 0] = amt
        lat = torch.zeros(time_dim, n_sample, latent_dim, device=dev)
        lat[0] = torch.empty(n_sample, latent_dim, device=dev).exponential_(100.0, generator=gen)
        return cls(x, a, lat, mask, val_size, test_size, device=dev)

    # ---- the reference generator's surface ---------------------------------------------------------------------------
    def split_sample(self):
        tr, va = self.train_size, self.val_size
        cut = lambda lo, hi: {k: getattr(self, k)[:, lo:hi].contiguous() for k in self.fields}
        self.data_train, self.data_val, self.data_test = cut(0, tr), cut(tr, tr + va), cut(tr + va, self.n_sample)
        self._schedules = {}

    # The dose schedule of a fold (dosage, grid indices of the doses), derived ONCE: ``RocheODE.set_action`` needs the dose
    # count on the host, i.e. host synchronisations in the middle of every training step; a batch's schedule is a gather of
    # the fold's.  Attached to the batch's action tensor as ``hode_schedule`` (model.RocheODE.set_action reads it).  Folds
    # whose patients have unequal dose counts get none (set_action then raises like the reference).
    def _schedule(self, fold):
        if fold not in self._schedules:
            a = self._fold(fold)["actions"]
            try:
                from .solver import dose_schedule_index
                self._schedules[fold] = dose_schedule_index(a) if a.shape[2] == 1 and a.shape[1] > 0 else None
            except RuntimeError:
                self._schedules[fold] = None
        return self._schedules[fold]

    def _attach(self, batch, fold, pick):
        sched = self._schedule(fold)
        if sched is not None:
            batch["actions"].hode_schedule = (pick(sched[0]), pick(sched[1]))
        return batch

    def set_device(self, device):
        if torch.device(device) != self.device:
            self.device = torch.device(device)
            for k in self.fields:
                setattr(self, k, getattr(self, k).to(self.device))
            self.split_sample()

    def set_train_size(self, n_sample):
        """``DataGeneratorRoche.set_train_size`` (dataloader.py:82-89): n_sample counts all three folds."""
        self.train_size = int(n_sample - self.val_size - self.test_size)
        self.n_sample = int(n_sample)
        print("train_size", self.train_size)
        print("n_sample", self.n_sample)
        self.data_train = {k: v[:, :self.train_size].contiguous() for k, v in self.data_train.items()}
        self._schedules.pop("train", None)

    def set_val_size(self, n_val):
        self.val_size = int(n_val)
        self.data_val = {k: v[:, :n_val].contiguous() for k, v in self.data_val.items()}
        self._schedules.pop("val", None)

    def _fold(self, fold):
        assert fold in ("train", "val", "test")
        return {"train": self.data_train, "val": self.data_val, "test": self.data_test}[fold]

    def _get_index_random(self, N, k):
        if self.index_rng == "device":
            return torch.randperm(N, device=self.device)[:k]
        return torch.as_tensor(np.random.choice(N, k, replace=False)).to(device=self.device, dtype=torch.int64)

    def get_mini_batch(self, fold, batch_size):
        data = self._fold(fold)
        idx = self._get_index_random(data["measurements"].shape[1], batch_size)
        return self._attach({k: v.index_select(1, idx) for k, v in data.items()}, fold, lambda x: x.index_select(0, idx))

    def get_split(self, fold, batch_size, chunk=0):
        data = self._fold(fold)
        lo, hi = chunk * batch_size, (chunk + 1) * batch_size
        n = data["measurements"].shape[1]
        if lo == 0 and hi >= n:
            return self._attach(dict(data), fold, lambda x: x)  # the whole fold: the resident tensors themselves
        return self._attach({k: v[:, lo:hi].contiguous() for k, v in data.items()}, fold, lambda x: x[lo:hi])
