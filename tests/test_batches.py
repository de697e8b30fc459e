"""hode.batches.DeviceFolds against a literal restatement of the reference generators' slicing (dataloader.py:272-341)."""
import numpy as np
import pytest
import torch

from hode.batches import DeviceFolds


def _raw(T=7, N=23, obs=5, D=6, statics=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    d = {"measurements": torch.randn(T, N, obs, generator=g), "actions": torch.rand(T, N, 1, generator=g),
         "latents": torch.randn(T, N, D, generator=g), "masks": (torch.rand(T, N, obs, generator=g) < 0.5).float()}
    if statics:
        d["statics"] = torch.rand(T, N, 3, generator=g)
    return d


def _reference_folds(d, val, test):
    n = d["measurements"].shape[1]
    tr = n - val - test
    return ({k: v[:, :tr] for k, v in d.items()}, {k: v[:, tr:tr + val] for k, v in d.items()},
            {k: v[:, tr + val:] for k, v in d.items()})


@pytest.mark.parametrize("statics", [False, True])
def test_splits_and_minibatches_equal_the_reference_slicing(statics):
    d = _raw(statics=statics)
    f = DeviceFolds(d["measurements"], d["actions"], d["latents"], d["masks"], 5, 4, statics=d.get("statics"), device="cpu")
    tr, va, te = _reference_folds(d, 5, 4)
    assert (f.train_size, f.val_size, f.test_size, f.expert_dim, f.latent_dim) == (14, 5, 4, 4, 6)
    for fold, ref in (("train", tr), ("val", va), ("test", te)):
        for bs in (2, 3):
            for chunk in range(ref["measurements"].shape[1] // bs):
                got = f.get_split(fold, bs, chunk)
                assert set(got) == set(ref)
                for k in ref:
                    assert torch.equal(got[k], ref[k][:, chunk * bs:(chunk + 1) * bs]) and got[k].is_contiguous()
    whole = f.get_split("val", 5, 0)
    assert whole["masks"].data_ptr() == f.data_val["masks"].data_ptr()  # a whole fold is handed out without a copy
    # random minibatches: the reference draws np.random.choice(N, k, replace=False) (dataloader.py:297-299)
    np.random.seed(3)
    got = [f.get_mini_batch("train", 6) for _ in range(3)]
    np.random.seed(3)
    for g in got:
        idx = torch.tensor(np.random.choice(14, 6, replace=False), dtype=torch.int64)
        for k in tr:
            assert torch.equal(g[k], tr[k][:, idx, :]) and g[k].is_contiguous()


def test_set_train_size_device_index_mode_and_generator_adapter():
    d = _raw(N=30)
    f = DeviceFolds(d["measurements"], d["actions"], d["latents"], d["masks"], 5, 5, device="cpu", index_rng="device")
    f.set_train_size(17)  # counts all folds, like DataGeneratorRoche.set_train_size
    assert f.train_size == 7 and f.get_split("train", 7, 0)["latents"].shape == (7, 7, 6)
    b = f.get_mini_batch("train", 5)
    cols = {tuple(b["latents"][:, i].flatten().tolist()) for i in range(5)}
    assert len(cols) == 5  # without replacement
    assert all(any(torch.equal(b["latents"][:, i], d["latents"][:, j]) for j in range(7)) for i in range(5))

    class _Dg:  # what a pickled reference generator exposes
        measurements, actions, latents, masks = d["measurements"], d["actions"], d["latents"], d["masks"]
        val_size, test_size, expert_dim = 4, 6, 4

    g = DeviceFolds.from_generator(_Dg(), "cpu")
    assert g.train_size == 20 and torch.equal(g.get_split("test", 6, 0)["measurements"], d["measurements"][:, 24:])


def test_synthetic_folds_have_the_benchmark_distribution():
    f = DeviceFolds.synthetic(400, 20, 8, 12, 50, 50, "cpu", seed=1)
    b = f.get_split("train", 100, 1)
    assert b["measurements"].shape == (20, 100, 8) and b["latents"].shape == (20, 100, 12)
    chan = f.actions[..., 0]
    assert bool(((chan != 0).sum(dim=0) == 1).all()) and bool((chan[-1] == 0).all()) and float(chan.max()) <= 10.01
    assert 0.4 < float(f.masks.mean()) < 0.6 and abs(float(f.latents[0].mean()) - 0.01) < 2e-3


def test_batches_carry_their_dose_schedule_and_set_action_uses_it():
    """`DeviceFolds` derives a fold's dose schedule once and attaches the batch's gather of it to the action tensor;
    `RocheODE.set_action` then needs no host synchronisation (reference model.py:495-507 loops over the patients).  The
    attached schedule must be what set_action derives from the tensor itself, for random minibatches, fixed chunks and
    the whole fold; a tensor analysed before and not written since is served from the identity cache, a written one is not."""
    import model
    from hode.batches import DeviceFolds
    cpu = torch.device("cpu")
    folds = DeviceFolds.synthetic(60, 12, 6, 8, 10, 10, cpu, seed=3)
    ode = model.RocheODE(8, 1, 11 * 0.125, 0.125, device=cpu)
    np.random.seed(1)
    for batch in (folds.get_mini_batch("train", 16), folds.get_split("train", 16, 1), folds.get_split("val", 10, 0)):
        a = batch["actions"]
        assert hasattr(a, "hode_schedule")
        ode.set_action(a)
        dosage, times = ode.dosage.clone(), ode.times.clone()
        ode.set_action(a.clone())            # a plain tensor: the reference's derivation
        assert torch.equal(ode.dosage, dosage) and torch.equal(ode.times, times) and times.dtype == torch.float32
    # identity cache
    a = folds.get_split("train", 16, 0)["actions"].clone()
    calls = {"n": 0}
    inner = model.hode.solver.dose_schedule_index
    def counted(x):
        calls["n"] += 1
        return inner(x)
    model.dose_schedule_index = counted
    try:
        ode.set_action(a); ode.set_action(a)
        assert calls["n"] == 1
        b2 = a.clone()                      # an equal tensor that is another object (e.g. a recycled address): not served
        ode.set_action(b2)
        assert calls["n"] == 2
        ode.set_action(a)
        calls["n"] = 1
        ode.set_action(a)                   # (a is cached again now)
        assert calls["n"] == 1
        t_dose = int(a[:, 0, 0].nonzero()[0])
        a[t_dose, 0, 0] *= 2.0              # in-place write bumps the version counter: the cache must not serve it
        ode.set_action(a)
        assert calls["n"] == 2 and float(ode.dosage[0]) == float(a[t_dose, 0, 0])
    finally:
        model.dose_schedule_index = inner


def test_shards_keep_the_dose_schedule():
    from hode.batches import DeviceFolds
    from hode.parallel import shard_batch
    folds = DeviceFolds.synthetic(40, 10, 4, 8, 8, 8, torch.device("cpu"), seed=5)
    batch = folds.get_split("train", 24, 0)
    for rank in (0, 1, 2):
        part = shard_batch(batch, rank, 3)
        dosage, idx = part["actions"].hode_schedule
        assert dosage.shape[0] == idx.shape[0] == part["actions"].shape[1] == 8
        assert torch.equal(dosage, part["actions"][..., 0].max(dim=0)[0])
