"""INTEGRATION.md option B, end to end: the reference's own ``experiments/run_simulation.py`` driven through THIS package's
``model`` / ``sim_config`` / ``training_utils`` / ``global_config`` mirrors (put first on ``sys.path``), with the reference's
``dataloader`` generating the data.  Runs only where /root/reference exists (this container; never on the GPU box).  The
solver and the CRPS kernel are swapped for their CPU oracles through the test-only hooks, so this checks the drop-in
SURFACE -- names, signatures, state-dict / checkpoint flow, evaluate() printout -- not the kernels."""
import importlib.util
import os
import pickle
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "experiments", "run_simulation.py")),
                                reason="reference tree not present")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_reference_run_simulation_runs_on_the_mirror(tmp_path, monkeypatch, capsys):
    import model
    import sim_config
    import training_utils
    from oracle.solvers import odeint as oracle_odeint
    from test_evaluate import oracle_ensemble_crps

    assert os.path.dirname(model.__file__).endswith("hybrid-ode-neurips-2021_amd")  # the mirror, not the reference
    monkeypatch.setattr(model.hode, "odeint", oracle_odeint)               # decoders pick this up at construction
    monkeypatch.setattr(training_utils, "_ensemble_crps", oracle_ensemble_crps)
    # the reference's generator (numpy / scipy LSODA) under its own module name, as its pickles expect
    for k in ("dataloader",):
        sys.modules.pop(k, None)
    monkeypatch.syspath_prepend(REF)  # after the mirror's directory, which conftest put first: mirrors still win
    sys.path.remove(REF)
    sys.path.append(REF)
    dataloader = _load("dataloader", os.path.join(REF, "dataloader.py"))
    cfg = sim_config.DataConfig(obs_dim=6, latent_dim=8, t_max=2, step_size=0.25, output_sigma=0.1, dose_max=5)
    np.random.seed(3)
    torch.manual_seed(3)
    cpu = torch.device("cpu")
    dg = dataloader.DataGeneratorRoche(40, cfg.obs_dim, cfg.t_max, cfg.step_size, sim_config.RochConfig(), cfg.output_sigma,
                                       dose_max=cfg.dose_max, latent_dim=cfg.latent_dim, sparsity=cfg.sparsity,
                                       output_sparsity=cfg.output_sparsity, val_size=10, test_size=10,
                                       p_remove=cfg.p_remove, device=cpu)
    dg.generate_data()
    dg.split_sample()
    data_path = tmp_path / "datafile.pkl"
    with open(data_path, "wb") as f:
        pickle.dump(dg, f)

    runsim = _load("ref_run_simulation", os.path.join(REF, "experiments", "run_simulation.py"))
    model_dir = str(tmp_path) + "/"
    runsim.run(seed=666, elbo=True, device="c", eval_only=False, init_path=None, data_path=str(data_path), sample=40,
               data_config=cfg, roche_config=sim_config.RochConfig(),
               model_config=sim_config.ModelConfig(path=model_dir),
               optim_config=sim_config.OptimConfig(ode_method="rk4", niters=4, batch_size=5, test_freq=2, n_restart=1,
                                                   early_stop=5),
               eval_config=sim_config.EvalConfig(t0=4))
    out = capsys.readouterr().out
    assert "Iter 0002 | Total Loss" in out and "Overall best loss" in out
    lines = [l for l in out.split("\n") if l.startswith(("rmse_z0,", "rmse_x,", "cprs_z0,", "cprs_x,"))]
    assert [l.split(",")[0] for l in lines] == ["rmse_z0", "rmse_x", "cprs_z0", "cprs_x"]
    assert all(np.isfinite(float(v)) for l in lines for v in l.split(",")[1:])
    names = [f for f in os.listdir(model_dir) if f.endswith(".pkl") and f != "datafile.pkl"]
    assert names, "no checkpoint written"
    best = torch.load(model_dir + names[0])
    assert {"encoder_state_dict", "decoder_state_dict", "best_loss"} <= set(best)
    assert "ode.ml_net.0.weight" in best["decoder_state_dict"] and "lstm.weight_ih_l0" in best["encoder_state_dict"]
