"""Experiment constants, same names / fields / defaults as reference ``sim_config.py`` (values only, restated)."""
from collections import namedtuple


def _record(name, **defaults):
    return namedtuple(name, list(defaults), defaults=list(defaults.values()))


# 13 expert rate constants in the order RocheODE creates its parameters; Hill exponents 2, everything else 1
RochConfig = _record(
    "RochConfig",
    HillCure=2, HillPatho=2, ec50_patho=1, emax_patho=1, k_dexa=1, k_discure_immunereact=1, k_discure_immunity=1,
    k_disprog=1, k_immune_disease=1, k_immune_feedback=1, k_immune_off=1, k_immunity=1, kel=1,
)

DataConfig = _record(
    "DataConfig",
    n_sample=1000, obs_dim=20, latent_dim=6, action_dim=1, t_max=14, step_size=1, sparsity=0.5, output_sparsity=0.5,
    output_sigma=0.1, dose_max=1, p_remove=0.5,
)

dim8_config = DataConfig(obs_dim=40, latent_dim=8, output_sparsity=1 - 0.375, output_sigma=0.2, dose_max=10)
dim12_config = DataConfig(obs_dim=80, latent_dim=12, output_sparsity=1 - 0.25, output_sigma=0.2, dose_max=10)

ModelConfig = _record("ModelConfig", encoder_latent_ratio=2.0, expert_only=False, neural_ode=False, path="model/")

OptimConfig = _record(
    "OptimConfig",
    lr=0.01, ode_method="dopri5", niters=400, batch_size=50, test_freq=10, shuffle=True, n_restart=5, early_stop=10,
)

EvalConfig = _record("EvalConfig", t0=5)
