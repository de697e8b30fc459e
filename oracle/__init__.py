"""CPU oracle for the hybrid-ODE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This package restates, in plain PyTorch-CPU eager ops, the arithmetic of the
reference's latent-ODE forward / autograd-backward path:

* ``oracle.rhs``      right-hand sides     (reference ``model.py:446-555``, ``:969-1026``, ``:570-657``)
* ``oracle.solvers``  ``odeint`` semantics (third-party ``torchdiffeq==0.2.2``, called at
                      reference ``model.py:837,842,1116``; NOT present in the container,
                      restated from its published algorithm)
* ``oracle.encoder``  masked reverse-time LSTM encoder (reference ``model.py:383-440``)
* ``oracle.vi``       loss assembly        (reference ``model.py:1150-1214``)
* ``oracle.evalmetrics``  ensemble CRPS (third-party ``properscoring``, absent: parity unpinned) and the per-chunk
                      pieces of ``training_utils.evaluate`` (reference ``training_utils.py:100-201``)

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / timed CPU baseline only.
Nothing under ``hybrid-ode-neurips-2021_amd/`` imports it; the product path raises
when the HIP library is missing instead of falling back to this code.

Parity status
-------------
* rhs / encoder / loss pieces: PINNED by golden vectors captured from the imported
  reference (``tests/golden/*.npz``, generator ``tests/golden/make_golden.py``).
* solver boundary (``torchdiffeq.odeint``): the dependency is absent and the reference
  holds no tests or golden vectors for it => **parity unpinned** at that boundary.
  It is anchored instead by (1) scipy's Dormand-Prince tableau / single-step results,
  (2) analytic ODE convergence orders, (3) tableau identities, (4) the generator's
  LSODA latents as a loose known answer.  See ``tests/test_oracle_solvers.py``.
"""

from . import rhs, solvers  # noqa: F401
