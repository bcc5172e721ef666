"""CPU oracle for the DyCON training step -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32/fp64 CPU restatement of the reference's
hot path (rogeliorjr/DyCON_Paper_Replication, code/train_DyCON_BraTS19.py:298-372
and the modules it calls).  It exists so that the HIP path can be checked on a
GPU box where the reference's own Python files are absent.

Rules:
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from here;
  * nothing under dycon_paper_replication_amd/ imports it -- the product path
    fails loudly if the HIP library is missing, it never falls back to this;
  * parity is PINNED: every function here is checked (tests/test_oracle_golden.py)
    against fixtures under tests/golden/ that were produced by importing the
    reference itself in the build container (tests/golden/make_golden.py).
"""
from . import nets, losses, step  # noqa: F401
