"""MI355X-native DyCON training step (drop-in for the hot path of rogeliorjr/DyCON_Paper_Replication).

Host side: Python mirroring the reference's callables (``networks.net_factory_3d``,
``utils.dycon_losses.UnCLoss/FeCLoss``, ``utils.losses``, ``utils.ramps``) and its training step
(``trainer.DyconTrainer`` == code/train_DyCON_BraTS19.py:298-372).  Device side: hand-written HIP
kernels for gfx950 in ``libdycon_hip.so`` behind the C ABI of ``include/dycon_hip.h``.
There is no CPU / PyTorch fallback: ops raise if the library is not built.
"""
__version__ = "0.1.0"

import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step uses four streams; the
# data-parallel run adds torch's collective stream and the bucket-issue stream, and two streams then SHARE a queue: the teacher
# forward and the weight-gradient launches serialise (profiles/r03_ddp_one_rank_trace.txt: 3 busy queues instead of 4, +0.28 ms/step
# -- the whole "exchange path" overhead a rank paid besides wire time).  Read by the HIP runtime when it initialises: set before the
# first CUDA call of the process; an existing setting is respected.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
