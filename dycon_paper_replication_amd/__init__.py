"""MI355X-native DyCON training step (drop-in for the hot path of rogeliorjr/DyCON_Paper_Replication).

Host side: Python mirroring the reference's callables (``networks.net_factory_3d``,
``utils.dycon_losses.UnCLoss/FeCLoss``, ``utils.losses``, ``utils.ramps``) and its training step
(``trainer.DyconTrainer`` == code/train_DyCON_BraTS19.py:298-372).  Device side: hand-written HIP
kernels for gfx950 in ``libdycon_hip.so`` behind the C ABI of ``include/dycon_hip.h``.
There is no CPU / PyTorch fallback: ops raise if the library is not built.
"""
__version__ = "0.1.0"

