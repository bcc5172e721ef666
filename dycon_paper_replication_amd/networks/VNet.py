"""3D V-Net with the DyCON feature head, on the HIP executor.

Topology and state_dict keys follow code/networks/VNet.py:145-239 (normalization='groupnorm',
Dropout3d(0.5) at x5 and x9 when has_dropout); the projection head is the one of
code/networks/UNet3D_contrastive.py:261-267 applied to the bottleneck, so that ``forward`` returns
the 3-tuple the training step unpacks (code/train_DyCON_BraTS19.py:304).  The reference's own
``vnet`` factory path raises TypeError; this class is what it was meant to build.
"""
from ._base import HipSegNet


class VNet(HipSegNet):
    net_type = "vnet"

    def __init__(self, n_channels=1, n_classes=2, n_filters=16, normalization="groupnorm", has_dropout=False,
                 scale_factor=2, **kw):
        if n_filters != 16:
            raise NotImplementedError("n_filters is fixed to 16 (GroupNorm(16, C) needs C % 16 == 0)")
        super().__init__(in_channels=n_channels, n_classes=n_classes, scale_factor=scale_factor,
                         normalization=normalization, has_dropout=has_dropout, **kw)
