"""3D U-Net with projection head (the model the reference's run scripts train), on the HIP executor.

Topology and state_dict keys follow code/networks/UNet3D_contrastive.py:207-316 with feature_scale=4
(filters 16..256), InstanceNorm3d blocks (networks/utils.py:99-123), MaxPool3d(2), trilinear x2 +
concat decoder (networks/utils.py:260-276), Dropout(0.3) at the bottleneck and before the heads.
"""
from ._base import HipSegNet


class UNet3D(HipSegNet):
    net_type = "unet_3D"

    def __init__(self, in_channels=1, feature_scale=4, n_classes=2, scale_factor=2, use_aspp=False, **kw):
        if feature_scale != 4:
            raise NotImplementedError("feature_scale is fixed to 4 (filters 16, 32, 64, 128, 256)")
        if use_aspp:
            raise NotImplementedError("ASPP is never enabled by the reference's scripts (net_factory_3d.py:5)")
        super().__init__(in_channels=in_channels, n_classes=n_classes, scale_factor=scale_factor, has_dropout=True, **kw)
