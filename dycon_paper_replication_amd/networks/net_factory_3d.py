"""Model factory with the reference's signature (code/networks/net_factory_3d.py:5-12)."""
import torch

from .UNet3D_contrastive import UNet3D
from .VNet import VNet


def net_factory_3d(net_type="unet_3D", in_chns=1, class_num=2, scaler=4, use_aspp=False, dtype=torch.float32,
                   normalization="groupnorm"):
    """Returns an nn.Module whose forward gives (tanh_map, logits, features).

    ``dtype`` selects the activation storage of the HIP kernels (fp32 parity mode or bf16);
    ``normalization`` applies to the V-Net only (the reference factory could not pass one)."""
    if net_type == "unet_3D":
        return UNet3D(in_channels=in_chns, n_classes=class_num, scale_factor=scaler, use_aspp=use_aspp, dtype=dtype)
    if net_type == "vnet":
        return VNet(n_channels=in_chns, n_classes=class_num, scale_factor=scaler, has_dropout=True,
                    normalization=normalization, dtype=dtype)
    return None
