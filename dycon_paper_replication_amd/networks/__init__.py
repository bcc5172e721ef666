from .net_factory_3d import net_factory_3d  # noqa: F401
from .UNet3D_contrastive import UNet3D  # noqa: F401
from .VNet import VNet  # noqa: F401
