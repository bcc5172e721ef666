from . import dycon_losses, losses, ramps  # noqa: F401
