from csts_amd.losses import KLDiv, EgoNCE, get_loss_func  # noqa: F401
