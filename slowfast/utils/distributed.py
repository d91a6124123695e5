from csts_amd.distributed import all_gather_with_grad, all_gather, all_reduce  # noqa: F401
