from csts_amd.cli import parse_args, load_config  # noqa: F401
