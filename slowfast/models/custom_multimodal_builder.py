from csts_amd.model import CSTS  # noqa: F401
