from csts_amd.build import build_model  # noqa: F401
from csts_amd.registry import MODEL_REGISTRY  # noqa: F401
