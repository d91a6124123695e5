from .build import MODEL_REGISTRY, build_model  # noqa: F401
from .custom_multimodal_builder import CSTS  # noqa: F401
