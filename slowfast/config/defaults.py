from csts_amd.config import get_cfg, assert_and_infer_cfg, CfgNode  # noqa: F401
