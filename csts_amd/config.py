"""Minimal config system reading the reference's YAML schema (yacs/fvcore CfgNode in the reference:
slowfast/config/defaults.py:12-970, custom_config.py:7-24).  Only the keys the CSTS hot path and its
thin driver read are given defaults; unknown keys found in a YAML are accepted and kept,
so the reference YAMLs load unchanged.  ``CSTS_AMD.*`` are this implementation's own switches."""
from __future__ import annotations

import ast
import copy
from typing import Any, Sequence

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _coerce(v):
        if isinstance(v, str):
            try:
                v = ast.literal_eval(v)      # YAMLs carry strings like "(3, 7, 7)"
            except Exception:
                pass
        if isinstance(v, tuple):
            v = list(v)
        return v

    def _merge(self, d: dict):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v)

    def merge_from_file(self, path: str):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts: Sequence[Any]):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            node[parts[-1]] = self._coerce(v)

    def dump(self) -> str:
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self))


def _node(**kw):
    n = CfgNode()
    for k, v in kw.items():
        n[k] = v
    return n


def get_cfg() -> CfgNode:
    """Defaults for the keys of the path (values = reference defaults, slowfast/config/defaults.py)."""
    c = CfgNode()
    c.TRAIN = _node(ENABLE=True, DATASET="ego4d_av_gaze_forecast", BATCH_SIZE=64, EVAL_PERIOD=10, CHECKPOINT_PERIOD=10,
                    AUTO_RESUME=True, CHECKPOINT_FILE_PATH="", CHECKPOINT_EPOCH_RESET=False, MIXED_PRECISION=False,
                    AUDIO_CHECKPOINT_FILE_PATH="")
    c.TEST = _node(ENABLE=True, DATASET="ego4d_av_gaze_forecast", BATCH_SIZE=8, NUM_ENSEMBLE_VIEWS=10, NUM_SPATIAL_CROPS=3,
                   CHECKPOINT_FILE_PATH="")
    c.DATA = _node(PATH_PREFIX="", NUM_FRAMES=8, SAMPLING_RATE=8, TRAIN_JITTER_SCALES=[256, 320], TRAIN_CROP_SIZE=224,
                   TEST_CROP_SIZE=256, INPUT_CHANNEL_NUM=[3, 3], TARGET_FPS=30, USE_OFFSET_SAMPLING=False,
                   GAUSSIAN_KERNEL=19, MEAN=[0.45, 0.45, 0.45], STD=[0.225, 0.225, 0.225])
    c.MVIT = _node(MODE="conv", POOL_FIRST=False, CLS_EMBED_ON=True, PATCH_KERNEL=[3, 7, 7], PATCH_STRIDE=[2, 4, 4],
                   PATCH_PADDING=[2, 4, 4], PATCH_2D=False, EMBED_DIM=96, NUM_HEADS=1, MLP_RATIO=4.0, QKV_BIAS=True,
                   DROPPATH_RATE=0.1, DEPTH=16, NORM="layernorm", DIM_MUL=[], HEAD_MUL=[], POOL_KV_STRIDE=None,
                   POOL_KV_STRIDE_ADAPTIVE=None, POOL_Q_STRIDE=[], POOL_KVQ_KERNEL=None, ZERO_DECAY_POS_CLS=True,
                   NORM_STEM=False, SEP_POS_EMBED=False, DROPOUT_RATE=0.0, AUDIO_BRANCH_ON=True, SPATIAL_AUDIO_ATTN=False)
    c.MODEL = _node(ARCH="mvit", MODEL_NAME="CSTS", NUM_CLASSES=400, LOSS_FUNC="kldiv", LOSS_ALPHA=1.0, DROPOUT_RATE=0.5,
                    ACT_CHECKPOINT=False)
    c.SOLVER = _node(BASE_LR=0.1, LR_POLICY="cosine", COSINE_END_LR=0.0, MAX_EPOCH=300, MOMENTUM=0.9, DAMPENING=0.0,
                     NESTEROV=True, WEIGHT_DECAY=1e-4, WARMUP_EPOCHS=0.0, WARMUP_START_LR=0.01, OPTIMIZING_METHOD="sgd",
                     BASE_LR_SCALE_NUM_SHARDS=False, COSINE_AFTER_WARMUP=False, ZERO_WD_1D_PARAM=False,
                     CLIP_GRAD_VAL=None, CLIP_GRAD_L2NORM=None)
    c.BN = _node(USE_PRECISE_STATS=False, NUM_BATCHES_PRECISE=200)
    c.DATA_LOADER = _node(NUM_WORKERS=8, PIN_MEMORY=True, RETURN_TARGET_FRAME=False)
    c.TENSORBOARD = _node(ENABLE=False)
    c.NUM_GPUS = 1
    c.NUM_SHARDS = 1
    c.SHARD_ID = 0
    c.OUTPUT_DIR = "."
    c.RNG_SEED = 1
    c.LOG_PERIOD = 10
    c.LOG_MODEL_INFO = True
    c.DIST_BACKEND = "nccl"          # == RCCL on ROCm
    # this implementation's switches
    c.CSTS_AMD = _node(COMPUTE="auto",               # "fp32": exact-fp32 MFMA parity mode ; "bf16": throughput mode ; "fp16": the reference's autocast arithmetic + dynamic loss scaling ; "auto": by TRAIN.MIXED_PRECISION (model.resolve_compute)
                       SYNTHETIC_DATA=True,          # the data pipeline is out of scope: synthetic clips
                       STEPS_PER_EPOCH=50,
                       EPOCHS_THIS_RUN=0,            # > 0: stop this invocation after that many epochs (pre-emption; the next one auto-resumes)
                       EVAL_STEPS=2,                 # synthetic validation iterations of the periodic eval pass (TRAIN.EVAL_PERIOD)
                       GRAD_BUCKET_MB=64,
                       FACTORED_ADAMW=True,          # 16-bit compute modes: FusedAdamW forms the fusion convs' weight gradient dY^T A on the matrix cores inside their update, from the rank-(B T') factors (never written / re-read: 453 MB per step; -0.3 ms at b = 4)
                       FUSION_GRAD_FACTORS=True,     # data-parallel chain: the ranks all-gather the rank-(B*T') factors of the three fusion-conv weight gradients (3.2 MB each) instead of all-reducing 151 MB each
                       GRAD_BUCKET_DTYPE="fp32",     # "bf16" / "fp16": data-parallel gradient buckets travel in the library's 16-bit type (half the xGMI bytes), fp32 accumulation in the optimizer
                       TRUNK_CUT=3,                  # data-parallel graph chain: second autograd cut in front of this video block (0 = trunks in one piece)
                       TWO_STREAMS=True,             # audio trunk on a second HIP stream, concurrent with the video trunk
                       SAVE_CHECKPOINTS=False,       # write checkpoints/checkpoint_epoch_XXXXX.pyth (reference wire format) every CHECKPOINT_PERIOD
                       HIP_GRAPH=True,               # single GPU: capture the whole iteration once, replay it (train.GraphedTrainStep)
                       FUSION_KERNEL_FROM_GRID=False)  # True: (1,S/32,S/32) fusion kernels -> 224^2 works (parity unpinned)
    return c


def assert_and_infer_cfg(cfg: CfgNode) -> CfgNode:
    """The checks of slowfast/config/defaults.py:945-970 that concern this path."""
    assert cfg.NUM_GPUS == 0 or cfg.TRAIN.BATCH_SIZE % cfg.NUM_GPUS == 0
    assert cfg.NUM_GPUS == 0 or cfg.TEST.BATCH_SIZE % cfg.NUM_GPUS == 0
    if cfg.SOLVER.BASE_LR_SCALE_NUM_SHARDS:
        cfg.SOLVER.BASE_LR *= cfg.NUM_SHARDS
        cfg.SOLVER.WARMUP_START_LR *= cfg.NUM_SHARDS
        cfg.SOLVER.COSINE_END_LR *= cfg.NUM_SHARDS
    assert cfg.SHARD_ID < cfg.NUM_SHARDS
    return cfg


def load_yaml(path: str, opts: Sequence[Any] = ()) -> CfgNode:
    cfg = get_cfg()
    cfg.merge_from_file(path)
    if opts:
        cfg.merge_from_list(list(opts))
    return cfg
