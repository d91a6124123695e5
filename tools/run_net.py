#!/usr/bin/env python3
"""Entry point with the reference's command line (tools/run_net.py):
    python tools/run_net.py --cfg configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml NUM_GPUS 8 TRAIN.BATCH_SIZE 32 \
        MODEL.LOSS_FUNC kldiv+egonce MODEL.LOSS_ALPHA 0.05
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from csts_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
