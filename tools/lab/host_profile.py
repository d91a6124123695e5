import os, sys, cProfile, pstats, time
sys.path.insert(0, "/root/repo")
import torch
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T
dev = torch.device("cuda:0")
cfg = load_yaml("/root/repo/configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 4, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", 16, "CSTS_AMD.COMPUTE", "bf16"])
m = build_model(cfg); m.train()
opt = T.construct_optimizer(m, cfg)
batch = T.synthetic_batch(4, 16, 256, 1000, dev)
for _ in range(3): T.train_step(cfg, m, batch, opt, 1e-4)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5): T.train_step(cfg, m, batch, opt, 1e-4)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step %.1f ms, drain %.1f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t1) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
