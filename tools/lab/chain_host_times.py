"""Host time of every stage of SegmentedTrainStep.run (CSTS_CHAIN_HOST_TIMES=1) in the 1-rank rehearsal: is the host ahead of the device?
usage: python tools/lab/chain_host_times.py"""
import os, sys, time
os.environ["CSTS_CHAIN_HOST_TIMES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29677")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from csts_amd import distributed as du
du._FORCE = True
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T
from csts_amd.distributed import GradAllReduce
cfg = load_yaml(os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"), ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 4, "MODEL.LOSS_FUNC", "kldiv+egonce", "MODEL.LOSS_ALPHA", 0.05,
                "DATA.NUM_FRAMES", 16, "CSTS_AMD.COMPUTE", "bf16", "CSTS_AMD.GRAD_BUCKET_DTYPE", "bf16"])
torch.manual_seed(0)
model = GradAllReduce(build_model(cfg), bucket_mb=cfg.CSTS_AMD.GRAD_BUCKET_MB)
model.train(True)
batch = T.synthetic_batch(4, 16, 256, 1000, dev)
opt = T.construct_optimizer(model, cfg, capturable=True)
g = T.SegmentedTrainStep(cfg, model, opt, batch)
for _ in range(5):
    g.run(batch, 1e-4)
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter()
    g.run(batch, 1e-4)
    t1 = time.perf_counter()
    print(f"run() host {1e6 * (t1 - t0):7.0f} us   stages {g.host_us}", flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    g.run(batch, 1e-4)
torch.cuda.synchronize()
print(f"20 steps: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
