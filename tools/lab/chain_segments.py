"""Where the data-parallel graph chain spends its time with one rank on RCCL: HIP-event segments (forward graph | eager losses |
backward head | backward trunks + exposed all-reduce | optimizer) and the wall-clock step, against the single graph."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from csts_amd import distributed as du
du._FORCE = True
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T
cut = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = load_yaml("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 4, "MODEL.LOSS_FUNC", "kldiv+egonce",
                                                                  "MODEL.LOSS_ALPHA", 0.05, "DATA.NUM_FRAMES", 16, "CSTS_AMD.COMPUTE", "bf16",
                                                                  "CSTS_AMD.TRUNK_CUT", cut])
torch.manual_seed(cfg.RNG_SEED)
model = build_model(cfg)
if not isinstance(model, du.GradAllReduce):
    model = du.GradAllReduce(model, bucket_mb=64)
model.train()
batch = T.synthetic_batch(4, 16, 256, 1000, dev)
opt = T.construct_optimizer(model, cfg, capturable=True)
seg = T.SegmentedTrainStep(cfg, model, opt, batch)
for _ in range(5):
    seg.run(batch, 1e-4)
rows, wall = [], []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seg.run(batch, 1e-4, timed=True)
    torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) * 1e3)
    rows.append(seg.segment_ms())
med = [statistics.median(r[i] for r in rows) for i in range(5)]
print(f"trunk_cut {cut}: fwd {med[0]:.3f}  loss {med[1]:.3f}  bwd_head {med[2]:.3f}  bwd_trunks+AR {med[3]:.3f}  opt {med[4]:.3f}  sum {sum(med):.3f}  wall {statistics.median(wall):.3f} ms")
torch.distributed.destroy_process_group()
