import os, sys
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from sift_amd.sharding import gather_rows, shard_videos, clip_scores
rows = np.arange(3 * 5 * 6, dtype=np.float64).reshape(3, 5, 6)
allr = gather_rows(rows, 3, 5, device="cuda")
print("gathered", allr.shape, np.array_equal(allr, rows))
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); print(t.item())
dist.barrier(); dist.destroy_process_group(); print("ok")
