"""Probe (GPU box): does RCCL accept two ranks of one job on the SAME device?  (If it does, the RCCL transport of art_mgpu_* can be
exercised with 2 ranks on the one-GPU box; if it refuses -- "Duplicate GPU detected" -- only 1-rank jobs can.)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 tools/rccl_two_ranks_one_gpu_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from araytracingjourney_amd import renderer, scenes

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
ids = [renderer.mgpu_unique_id() if rank == 0 else None]
dist.broadcast_object_list(ids, src=0)
sc = scenes.cornell()
r = renderer.renderer_for_scene(sc, (256, 256), shard=renderer.mgpu_shard(rank, world), frames_in_flight=2)
r.upload_state()
try:
    mg = renderer.MultiGpu(r, rank, world, unique_id=ids[0])
    for _ in range(4):
        mg.trace()
    mg.flush()
    print(f"rank {rank}: RCCL accepted {world} ranks on one device; gathers = {mg.counts()['gathers']}")
except Exception as e:
    print(f"rank {rank}: {e}")
dist.destroy_process_group()
