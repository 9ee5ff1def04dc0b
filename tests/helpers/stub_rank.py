"""A stand-in for a bench.py rank, started by bench.launch_ranks under torch.distributed.run in tests/test_bench_launcher.py:
joins a gloo group (so the rendezvous the launcher set up is really used), then behaves as STUB_MODE says.  No GPU."""
import json
import os
import sys
import time

import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = os.environ.get("STUB_MODE", "ok")
dist.init_process_group("gloo")
seen = [None] * world
dist.all_gather_object(seen, rank)
if mode == "fail" and rank == world - 1:
    print("stub rank %d: giving up on purpose" % rank, file=sys.stderr, flush=True)
    sys.exit(3)
if mode == "hang":
    time.sleep(600)
if rank == 0:
    print("some chatter on stdout that is not the result")
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "ranks_seen": seen, "argv": sys.argv[1:]}), flush=True)
dist.barrier()
dist.destroy_process_group()
