"""A stand-in for a bench.py rank, started by bench.launch_ranks under torch.distributed.run in tests/test_bench_launcher.py:
joins a gloo group (so the rendezvous the launcher set up is really used), then behaves as STUB_MODE says.  No GPU."""
import json
import os
import sys
import time

import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = os.environ.get("STUB_MODE", "ok")
bench.set_phase("start")
dist.init_process_group("gloo")
seen = [None] * world
dist.all_gather_object(seen, rank)
if mode.startswith("die:") or mode.startswith("hang:"):
    # the last rank goes through bench.py's phases and dies (or stops answering) in the named one; the others wait in the barrier below
    what, where = mode.split(":")
    for ph in bench.PHASES:
        bench.set_phase(ph, **({"mine": {"rank": rank, "ms_per_step": 0.5}} if ph in ("parity", "done") else {}))
        if rank == world - 1 and ph == where:
            if what == "die":
                print("stub rank %d: dying in phase %s" % (rank, ph), file=sys.stderr, flush=True)
                sys.exit(3)
            time.sleep(600)
if mode == "fail" and rank == world - 1:
    print("stub rank %d: giving up on purpose" % rank, file=sys.stderr, flush=True)
    sys.exit(3)
if mode == "hang":
    time.sleep(600)
if rank == 0:
    print("some chatter on stdout that is not the result")
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "ranks_seen": seen, "argv": sys.argv[1:]}), flush=True)
bench.set_phase("done")
dist.barrier()
dist.destroy_process_group()
