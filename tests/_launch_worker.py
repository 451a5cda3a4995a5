"""worker for test_launcher: one rank started by abacusutils_amd.launch.launch_ranks.  Reduces (rank + 1) over a gloo
group built from the environment the launcher exported; can be told to crash or to hang on one rank."""
import argparse
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fail-rank', type=int, default=-1)
    ap.add_argument('--hang-rank', type=int, default=-1)
    a = ap.parse_args()
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert os.environ['LOCAL_RANK'] == str(rank) and os.environ['ABACUS_RDZV_KEY']
    if rank == a.fail_rank:
        print('boom from rank', rank, file=sys.stderr)
        sys.exit(3)
    if rank == a.hang_rank:
        time.sleep(3600)
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method=f"tcp://{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}", rank=rank,
                            world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print('RESULT ' + json.dumps({'sum': int(t[0]), 'world': world, 'key': os.environ['ABACUS_RDZV_KEY']}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
