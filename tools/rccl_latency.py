#!/usr/bin/env python3
"""Latency of a halo exchange and of a scalar all-reduce as the solver issues them (gmg_comm_latency_probe), on whatever
communicator is available: 1 rank on the development boxes (RCCL's self send/recv: launch + proxy floor, no xGMI hop), N ranks
under torch.distributed.run on a multi-GPU node.  One JSON line; DESIGN.md section 5's model uses it.

    python tools/rccl_latency.py [--reps 200]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/rccl_latency.py"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import torch
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    pkg = entry.import_package()
    from gridapsolvers_jl_amd import multigpu as mg
    abi = pkg.abi
    lib = abi.load()
    h = C.c_void_p()
    abi.check(None, lib.gmg_create(C.byref(h), 2, local))
    path = mg.rccl_path().encode()
    uid = C.create_string_buffer(128)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        if rank == 0:
            abi.check(None, lib.gmg_comm_unique_id(path, uid))
        blob = [bytes(uid.raw) if rank == 0 else None]
        dist.broadcast_object_list(blob, src=0)
        uidb = blob[0]
    else:
        abi.check(None, lib.gmg_comm_unique_id(path, uid))
        uidb = bytes(uid.raw)
    abi.check(h, lib.gmg_comm_init_rccl(h, path, uidb, rank, world))
    rows = []
    # message shapes of BASELINE config 4 on 2x2x2 GPUs: 3 faces + 3 edges + 1 corner = 7 neighbours;
    # face of the 288^3 / 144^3 / 72^3 / 36^3 per-GPU levels at halo depth 1 and 5
    for nmsg, count, what in [(1, 1, "1 message x 1 double"), (7, 1, "7 messages x 1 double"),
                              (7, 36 * 36, "7 x 36^2 (level 3 face, depth 1)"), (7, 72 * 72, "7 x 72^2 (level 2 face, depth 1)"),
                              (7, 5 * 72 * 72, "7 x 5*72^2 (level 2, depth 5)"), (7, 144 * 144, "7 x 144^2 (level 1 face, depth 1)"),
                              (7, 5 * 144 * 144, "7 x 5*144^2 (level 1, depth 5)"), (7, 288 * 288, "7 x 288^2 (level 0 face)"),
                              (3, 288 * 288, "3 x 288^2 (faces only)")]:
        out = (C.c_double * 6)()
        abi.check(h, lib.gmg_comm_latency_probe(h, nmsg, count, args.reps, out))
        rows.append(dict(shape=what, nmsg=nmsg, doubles=count, exchange_us=out[0], exchange_host_enqueue_us=out[1],
                         allreduce1_us=out[2], allreduce1_host_enqueue_us=out[3], exchange_in_stream_us=out[4], empty_kernel_us=out[5]))
    lib.gmg_destroy(h)
    if rank == 0:
        print(json.dumps(dict(world=world, reps=args.reps, note="stream times per operation incl. a dependent 1-block kernel; "
                              "world=1: RCCL self send/recv (no xGMI hop)", rows=rows)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
