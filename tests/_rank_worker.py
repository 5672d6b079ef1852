"""Child process of the multi-rank GPU tests: one rank of a frame-parallel job on ONE device.

    python tests/_rank_worker.py <rank> <world> <port> <n_frames> <device> <out.npy> [<distinct> [<inflight>]]

Started as a fresh program (never a fork of a GPU-initialised process).  The rank builds the frames of its contiguous
block (batch.shard_range) -- frame i of the job is synthetic frame ``i % distinct`` (distinct = 0: frame i) --, runs the
full pipe on them through its own BatchDetector, takes part in the gloo gather of the 48-byte records (no data-path
collective) and rank 0 saves the gathered array.  With ``distinct`` > 0 the block is device-resident (a BASELINE
configs[3] share: 8 192 frames over 8 GPUs = 1 024 per GPU), otherwise the frames are handed over as host buffers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, n, device = (int(x) for x in sys.argv[1:6])
    out = sys.argv[6]
    distinct = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    inflight = int(sys.argv[8]) if len(sys.argv) > 8 else 4           # 4 slots: several chunks per rank
    import numpy as np
    import torch.distributed as dist
    from lfd_amd import _native, batch, synth
    from lfd_amd.detecttrails import default_params
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb, pd, prs = default_params()
    rs = _native.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    a, b = batch.shard_range(n, rank, world)
    det = batch.BatchDetector(device, synth.SDSS_SHAPE, inflight=inflight)
    if distinct:
        import torch
        base, cats = synth.make_frames(0, distinct, synth.SDSS_SHAPE, workers=6)
        idx = np.arange(a, b) % distinct
        dbase = torch.from_numpy(base).cuda(device)
        frames = dbase[torch.from_numpy(idx).cuda(device)].contiguous()
        del dbase
        packed = synth.pack_catalogs([cats[i] for i in idx])
        catalogs = {k: torch.from_numpy(v).cuda(device) for k, v in packed.items()}
        torch.cuda.synchronize()
    else:
        fr, cats = zip(*[synth.make_frame(k)[:2] for k in range(a, b)])
        frames, catalogs = np.stack(fr), synth.pack_catalogs(list(cats))
    local = det.detect(frames, pb, pd, catalogs, rs)
    spilled = det.spill_count()
    full = batch.gather_results(local, n)
    if rank == 0:
        np.save(out, full)
    dist.barrier()
    det.close()
    dist.destroy_process_group()
    if spilled:
        raise SystemExit(f"rank {rank}: {spilled} frames spilled to the worst-case workspace")


if __name__ == "__main__":
    main()
