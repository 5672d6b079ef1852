"""Child process of tests/test_gpu_multirank.py: one rank of a frame-parallel job on ONE device.

    python tests/_rank_worker.py <rank> <world> <port> <n_frames> <device> <out.npy>

Started as a fresh program (never a fork of a GPU-initialised process).  The rank generates the frames of its
contiguous block (batch.shard_range), runs the full pipe on them through its own BatchDetector, takes part in the gloo
gather of the 48-byte records (no data-path collective) and rank 0 saves the gathered array."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, n, device = (int(x) for x in sys.argv[1:6])
    out = sys.argv[6]
    import numpy as np
    import torch.distributed as dist
    from lfd_amd import _native, batch, synth
    from lfd_amd.detecttrails import default_params
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb, pd, prs = default_params()
    rs = _native.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    a, b = batch.shard_range(n, rank, world)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(a, b)])
    det = batch.BatchDetector(device, synth.SDSS_SHAPE, inflight=4)        # 4 slots: several chunks per rank
    local = det.detect(np.stack(frames), pb, pd, synth.pack_catalogs(list(cats)), rs)
    full = batch.gather_results(local, n)
    if rank == 0:
        np.save(out, full)
    dist.barrier()
    det.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
