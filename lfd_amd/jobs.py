"""One selection over the GPUs of a node: the replacement of the reference's PBS job fan-out.

The reference scales out by cutting its run list into jobs, writing a PBS script per job from a template and submitting them
(``lfd/createjobs/createjobs.py:173-202``, ``createjobs/writer.py:22-88``); every job runs ``DetectTrails(...).process()`` on its
runs and leaves its own results / errors files.  Frames are independent, so here the *selection* (whatever ``DetectTrails``'
keyword arguments pick: a run, a run list, a camcol of a run, ...) is cut into contiguous blocks of frames, one per GPU
(``lfd_amd.batch.shard_bounds``, the rule ``DetectTrails.process(rank=, world_size=)`` and ``bench.py`` use), one worker process per
GPU, and the workers' files are joined in selection order at the end: ``results.txt`` and ``errors.txt`` are then what a single
process would have written.  No scheduler, no templates: a node's GPUs are the queue.

    from lfd_amd.jobs import Jobs
    Jobs(8, run=94, savepath="/scratch/out").launch(batch=256)           # eight GPUs, one worker each

    python -m lfd_amd.jobs --gpus 8 --savepath /scratch/out --batch 256 run=94 camcol=1

``resume=True`` continues an interrupted launch (every worker skips what its own progress file lists, ``DetectTrails.process``).
"""
import os
import pickle
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def merge_rank_files(path, world_size, remove=False):
    """``path.rank0`` .. ``path.rank<world_size - 1>`` joined, in rank order, into ``path`` (appended to, as the reference's files
    are): the ranks hold contiguous blocks of the selection, so this is the selection's order.  Returns the bytes written."""
    if world_size <= 1:
        return 0
    written = 0
    with open(path, "ab") as out:
        for r in range(world_size):
            part = f"{path}.rank{r}"
            if not os.path.exists(part):
                continue
            with open(part, "rb") as f:
                data = f.read()
            out.write(data)
            written += len(data)
            if remove:
                os.remove(part)
    return written


class Jobs:
    """``n_workers`` worker processes, one per entry of ``devices`` (default 0 .. n_workers - 1), over the selection the remaining
    keyword arguments describe to ``DetectTrails`` (run / runs / camcol / filter / field, params_*, savepath, results, errors)."""

    def __init__(self, n_workers, devices=None, python=None, **detecttrails_kwargs):
        self.n = int(n_workers)
        if self.n < 1:
            raise ValueError("n_workers must be at least 1")
        self.devices = list(range(self.n)) if devices is None else [int(d) for d in devices]
        if len(self.devices) != self.n:
            raise ValueError("one device per worker")
        self.python = python or sys.executable
        self.kwargs = dict(detecttrails_kwargs)
        self.returncodes = None

    def commands(self, spec_path):
        """[(argv, environment additions)] of the workers (what ``launch`` starts)."""
        cmds = []
        for r in range(self.n):
            env = {"RANK": str(r), "WORLD_SIZE": str(self.n), "LOCAL_RANK": str(r), "LOCAL_WORLD_SIZE": str(self.n),
                   "LFD_DEVICE": str(self.devices[r])}
            cmds.append(([self.python, "-m", "lfd_amd.jobs", "--worker", spec_path], env))
        return cmds

    def launch(self, batch=256, resume=False, timeout=None, merge=True, log_dir=None):
        """Runs the workers side by side and waits for them; joins their results / errors files (``merge``).  Raises
        ``RuntimeError`` naming the ranks that failed (their files are left as they are; ``resume=True`` picks up from there)."""
        from .detecttrails import DetectTrails
        probe = DetectTrails(**self.kwargs)                          # (validates the selection; gives the output paths)
        results, errors = probe.results, probe.errors
        if not resume:
            for path in (results, errors):                           # the joined files are rewritten by this launch
                if self.n > 1 and os.path.exists(path):
                    os.remove(path)
        log_dir = log_dir or os.path.dirname(os.path.abspath(results))
        with tempfile.NamedTemporaryFile("wb", suffix=".lfdjob", delete=False) as f:
            pickle.dump({"kwargs": self.kwargs, "batch": int(batch), "resume": bool(resume)}, f)
            spec = f.name
        procs, logs = [], []
        try:
            for r, (argv, env_add) in enumerate(self.commands(spec)):
                env = dict(os.environ)
                env.update(env_add)
                env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
                log = open(os.path.join(log_dir, f"worker{r}.log"), "wb")
                logs.append(log)
                procs.append(subprocess.Popen(argv, env=env, stdout=log, stderr=subprocess.STDOUT))
            codes = []
            for p in procs:
                try:
                    codes.append(p.wait(timeout=timeout))
                except subprocess.TimeoutExpired:
                    codes.append("timeout")
            self.returncodes = codes
        finally:
            for p in procs:                                          # nothing outlives the launch
                if p.poll() is None:
                    p.kill()
                    p.wait()
            for log in logs:
                log.close()
            os.remove(spec)
        bad = [r for r, c in enumerate(self.returncodes) if c != 0]
        if bad:
            raise RuntimeError(f"workers {bad} failed (exit codes {self.returncodes}); see worker<r>.log in {log_dir}")
        if merge and self.n > 1:
            merge_rank_files(results, self.n, remove=True)
            merge_rank_files(errors, self.n, remove=True)
        return results, errors


def _worker(spec_path):
    with open(spec_path, "rb") as f:
        spec = pickle.load(f)
    from .detecttrails import DetectTrails
    dt = DetectTrails(**spec["kwargs"])
    dt.process(batch=spec["batch"], rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), resume=spec["resume"])


def _parse_value(text):
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            pass
    if "," in text:
        return [_parse_value(t) for t in text.split(",") if t]
    return text


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="DetectTrails over the GPUs of this node (the reference's createjobs, without a scheduler)")
    ap.add_argument("--worker", metavar="SPEC", help=argparse.SUPPRESS)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--devices", type=str, default=None, help="comma-separated device indices (default 0 .. gpus - 1)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--savepath", type=str, default=None)
    ap.add_argument("selection", nargs="*", help="DetectTrails keyword arguments: run=94 camcol=1 filter=r field=100 runs=94,125")
    args = ap.parse_args(argv)
    if args.worker:
        _worker(args.worker)
        return 0
    kwargs = {}
    for item in args.selection:
        k, _, v = item.partition("=")
        kwargs[k] = _parse_value(v)
    if args.savepath:
        kwargs["savepath"] = args.savepath
    devices = [int(d) for d in args.devices.split(",")] if args.devices else None
    results, errors = Jobs(args.gpus, devices=devices, **kwargs).launch(batch=args.batch, resume=args.resume)
    print(results)
    print(errors)
    return 0


if __name__ == "__main__":
    sys.exit(main())
