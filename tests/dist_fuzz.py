"""Random drop-in trajectories on 1, 2 and 3 ranks (sweep, not collected by pytest; GPU):
`python tests/dist_fuzz.py [first_seed] [count]`.

For every seed the trajectory tests/traj_fuzz.py draws is run through tests/dist_worker.py's fuzz mode -- fresh
processes sharing cuda:0 over gloo, each holding its videos of dist.shard_videos -- and the 2- and 3-rank runs
must leave, after every call, the single-rank run's tree (ids, leaf flags) with its parameters (1e-4 of the largest
alpha, 1e-4 of each sigma), on every rank; a collapse to the root must raise in the same call everywhere.
"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.traj_fuzz import config   # noqa: E402


def run(world, seed, out):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    os.makedirs(out)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world),
                               port, out, "fuzz:%d" % seed], cwd=ROOT) for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    if codes != [0] * world:
        raise RuntimeError("world %d seed %d: exit codes %s" % (world, seed, codes))
    return [dict(np.load(os.path.join(out, "rank%d.npz" % r), allow_pickle=False)) for r in range(world)]


def compare(want, got, world):
    for ep in range(5):
        for r, g in enumerate(got):
            if ("e%d_raised" % ep in want) != ("e%d_raised" % ep in g):
                return "call %d: raised differs on rank %d of %d" % (ep, r, world)
            if "e%d_raised" % ep in want:
                continue
            if "e%d_ids" % ep not in want:
                continue
            if list(g["e%d_ids" % ep]) != list(want["e%d_ids" % ep]) or not np.array_equal(g["e%d_is_leaf" % ep], want["e%d_is_leaf" % ep]):
                return "call %d: STRUCTURE differs on rank %d of %d (K %d vs %d)" % (ep, r, world, len(g["e%d_ids" % ep]), len(want["e%d_ids" % ep]))
            wa, ga = want["e%d_alpha" % ep], g["e%d_alpha" % ep]
            da = float(np.abs(wa - ga).max()) / max(1.0, float(np.abs(wa).max()))
            ds = float(np.abs(g["e%d_sigma" % ep] / want["e%d_sigma" % ep] - 1).max())
            if da > 1e-4 or ds > 1e-4:
                return "call %d: VALUES differ on rank %d of %d: alpha %.2e sigma %.2e" % (ep, r, world, da, ds)
        if "e%d_raised" % ep in want:
            break
    return ""


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    bad = 0
    for seed in range(first, first + count):
        tag = config(seed)[-1]
        with tempfile.TemporaryDirectory() as tmp:
            want = run(1, seed, os.path.join(tmp, "w1"))[0]
            res = ""
            for world in (2, 3):
                res = compare(want, run(world, seed, os.path.join(tmp, "w%d" % world)), world)
                if res:
                    break
        ks = [len(want["e%d_ids" % ep]) for ep in range(5) if "e%d_ids" % ep in want]
        bad += bool(res)
        print(tag + " -> " + (res or "ok (K per call %s)" % ks), flush=True)
    print("%d of %d differ" % (bad, count))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
