"""Developer: run seeds of tests/test_gpu_random.py::test_random_configuration and print WHERE a result differs from the oracle's."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AMR_RANDOM_SEEDS", "3000")
from tests import util
import tests.test_gpu_random as t

def verbose_same(o_res, g_res, packet_symbols=None):
    _, oq, oh, op = o_res
    gq, gh, gp = g_res
    print("   q equal", np.array_equal(oq, gq), "hits equal", oh.shape == gh.shape and np.array_equal(oh, gh), "n", len(oh))
    if op.shape == gp.shape and not np.array_equal(op, gp):
        bad = np.argwhere(op != gp)
        rows = np.unique(bad[:, 0])
        print("   pkt differs in", len(bad), "bytes, columns", np.unique(bad[:, 1]), "rows", rows[:10], "of", len(op))
        for r in rows[:6]:
            print("     row", r, "hit", oh[r], "oracle last %02x gpu last %02x" % (op[r, -1], gp[r, -1]), "prev row oracle last %02x" % (op[r - 1, -1] if r else 0),
                  "prev hit", oh[r - 1] if r else None)
        # order of slicing: (block, pid, idx)
        order = np.lexsort((oh[:, 2], oh[:, 0], oh[:, 1]))
        inv = np.empty_like(order); inv[order] = np.arange(len(order))
        for r in rows[:6]:
            k = inv[r]
            pr = order[k - 1] if k else None
            print("     row", r, "slicing-order predecessor row", pr, "its hit", oh[pr] if pr is not None else None,
                  "its low nibble %x" % ((op[pr, -1] & 15) if pr is not None else 0))
util.assert_same = verbose_same
for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(1000 + seed)
    protos = t.PROTO_SETS[int(rng.integers(len(t.PROTO_SETS)))]
    print("seed", seed, protos)
    for rep in range(3):
        try:
            t.test_random_configuration(seed)
        except Exception as e:
            print("   exception", repr(e)[:200])
