"""Are two builds of the library bit-identical on a parity case?  (GPU box)
    python tools/ab_bits.py dump out.npz [case]      # with GT_HIP_LIB selecting the build
    python tools/ab_bits.py cmp a.npz b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

if sys.argv[1] == "dump":
    import cases as C
    from hip_runner import run_hip_case
    got = run_hip_case(C.CASES[sys.argv[3] if len(sys.argv) > 3 else "acoustic_mlp"])
    np.savez(sys.argv[2], **{k: np.asarray(v) for k, v in got.items()})
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint8) if a[k].dtype.kind == "f" else a[k], b[k].view(np.uint8) if b[k].dtype.kind == "f" else b[k])]
    print("%d arrays, %d differ%s" % (len(a.files), len(bad), (": " + " ".join(bad[:8])) if bad else ""))
    sys.exit(1 if bad else 0)
