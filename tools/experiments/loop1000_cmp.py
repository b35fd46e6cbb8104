"""Compare two dumps of tools/experiments/loop1000_dump.py."""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in ("fwd", "x0"):
    d = np.abs(a[k].astype(np.float64) - b[k])
    print("%s: max-abs diff %.3e  rms diff %.3e  (absmax %.3f)" % (k, d.max(), np.sqrt((d ** 2).mean()), np.abs(a[k]).max()))
