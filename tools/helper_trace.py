#!/usr/bin/env python3
"""PWA_TRACE_STRIPE trace -> what the helper wave of the traced workgroup did: iterations, their period, columns per iteration."""
import sys
import numpy as np
t = np.loadtxt(sys.argv[1]).astype(np.float64)
tm = t[:, 3]
pk = t[:, 4]
ok = (tm > 0) & (tm < 1e9)
tm, pk = tm[ok], pk[ok].astype(np.uint64)
kin, kout = (pk >> np.uint64(32)).astype(np.int64), (pk & np.uint64(0xffffffff)).astype(np.int64)
print("helper iterations recorded: %d (trace holds 8192); time span %.0f us; kin %d..%d kout %d..%d" % (len(tm), (tm[-1] - tm[0]) / 100, kin[0], kin[-1], kout[0], kout[-1]))
dt = np.diff(tm) / 100.0
din, dout = np.diff(kin), np.diff(kout)
busy = (din > 0) | (dout > 0)
print("iteration period us: median %.2f mean %.2f p90 %.2f max %.1f; productive iterations %d of %d" % (np.median(dt), dt.mean(), np.percentile(dt, 90), dt.max(), busy.sum(), len(dt)))
print("period of productive iterations: median %.2f mean %.2f us; idle ones: median %.2f us" % (np.median(dt[busy]), dt[busy].mean(), np.median(dt[~busy]) if (~busy).any() else 0))
print("columns staged per productive in-trip: median %d mean %.1f max %d; published per out-trip: median %d mean %.1f max %d"
      % (np.median(din[din > 0]), din[din > 0].mean(), din.max(), np.median(dout[dout > 0]) if (dout > 0).any() else 0, dout[dout > 0].mean() if (dout > 0).any() else 0, dout.max()))
half = len(tm) // 2
print("rate over the second half of the record: staged %.2f columns/us, published %.2f columns/us" % ((kin[-1] - kin[half]) / ((tm[-1] - tm[half]) / 100), (kout[-1] - kout[half]) / ((tm[-1] - tm[half]) / 100)))
