"""compareValues with the reference's signature and return values (verifyRoutines.py:12-66):
largest raw and largest fractional difference between a reference array and a test array."""

import numpy as np


def compareValues(x, y, plotAbs=False, verbose=True):
    x = np.asarray(x)
    y = np.asarray(y)
    d = np.abs(x - y)
    ii = int(np.argmax(d))
    rawChg = d.reshape(-1)[ii]
    nz = np.flatnonzero(x.reshape(-1) != 0)
    frac = d.reshape(-1)[nz] / np.abs(x.reshape(-1)[nz])
    ip = int(np.argmax(frac)) if nz.size else 0
    fracChg = frac[ip] if nz.size else 0.0
    if verbose:
        print("Values with largest raw change (index %d):" % ii)
        print(x.reshape(-1)[ii])
        print(y.reshape(-1)[ii])
        if nz.size:
            print("Values with largest %% change (index %d):" % nz[ip])
            print(x.reshape(-1)[nz[ip]])
            print(y.reshape(-1)[nz[ip]])
    if plotAbs:  # plotting is optional and only imported on request
        import matplotlib.pyplot as plt

        plt.figure()
        plt.plot(np.abs(x), label="x")
        plt.plot(np.abs(y), label="y")
        plt.plot(d, "k--", label="x-y")
        plt.legend()
    return rawChg, fracChg
