"""Wall-clock Timer with the reference's interface (timingRoutines.py:12-75):
start() / evt(label) / rpt(showSteps) / end(label) on time.perf_counter()."""

import time


class Timer:
    def __init__(self):
        self.t = []
        self.labels = []

    def reset(self):
        self.t.clear()
        self.labels.clear()

    def start(self):
        self.reset()
        self.evt()

    def evt(self, label=""):
        self.t.append(time.perf_counter())
        self.labels.append(label)

    def rpt(self, showSteps=True):
        if showSteps:
            for i in range(1, len(self.t)):
                print("%d->%d : %fs. %s" % (i - 1, i, self.t[i] - self.t[i - 1], self.labels[i]))
        total = self.t[-1] - self.t[0]
        print("Total: %fs." % total)
        return total

    def end(self, label="", showSteps=True):
        self.evt(label)
        return self.rpt(showSteps=showSteps)
