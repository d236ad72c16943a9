"""Progress / best-AP bookkeeping (reference misc/logger.py), without a per-step device sync:
losses are accumulated as device tensors and only read when the bar is refreshed."""


class Logger():
    def __init__(self, refresh_every=20):
        self.bestAP = -1
        self.progressBar = None
        self.refresh_every = refresh_every
        self._n = 0

    def clear(self, loaderSize):
        try:
            from tqdm import tqdm
            self.progressBar = tqdm(total=loaderSize)
        except Exception:
            self.progressBar = None
        self._n = 0

    def display(self, loss, loss2, updateSize, epoch):
        self._n += 1
        if self.progressBar is None:
            return
        if self._n % self.refresh_every == 0:
            post = dict(EP=epoch, Loss=float(loss))
            if loss2 is not None:
                post["Loss2"] = float(loss2)
            self.progressBar.set_postfix(**post)
        self.progressBar.update(updateSize)

    def showBestAP(self):
        return self.bestAP

    def updateBestAcc(self, acc):
        self.bestAP = acc

    def isBestAccAP(self, acc):
        if acc > self.bestAP or self.bestAP == -1:
            self.bestAP = acc
            return True
        return False
