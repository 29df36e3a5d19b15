"""Test scaffolding: the numpy restatement of the KERNELS (tests/model_bulk.py: level-ordered can_see,
round-synchronous round loop, finalize, voter masks, candidate-major elections) behind the
partition interface of engine.Hashgraph (`decide_fame_partial` / `commit_fame`), so that the
multi-GPU partition logic of py-swirld_amd/partition.py can run on CPU ranks over gloo.  Not the
sequential oracle: this is the reformulated algorithm the HIP kernels implement."""
import numpy as np

import model_bulk as mb


class ModelHashgraph:
    def __init__(self, n, stream, stake=None):
        cr, sp, op, t, sig = stream
        self.n = n
        self.stake = np.ones(n, np.int64) if stake is None else np.asarray(stake, np.int64)
        self.coin = sig[:, 0] >= 128
        self.L, lo, _ = mb.bulk_rounds_v3(n, cr, sp, op, self.stake, K=6, skip=1)
        self.rnd, S, self.wit = mb.finalize(n, cr, self.L, lo)
        self.Sw = mb.voter_masks(n, self.L, self.rnd, S, self.wit, self.stake)
        R = self.wit.shape[0]
        self.famous = np.full((R, n), -1, np.int8)
        self.consensus = np.zeros(R, np.uint8)

    def _max_c(self):
        m = 0
        while m < len(self.consensus) and self.consensus[m]:
            m += 1
        return m

    def decide_fame_partial(self, part, nparts):
        """Elections of the candidate rounds max_c + part, + nparts, ... on private copies."""
        R = self.wit.shape[0]
        max_c = self._max_c()
        fam = self.famous.copy()
        cons = self.consensus.copy()
        # rounds of other parts are masked as "in consensus" so that elections() skips them
        for r in range(max_c, R):
            if (r - max_c) % nparts != part:
                cons[r] = 1
        before = cons.copy()
        # (elections() takes max_c from the consensus flags: keep it by handing it a table whose first
        # open round is this part's first round; rounds below stay closed)
        new_c, _ = mb.elections(self.n, self.wit, self.Sw, self.stake, self.coin, fam, cons)
        decided = np.zeros(R, np.uint8)
        for r in new_c:
            decided[r] = 1
        for r in range(max_c, R):
            if (r - max_c) % nparts != part:
                fam[r] = -1
        assert (cons >= before).all()
        return fam, decided

    def commit_fame(self, famous, decided):
        max_c = self._max_c()
        self.famous[max_c:] = np.asarray(famous, np.int8)[max_c:]
        new_c = [int(r) for r in np.nonzero(decided)[0] if not self.consensus[r]]
        for r in new_c:
            self.consensus[r] = 1
        return new_c

    def decide_fame(self):
        new_c, _ = mb.elections(self.n, self.wit, self.Sw, self.stake, self.coin, self.famous, self.consensus)
        return new_c
