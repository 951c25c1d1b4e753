#!/usr/bin/env python3
"""Golden vectors for the data-key plumbing, produced by the REFERENCE's own
`models/utils.py:prepare_data` (this container only, behind the shim of make_golden.py):
the --testing subsample (drawn from numpy's GLOBAL stream seeded like run_mm_late.py:49), the split, the one-hot label
vectors and sklearn's balanced class weights.  Inputs (a synthetic data-key frame) are stored with the outputs.
The evaluation loop's dict (mm_late.py:534-638) has its own reference-derived fixture: make_eval_golden.py.
The reference's metric functions (utils.py:294-335) need torchmetrics, which this container does not have: no
reference-derived vectors exist for them here (tests check them against scikit-learn instead).
Run:  python tests/golden/make_f3_golden.py
"""
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shim  # noqa: E402


def frame(n, num_labels, seed):
    r = np.random.RandomState(seed)
    split = r.choice(["train", "val", "test"], size=n, p=[0.7, 0.15, 0.15])
    return pd.DataFrame({"tweet_id": r.permutation(n) + 1000, "text": [f"post {i}" for i in range(n)],
                         "label": r.randint(0, num_labels, size=n), "split": split})


def main():
    install_shim()
    import utils as ref_utils      # /root/reference/models/utils.py
    # version skew only: scikit-learn >= 1.2 validates `classes` as an ndarray, the reference (scikit-learn 1.1) passes a list
    _ccw = ref_utils.compute_class_weight
    ref_utils.compute_class_weight = lambda class_weight, classes, y: _ccw(class_weight=class_weight, classes=np.asarray(classes), y=y)
    cases = []
    for n, C, testing, nsamples in ((400, 3, True, -1), (400, 3, False, -1), (260, 4, True, -1), (300, 2, False, 50)):
        data = frame(n, C, 7 + n)
        np.random.seed(30)         # run_mm_late.py:49
        tr, ytr, va, yva, te, yte, cw, _ = ref_utils.prepare_data(data, C, testing=testing, nsamples=nsamples)
        cases.append(dict(n=n, num_labels=C, testing=testing, nsamples=nsamples, frame_seed=7 + n,
                          train_ids=tr.tweet_id.tolist(), val_ids=va.tweet_id.tolist(), test_ids=te.tweet_id.tolist(),
                          y_train=np.asarray(ytr).astype(int).tolist(), y_val=np.asarray(yva).astype(int).tolist(),
                          y_test=np.asarray(yte).astype(int).tolist(), class_weights=[float(x) for x in cw.cpu().numpy()],
                          next_rand=float(np.random.rand())))          # where the global stream stands afterwards (ITM sampling continues from here)
    with open(os.path.join(HERE, "f3_prepare_data.json"), "w") as f:
        json.dump(cases, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
