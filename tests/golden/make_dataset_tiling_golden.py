#!/usr/bin/env python3
"""Generates tests/golden/dataset_tiling_golden.json with the reference's own SceneDataLoader
(gpudrive/env/dataset.py:64-67: a dataset smaller than the batch is repeated until it fills it).
Run in the authoring container only:

    PYTHONPATH=/root/reference python tests/golden/make_dataset_tiling_golden.py
"""
import json
import os
import tempfile

from gpudrive.env.dataset import SceneDataLoader  # reference code

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cases = []
for n_files, batch in ((3, 8), (3, 1024), (5, 5), (7, 16), (1, 4)):
    d = tempfile.mkdtemp()
    for i in range(n_files):
        open(os.path.join(d, "tfrecord-%05d.json" % i), "w").close()
    dl = SceneDataLoader(root=d, batch_size=batch, dataset_size=n_files)  # dataset_size < batch_size triggers the repetition
    batch_files = next(iter(dl))
    cases.append(dict(n_files=n_files, batch=batch, indices=[int(os.path.basename(f)[9:14]) for f in batch_files]))
out = os.path.join(ROOT, "tests", "golden", "dataset_tiling_golden.json")
json.dump(cases, open(out, "w"))
print("wrote", out, [(c["n_files"], c["batch"], c["indices"][:10]) for c in cases])
