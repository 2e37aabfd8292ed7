#!/usr/bin/env python3
"""Evaluation entry point with the reference's CLI (evaluate.py:51-170 upstream):

    python evaluate.py --config_file F --checkpoint P [--overwrite JSON]

The student's detections (score > conf_threshold, class filter, NMS — the same device kernels that make the
teachers' pseudo-labels) are scored against the merged multi-teacher pseudo ground truth with the reference's
AP@0.5 / AP@0.75 / AP@Ave / CDx / CDy definitions (mm_distillnet_amd/metrics.py, pinned by the reference's evaluate() through
tests/golden/metrics_eval.npz) and written to <exp>/results.<rank>.csv with the reference's columns.
"""
import argparse
import os
import sys

import numpy as np

# Hardware queues (opt-in): the step's graph has six concurrent branches (main chain, three teachers, weight gradients,
# regressor head); the HIP runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).
# MMD_HW_QUEUES=8 measured 21.35 ms/step (mean of 12 runs) against 21.73 with the default; left off by default because the
# runtime proved fragile away from its default (2 queues: 40.7 ms/step or a segfault inside graph replay) - profiles/r01_notes.md.
# Must be in the environment before the HIP runtime initialises, i.e. before torch is imported.
if os.environ.get("MMD_HW_QUEUES"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ["MMD_HW_QUEUES"])
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import train as T  # noqa: E402
from mm_distillnet_amd.data import SyntheticMultimodalDetection, collate  # noqa: E402
from mm_distillnet_amd.metrics import evaluate_table  # noqa: E402
from mm_distillnet_amd.step import DistillEngine  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_file", required=True)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--overwrite", type=str, default=None)
    ap.add_argument("--just_plot", action="store_true")
    a = ap.parse_args(argv)
    cfg, _ = T.parse_config(["--config_file", a.config_file] + (["--overwrite", a.overwrite] if a.overwrite else []))
    torch.cuda.set_device(0)
    dev = "cuda:0"
    sspec, sstate, tspecs, tstates = T.load_states(cfg, int(cfg.get("compound_coef", 2)))
    if a.checkpoint:
        c = torch.load(a.checkpoint, map_location="cpu", weights_only=False)
        sstate = c["state_dict"] if "state_dict" in c else c
    eng = DistillEngine(sspec, tspecs, dev, T.step_config(cfg))
    eng.load(sstate, tstates)
    test_set = SyntheticMultimodalDetection(cfg, "test")
    loader = torch.utils.data.DataLoader(test_set, batch_size=cfg.getint("batch_size"), shuffle=False, collate_fn=collate)
    # get_predictions_multiteacher (src/utils/utils.py:1720-1830): per batch the student's detections and the merged teacher
    # pseudo ground truth, plus the flat list of ground-truth class ids
    all_pred, all_lab, labels = [], [], []
    for rgb, thermal, depth, audio, _, ids in loader:
        batch = {"rgb": rgb.to(dev), "thermal": thermal.to(dev), "depth": depth.to(dev), "audio": audio.to(dev)}
        p, l = eng.predict(batch)
        all_pred.append(p); all_lab.append(l)
        labels += [float(r[4]) for t in l for r in np.asarray(t, np.float32).reshape(-1, 5)]
    eng.check_overflow()
    # evaluate() (src/utils/utils.py:2018-2181): one row per testing point; 'ALL' when the three teachers are in use
    mods = [m for m in ("rgb", "depth", "thermal") if cfg.getboolean(f"use_{m}", True)]
    modality = "ALL" if len(mods) == 3 else ",".join(mods)
    table = evaluate_table(all_pred, all_lab, labels, cfg.getint("image_size"))
    print({k: round(v, 3) for k, v in table.items()})
    if os.path.exists(cfg["exp_name"]):
        import pandas as pd
        pd.DataFrame([dict(exp_name=cfg["exp_name"], modality=modality, **table)]).to_csv(
            f"{cfg['exp_name']}/results.{cfg['rank']}.csv", index=False)
    return table


if __name__ == "__main__":
    main()
