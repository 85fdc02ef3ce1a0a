"""End-to-end run of the WHOLE workflow (StartProcess steps 0 - 6b: tiling, WGAN-GP, mask simulation, CycleGAN, fake images + filtering,
MultiResUNet, segmentation of the real images) on a MOUNTED copy of the publication's data, followed by the publication's scores
(Calculate_Scores.py: whole-image IoU, instance IoU, Youden index against the manual masks) -- the figure BASELINE.json's
"0.87 val IoU" refers to (README.md:53-57 of the reference: tensorflow 0.8762 / pytorch 0.8502 whole-image IoU, 4:18 h / 6:35 h).

    python tools/real_data_eval.py --data <dir with Input_Images/ Input_Masks/ gt/> --out gpurun_out/real_eval [--set NAME=VALUE ...]

The data (CC BY-NC-ND) is never part of this repository: <dir> is staged next to it for the run (`_eval_data/`, git-ignored).
Writes <out>/results.json (scores, per-step wall times, options), the training logs and the 40 final masks as PNG."""
import argparse
import dataclasses
import importlib
import json
import os
import shutil
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PKG = "automatic-sem-image-segmentation_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--root", default="/tmp/ss_real_eval")
    ap.add_argument("--steps", default="0,1,2,3,4,5,6a,6b")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--score-workers", type=int, default=15)
    ap.add_argument("--no-instance-scores", action="store_true")
    a = ap.parse_args()
    import random
    import torch
    random.seed(a.seed)
    np.random.seed(a.seed)
    torch.manual_seed(a.seed)
    SP = importlib.import_module(PKG + ".StartProcess")
    SC = importlib.import_module(PKG + ".Scoring")
    os.makedirs(a.out, exist_ok=True)
    steps = a.steps.split(",")
    if "0" in steps:
        shutil.rmtree(a.root, ignore_errors=True)
        os.makedirs(a.root)
        for sub in ("Input_Images", "Input_Masks"):
            shutil.copytree(os.path.join(a.data, sub), os.path.join(a.root, sub))
    opts = SP.WorkflowOptions(ROOT_DIR=a.root)
    for kv in a.set:
        k, _, v = kv.partition("=")
        opts.set(k.strip(), v)
    wf = SP.Workflow(opts)
    res = {"options": {k: v for k, v in dataclasses.asdict(opts).items() if not k.startswith("_")}, "seed": a.seed, "step_seconds": {}}
    log = open(os.path.join(a.out, "log.txt"), "a")

    def say(msg):
        print(msg, flush=True)
        log.write(msg + "\n")
        log.flush()

    t_all = time.perf_counter()
    for key in steps:
        t0 = time.perf_counter()
        wf.run_step(key)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        res["step_seconds"][key] = round(time.perf_counter() - t0, 1)
        say(f"step {key}: {res['step_seconds'][key]} s")
        json.dump(res, open(os.path.join(a.out, "results.json"), "w"), indent=1)
    res["workflow_seconds"] = round(time.perf_counter() - t_all, 1)

    # training logs
    for sub in ("1_WGAN/Models", "2_CycleGAN/Models", "3_UNet/Models"):
        d = os.path.join(a.root, sub)
        if os.path.isdir(d):
            for run in sorted(os.listdir(d)):
                src = os.path.join(d, run, "training_log.csv")
                if os.path.exists(src):
                    shutil.copy(src, os.path.join(a.out, sub.split("/")[0] + "_training_log.csv"))

    # scores against the manual masks (rows 0..711: the inputs are the cropped images, the masks carry the SEM info bar rows)
    gt_dir = os.path.join(a.data, "gt")
    t0 = time.perf_counter()
    from PIL import Image
    for name, pred_dir in (("unet", opts.OUTPUT_DIR_UNET), ("cyclegan", opts.OUTPUT_DIR_CYCLEGAN)):
        if not os.path.isdir(pred_dir) or not os.listdir(pred_dir):
            continue
        # (a) the masks the workflow itself writes (Otsu / watershed / 4-connectivity inside run_inference / filter_gan_masks)
        ious = []
        os.makedirs(os.path.join(a.out, name + "_masks"), exist_ok=True)
        for f in sorted(os.listdir(gt_dir)):
            ident = f[:-len("_m.tif")]
            pp = os.path.join(pred_dir, ident + ".tif")
            if not os.path.exists(pp):
                continue
            p = np.asarray(Image.open(pp))
            g = np.asarray(Image.open(os.path.join(gt_dir, f)))[:p.shape[0]]
            ious.append(float(SC.whole_image_iou(p > 0, g > 0)))
            Image.fromarray(((p > 0) * 255).astype(np.uint8)).save(os.path.join(a.out, name + "_masks", ident + ".png"), optimize=True)
        res[name + "_final_masks"] = {"images": len(ious), "iou_whole_mean": float(np.mean(ious)) if ious else None,
                                      "iou_whole_min": float(np.min(ious)) if ious else None, "iou_whole_per_image": [round(v, 4) for v in ious]}
        say(f"{name}: final masks, mean whole-image IoU over {len(ious)} images: {res[name + '_final_masks']['iou_whole_mean']}")
        # (b) Calculate_Scores.py: threshold sweep over the probability maps (UNet: <id>_raw.tif)
        if name == "unet" and not a.no_instance_scores:
            sw = SC.score_directories(pred_dir, gt_dir, crop_rows=712, watershed=True, raw=True, workers=a.score_workers)
            res["unet_calculate_scores"] = sw
            say("unet: Calculate_Scores sweep: " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in sw.items() if not isinstance(v, list)}))
    res["scoring_seconds"] = round(time.perf_counter() - t0, 1)
    res["reference_published"] = {"tensorflow": {"iou_img": 0.8762, "iou_inst": 0.5750, "youden": 0.9120, "run_time": "4:18 h"},
                                  "pytorch": {"iou_img": 0.8502, "iou_inst": 0.5162, "youden": 0.9008, "run_time": "6:35 h"},
                                  "source": "README.md:53-57 of the reference (one run each, unseeded)"}
    json.dump(res, open(os.path.join(a.out, "results.json"), "w"), indent=1)
    say("done: " + json.dumps({k: res[k] for k in ("step_seconds", "workflow_seconds", "scoring_seconds")}))


if __name__ == "__main__":
    main()
