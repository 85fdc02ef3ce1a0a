"""Segmentation scores of the publication (Archive/Other Scripts/Calculate_Scores.py): whole-image IoU (:69-70), instance IoU
(:73-104), ROC rates / Youden index (:107-136) and the 11-threshold sweep with its best average (:221-272).  This is what defines the
"0.87 val IoU" target of BASELINE.json (README.md:53-57).  Host code (numpy / scipy); OpenCV is not available in this image, so
instances are connected components instead of ``cv2.findContours`` polygons (see ``instances``)."""
import numpy as np
from scipy import ndimage

from . import HelperFunctions as HF


def segment(image, threshold, do_watershed=True, min_distance=9):
    """Calculate_Scores.segment (:33-66): threshold (Otsu when < 0) -> EDT -> gaussian sigma 1 -> peak_local_max(9) -> marker watershed
    with lines -> holes filled (3x3 structure).  uint8 {0,1}."""
    img = np.asarray(image)
    if img.dtype != bool:
        if threshold < 0:
            threshold = HF.threshold_otsu(img)
        mask = img > threshold
    else:
        mask = img
    if np.min(mask) == np.max(mask) or not do_watershed:
        return np.asarray(mask > 0, dtype='uint8')
    distance = ndimage.gaussian_filter(ndimage.distance_transform_edt(mask), sigma=1)
    peaks = HF.peak_local_max(distance, min_distance=min_distance)
    local_maxi = np.zeros(mask.shape, dtype='uint8')
    local_maxi[tuple(peaks.T)] = 1
    markers = ndimage.label(local_maxi)[0]
    labels = HF.watershed(-distance, markers, mask=mask, watershed_line=do_watershed) > 0
    labels = ndimage.binary_fill_holes(labels, structure=np.ones((3, 3)))
    return np.asarray(labels * 1.0, dtype='uint8')


def whole_image_iou(image1, image2):
    """|A and B| / |A or B| (:69-70)."""
    return np.sum(np.logical_and(image1, image2)) / np.sum(np.logical_or(image1, image2))


def instances(image):
    """The instances ``cv2.findContours(RETR_LIST)`` + ``drawContours(FILLED)`` yield (:74-82): every 8-connected foreground component
    FILLED, and -- RETR_LIST also returns the hole borders -- every hole (background region not connected to the image border, 4-
    connected) as an instance of its own.  Returns a list of (bounding-box slices, boolean patch, polygon area estimate): the
    shoelace area of the border-pixel polygon is counted from the unit cells it encloses (see below)."""
    fg = np.asarray(image) > 0
    out = []
    lab, n = ndimage.label(fg, structure=np.ones((3, 3)))
    for i, sl in enumerate(ndimage.find_objects(lab), start=1):
        comp = ndimage.binary_fill_holes(lab[sl] == i)
        out.append((sl, comp))
    holes = ndimage.binary_fill_holes(fg) & ~fg
    lab, n = ndimage.label(holes)
    for i, sl in enumerate(ndimage.find_objects(lab), start=1):
        # a hole's border polygon runs through the FOREGROUND pixels around it: the filled instance is the hole dilated by one pixel
        y0, y1 = max(sl[0].start - 1, 0), min(sl[0].stop + 1, fg.shape[0])
        x0, x1 = max(sl[1].start - 1, 0), min(sl[1].stop + 1, fg.shape[1])
        big = (slice(y0, y1), slice(x0, x1))
        out.append((big, ndimage.binary_dilation(lab[big] == i, structure=np.ones((3, 3)))))
    res = []
    for sl, comp in out:
        # Shoelace area of the polygon the 8-connected border following draws through the border PIXEL CENTRES (:139-151 on the contour's
        # vertices), without tracing it: a unit cell between four pixel centres lies inside the polygon when all four pixels belong to the
        # region, half of it when three do (the border cuts the concave corner diagonally), not at all otherwise -- so one-pixel-wide
        # parts (a there-and-back border) contribute 0, as they do in the reference (checked against its polygon_area,
        # tests/golden/make_scoring_goldens.py)
        c = np.pad(comp, 1).astype(np.int8)
        cells = c[:-1, :-1] + c[1:, :-1] + c[:-1, 1:] + c[1:, 1:]
        area = float(np.count_nonzero(cells == 4)) + 0.5 * float(np.count_nonzero(cells == 3))
        res.append((sl, comp, area))
    return res


def instance_iou(image1, image2, min_area=0):
    """Mean over the instances of image1 (polygon area > min_area) of the best IoU with any bounding-box-overlapping instance of
    image2 (:73-104)."""
    inst1, inst2 = instances(image1), instances(image2)
    shape = np.asarray(image1).shape
    scores = []
    for sl1, c1, a1 in inst1:
        if not a1 > min_area:
            continue
        best = 0.0
        full1 = None
        for sl2, c2, _ in inst2:
            if sl2[1].start > sl1[1].stop - 1 or sl2[1].stop - 1 < sl1[1].start or sl2[0].start > sl1[0].stop - 1 or sl2[0].stop - 1 < sl1[0].start:
                continue
            if full1 is None:
                full1 = np.zeros(shape, bool)
                full1[sl1] = c1
            full2 = np.zeros(shape, bool)
            full2[sl2] = c2
            best = max(best, float(whole_image_iou(full1, full2)))
        scores.append(best)
    return 0 if not scores else float(np.sum(scores) / len(scores))


def roc(predicted, ground_truth):
    """TPR, TNR, FPR, FNR of two {0,1} images (:107-136)."""
    p, g = np.asarray(predicted), np.asarray(ground_truth)
    fp, fn = float(np.sum(p > g)), float(np.sum(p < g))
    tn, tp = float(np.sum((p == g) & (p == 0))), float(np.sum((p == g) & (p == 1)))
    tpr = tp / (tp + fn) if tp + fn > 0 else 0
    tnr = tn / (tn + fp) if tn + fp > 0 else 0
    fpr = fp / (tn + fp) if tn + fp > 0 else 0
    fnr = fn / (tp + fn) if tp + fn > 0 else 0
    return tpr, tnr, fpr, fnr


def _sweep_one(args):
    """The eleven thresholds of one (prediction, ground truth) pair: rows of (whole IoU, instance IoU all, instance IoU area > 9, TPR, TNR)."""
    pred, gt, watershed = args
    gt = np.asarray(gt, dtype='uint8')
    gt = gt // np.max(gt)
    image = np.asarray(pred, dtype='float32').copy()
    if np.max(image) > 1.0:
        image /= 255.0
    rows = []
    for t in range(0, 11):
        seg = HF.eight_to_four_connected(segment(image, threshold=t / 10.0, do_watershed=watershed))
        tpr, tnr, _, _ = roc(seg, gt)
        rows.append((whole_image_iou(seg, gt), instance_iou(seg, gt, 0), instance_iou(seg, gt, 9), tpr, tnr))
    return rows


def calculate_iou(predictions, ground_truths, watershed=True, workers=1):
    """The sweeps of calculateIoU (:221-272) and calculateROC (:172-218) over paired lists of prediction images (float in [0,1] or
    [0,255]) and ground-truth masks: thresholds 0.0 .. 1.0 in steps of 0.1, segment + 8->4 connectivity, averages over the images,
    best average per score.  The reference's IoU sweep accumulates threshold ``t`` into slot ``t - 1`` (threshold 0.0 lands in the LAST
    slot) and reports ``slot / 10`` as the "best threshold": reproduced (``best_threshold_*`` are the reference's reported values;
    ``*_true`` the real ones).  ``youden_index`` = the best average TPR + TNR - 1 (the README's "Avg Youdens Index").
    workers > 1: the images are swept in that many processes (the sweep is CPU work: ~10 s per image)."""
    n = float(len(ground_truths))
    jobs = [(p, g, watershed) for p, g in zip(predictions, ground_truths)]
    if workers > 1 and len(jobs) > 1:
        # (results come back in submission order; a lost worker process is noticed and its images are swept inline: HelperFunctions.JobPool)
        pool = HF.JobPool(_sweep_one, min(workers, len(jobs)))
        for job in jobs:
            pool.submit(job)
        per_image = pool.close()
    else:
        per_image = [_sweep_one(j) for j in jobs]
    whole, inst_all, inst_f, youden = [0.0] * 11, [0.0] * 11, [0.0] * 11, [0.0] * 11
    for rows in per_image:
        for t, (w_, ia_, if__, tpr, tnr) in enumerate(rows):
            whole[t - 1] += w_ / n
            inst_all[t - 1] += ia_ / n
            inst_f[t - 1] += if__ / n
            youden[t] += (tpr + tnr - 1) / n

    def best(v):
        b, bi = 0.0, 0
        for i, x in enumerate(v):
            if x > b:
                b, bi = x, i
        return b, bi / 10.0, ((bi + 1) % 11) / 10.0

    w, ia, if_ = best(whole), best(inst_all), best(inst_f)
    yi = max(range(11), key=lambda t: youden[t])
    return dict(iou_whole=w[0], best_threshold_whole=w[1], best_threshold_whole_true=w[2],
                iou_instance_all=ia[0], best_threshold_instance_all=ia[1], best_threshold_instance_all_true=ia[2],
                iou_instance_filtered=if_[0], best_threshold_instance_filtered=if_[1], best_threshold_instance_filtered_true=if_[2],
                youden_index=youden[yi], best_threshold_youden=yi / 10.0,
                iou_whole_by_threshold=[whole[(t - 1) % 11] for t in range(11)], youden_by_threshold=list(youden))


# ---- one-command evaluation of a mounted dataset ---------------------------------------------------------------------------------
# The publication's dataset (CC BY-NC-ND: mount it, do not vendor it) has the layout of the reference's ``Datasets/`` directory:
IMAGES_SUBDIR = "Electron Microscopy Images/SEM"                                  # 40 images <id>.tif, 768 x 1024 (rows 712.. = the SEM info bar)
GROUND_TRUTH_SUBDIR = "Electron Microscopy Image Masks/TiO2_Masks_Manual_4connected"     # <id>_m.tif


def score_directories(prediction_dir, ground_truth_dir, crop_rows=0, watershed=True, raw=True, limit=None, workers=1):
    """``calculateIoU(dir)`` of Calculate_Scores.py:221-272 over a directory pair: every ground truth ``<id>_m.tif`` is paired with the
    prediction ``<id>_raw.tif`` (the float probability map UNet.run_inference writes; ``raw=False`` or no such file: ``<id>.tif``), both
    cropped to their first ``crop_rows`` rows when > 0.  Returns calculate_iou's dict + the number of pairs."""
    import os
    from PIL import Image
    preds, gts, used = [], [], []
    for name in sorted(os.listdir(ground_truth_dir)):
        stem, ext = os.path.splitext(name)
        if ext.lower() not in (".tif", ".tiff", ".png"):
            continue
        ident = stem[:-2] if stem.endswith("_m") else stem
        cands = ([ident + "_raw.tif"] if raw else []) + [ident + ".tif", ident + ".png"]
        path = next((os.path.join(prediction_dir, c) for c in cands if os.path.exists(os.path.join(prediction_dir, c))), None)
        if path is None:
            continue
        p, g = np.asarray(Image.open(path)), np.asarray(Image.open(os.path.join(ground_truth_dir, name)))
        if p.ndim == 3:
            p = p[..., 0]
        if g.ndim == 3:
            g = g[..., 0]
        if crop_rows > 0:
            p, g = p[:crop_rows], g[:crop_rows]
        if p.shape != g.shape:
            raise ValueError(f"{path}: prediction {p.shape} and ground truth {g.shape} differ in shape")
        preds.append(p)
        gts.append(g)
        used.append(ident)
        if limit and len(used) >= limit:
            break
    if not used:
        raise FileNotFoundError(f"no prediction in {prediction_dir} matches a ground-truth file of {ground_truth_dir}")
    out = calculate_iou(preds, gts, watershed=watershed, workers=workers)
    out["images"] = len(used)
    return out


def main(argv=None):
    """python -m automatic-sem-image-segmentation_amd.Scoring --data-root /mnt/Datasets --model <run>/3_UNet/Models/<ts>/model.keras

    Scores a trained MultiResUNet (or a directory of predictions) on a MOUNTED copy of the publication's dataset and prints the
    whole-image / instance IoU of Calculate_Scores.py as one JSON line -- the figure BASELINE.json's "0.87 val IoU" refers to
    (README.md:53-57 of the reference)."""
    import argparse
    import json
    import os
    import tempfile
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--data-root", required=True, help="the dataset root (layout of the reference's Datasets/ directory)")
    ap.add_argument("--images", default=IMAGES_SUBDIR, help="image sub-directory of --data-root")
    ap.add_argument("--ground-truth", default=GROUND_TRUTH_SUBDIR, help="ground-truth mask sub-directory of --data-root (<id>_m.tif)")
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--model", help="model.keras / .npz of UNet.run_training: segment --images on the GPU first, then score")
    src.add_argument("--predictions", help="directory of predictions (<id>_raw.tif probability maps or <id>.tif masks)")
    ap.add_argument("--crop-rows", type=int, default=712, help="score the first N rows only (712: without the SEM info bar; 0: whole image)")
    ap.add_argument("--tile", type=int, nargs=2, metavar=("W", "H"), default=None, help="tile size for inference (default: whole image)")
    ap.add_argument("--no-watershed", action="store_true")
    ap.add_argument("--limit", type=int, default=None, help="score the first N images only")
    ap.add_argument("--workers", type=int, default=HF.default_workers(16), help="processes for the threshold sweep")
    a = ap.parse_args(argv)
    img_dir, gt_dir = os.path.join(a.data_root, a.images), os.path.join(a.data_root, a.ground_truth)
    for d in (img_dir, gt_dir):
        if not os.path.isdir(d):
            raise SystemExit(f"{d}: no such directory (is the dataset mounted at --data-root?)")
    pred_dir = a.predictions
    tmp = None
    if a.model:
        from . import UNet_Segmentation as UN
        tmp = tempfile.TemporaryDirectory()
        pred_dir = tmp.name
        src_dir = img_dir
        if a.crop_rows > 0:
            # the network must see what is scored: the percentile normalisation of an image that still carries the SEM info bar differs
            # from that of its first crop_rows rows (the published figures were obtained on cropped inputs, tools/real_data_eval.py)
            from PIL import Image
            src_dir = os.path.join(tmp.name, "_cropped_inputs")
            os.makedirs(src_dir)
            for name in sorted(os.listdir(img_dir)):
                if os.path.splitext(name)[1].lower() in (".tif", ".tiff", ".png", ".bmp", ".jpg"):
                    Image.fromarray(np.asarray(Image.open(os.path.join(img_dir, name)))[:a.crop_rows]).save(os.path.join(src_dir, name))
        un = UN.UNet(root_dir=tmp.name, image_dir=src_dir, mask_dir=gt_dir)
        if a.tile:
            un.image_shape = tuple(a.tile)
        un.run_inference(files=src_dir, output_directory=pred_dir, model=a.model, tile_images=a.tile is not None, use_gpu=True)
    res = score_directories(pred_dir, gt_dir, crop_rows=a.crop_rows, watershed=not a.no_watershed, limit=a.limit, workers=a.workers)
    res.update(data_root=a.data_root, source=a.model or a.predictions, crop_rows=a.crop_rows)
    print(json.dumps(res))
    if tmp is not None:
        tmp.cleanup()
    return res


if __name__ == "__main__":
    main()
