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
    shoelace area of the border-pixel polygon is estimated with Pick's theorem (filled pixels - border pixels / 2 - 1)."""
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
        border = comp & ~ndimage.binary_erosion(comp, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]], border_value=0)
        area = max(float(comp.sum()) - float(border.sum()) / 2.0 - 1.0, 0.0)
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


def calculate_iou(predictions, ground_truths, watershed=True):
    """The sweep of calculateIoU (:221-272) over paired lists of prediction images (float in [0,1] or [0,255]) and ground-truth masks:
    thresholds 0.0 .. 1.0 in steps of 0.1, segment + 8->4 connectivity, averages over the images, best average per score.
    The reference accumulates threshold ``t`` into slot ``t - 1`` (threshold 0.0 lands in the LAST slot) and reports ``slot / 10``
    as the "best threshold": reproduced (``best_threshold_*`` are the reference's reported values; ``*_true`` the real ones)."""
    n = float(len(ground_truths))
    whole, inst_all, inst_f = [0.0] * 11, [0.0] * 11, [0.0] * 11
    for pred, gt in zip(predictions, ground_truths):
        gt = np.asarray(gt, dtype='uint8')
        gt = gt // np.max(gt)
        image = np.asarray(pred, dtype='float32').copy()
        if np.max(image) > 1.0:
            image /= 255.0
        for t in range(0, 11):
            seg = HF.eight_to_four_connected(segment(image, threshold=t / 10.0, do_watershed=watershed))
            whole[t - 1] += whole_image_iou(seg, gt) / n
            inst_all[t - 1] += instance_iou(seg, gt, 0) / n
            inst_f[t - 1] += instance_iou(seg, gt, 9) / n

    def best(v):
        b, bi = 0.0, 0
        for i, x in enumerate(v):
            if x > b:
                b, bi = x, i
        return b, bi / 10.0, ((bi + 1) % 11) / 10.0

    w, ia, if_ = best(whole), best(inst_all), best(inst_f)
    return dict(iou_whole=w[0], best_threshold_whole=w[1], best_threshold_whole_true=w[2],
                iou_instance_all=ia[0], best_threshold_instance_all=ia[1], best_threshold_instance_all_true=ia[2],
                iou_instance_filtered=if_[0], best_threshold_instance_filtered=if_[1], best_threshold_instance_filtered_true=if_[2])
