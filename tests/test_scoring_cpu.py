"""Host-side scoring / mask filtering (SURVEY 8f-2 remainder): threshold_li against scikit-image 0.18.3 values, filter_gan_masks
(HelperFunctions.py:163-185) and the publication's IoU / ROC scores (Archive/Other Scripts/Calculate_Scores.py) on constructed cases."""
import importlib
import os

import numpy as np
import pytest

BASE = "automatic-sem-image-segmentation_amd"
HF = importlib.import_module(BASE + ".HelperFunctions")
SC = importlib.import_module(BASE + ".Scoring")


def test_threshold_li_matches_skimage(golden_dir):
    z = np.load(os.path.join(golden_dir, "threshold_li.npz"))
    for k in z.files:
        if k.endswith("_li"):
            got = HF.threshold_li(z[k[:-3]])
            assert abs(float(got) - float(z[k])) <= 1e-6 * max(abs(float(z[k])), 1.0), (k, got, float(z[k]))


def test_filter_gan_masks_keeps_rendered_particles(tmp_path):
    """Three particles in the mask; the generated image renders two of them brightly and one not at all: the dark one is removed
    (mean intensity below threshold_li of the image), the others are written filled -- including a hole inside one of them."""
    from PIL import Image
    img_dir, msk_dir, out_dir = (str(tmp_path / d) for d in ("img", "msk", "out"))
    os.makedirs(img_dir), os.makedirs(msk_dir)
    rng = np.random.default_rng(0)
    h = w = 96
    yy, xx = np.mgrid[0:h, 0:w]
    disc = lambda cy, cx, r: (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    mask = np.zeros((h, w), np.uint8)
    a, b, c = disc(24, 24, 12), disc(70, 30, 10), disc(40, 72, 11)
    mask[a | b | c] = 255
    mask[disc(24, 24, 5)] = 0                                  # a hole in particle a: not rendered -> its contour is dropped -> filled
    img = np.clip(rng.normal(30, 4, (h, w)), 0, 255)
    img[a] += 150
    img[b] += 140                                               # c is not rendered
    img[disc(24, 24, 5)] = 30                                   # ... nor is the inside of a's hole
    Image.fromarray(mask).save(os.path.join(msk_dir, "t.tif"))
    Image.fromarray(img.astype(np.uint8)).save(os.path.join(img_dir, "t.tif"))
    HF.filter_gan_masks(img_dir, msk_dir, out_dir, do_watershed_and_four_connectivity=False)
    out = np.array(Image.open(os.path.join(out_dir, "t.tif")))
    assert out.dtype == np.uint8 and set(np.unique(out)) == {0, 255}
    assert np.all(out[a] == 255) and np.all(out[b] == 255) and np.all(out[c] == 0)     # hole of a filled, c dropped
    assert np.all(out[~(a | b)] == 0)
    # bright background: keep particles DARKER than the threshold
    HF.filter_gan_masks(img_dir, msk_dir, out_dir, do_watershed_and_four_connectivity=False, dark_background=False)
    out = np.array(Image.open(os.path.join(out_dir, "t.tif")))
    # (a's dark hole has a contour of its own, which IS darker than the threshold: it survives and is drawn -- OpenCV semantics)
    assert np.all(out[c] == 255) and np.all(out[a & ~disc(24, 24, 7)] == 0) and np.all(out[disc(24, 24, 4)] == 255)


def _filter_one(tmp_path, mask, img, **kw):
    from PIL import Image
    img_dir, msk_dir, out_dir = (str(tmp_path / d) for d in ("img", "msk", "out"))
    for d in (img_dir, msk_dir):
        os.makedirs(d, exist_ok=True)
    Image.fromarray(mask.astype(np.uint8)).save(os.path.join(msk_dir, "t.tif"))
    Image.fromarray(img.astype(np.uint8)).save(os.path.join(img_dir, "t.tif"))
    HF.filter_gan_masks(img_dir, msk_dir, out_dir, do_watershed_and_four_connectivity=False, threshold_method=lambda im: 100.0, **kw)
    return np.array(Image.open(os.path.join(out_dir, "t.tif")))


def test_contour_statistics_follow_the_opencv_semantics_of_measure(tmp_path):
    """Measure's contour list / mean intensities / filled drawing (Measurements.py:158-191,321-342,569-611; HelperFunctions.py:169-178)
    on hand-built cases -- what cv2.findContours(RETR_TREE), pointPolygonTest >= 0 and drawContours(-1, thickness=-1) do:
    every border is a contour (holes too), a contour's mean runs over the points inside or ON its polygon, contours with fewer than
    5 vertices and perimeter < 8 are dropped, and the kept contours are filled together with the even-odd rule."""
    mask = np.zeros((40, 64), np.uint8)
    img = np.full((40, 64), 20.0)
    # A: 9x9 square with a DARK 3x3 hole: outer kept (bright), hole contour dropped (hole dark, ring bright: mean < 100) -> hole filled
    mask[2:11, 2:11] = 255; mask[5:8, 5:8] = 0
    img[2:11, 2:11] = 150; img[5:8, 5:8] = 10
    # B: same with a BRIGHT hole: hole contour kept -> the hole stays a hole (even-odd fill), its ring is drawn
    mask[2:11, 14:23] = 255; mask[5:8, 17:20] = 0
    img[2:11, 14:23] = 200; img[5:8, 17:20] = 220
    # C: one-pixel hole, bright: its contour (a 4-vertex diamond, perimeter 4 sqrt 2 < 8) is dropped for its SIZE -> filled
    mask[2:9, 26:33] = 255; mask[5, 29] = 0
    img[2:9, 26:33] = 200; img[5, 29] = 250
    # D: specks: 2x2 (4 vertices, perimeter 4: dropped) and 3x3 (4 vertices, perimeter 8: kept), both bright
    mask[14:16, 2:4] = 255; img[14:16, 2:4] = 250
    mask[14:17, 8:11] = 255; img[14:17, 8:11] = 250
    # E: two 4x4 squares touching only diagonally = ONE 8-connected particle; one half is dark, the mean over both decides: kept
    mask[20:24, 2:6] = 255; mask[24:28, 6:10] = 255
    img[20:24, 2:6] = 250; img[24:28, 6:10] = 60          # mean (16 * 250 + 16 * 60) / 32 = 155 >= 100
    # F: a particle cut by the image border is kept (excludeEdges=False)
    mask[30:40, 54:64] = 255; img[30:40, 54:64] = 180
    # G: a DARK ring around a BRIGHT hole: the outer contour goes (mean < 100), the hole's contour stays -> ring drawn, hole filled
    mask[14:23, 30:39] = 255; mask[16:21, 32:37] = 0
    img[14:23, 30:39] = 30; img[16:21, 32:37] = 240
    cs = HF.find_contours(mask)
    kinds = sorted((c["kind"], int(c["region"].sum())) for c in cs)
    # outer contours: A 81, B 81, C 49, 3x3 speck 9, E 32, F 100, G 81; holes (hole + ring): A/B 9 + 12 = 21, G 25 + 20 = 45
    assert kinds == sorted([("outer", 81), ("outer", 81), ("outer", 49), ("outer", 9), ("outer", 32), ("outer", 100), ("outer", 81),
                            ("hole", 21), ("hole", 21), ("hole", 45)]), kinds
    means = dict(zip([(c["kind"], c["slice"][0].start, c["slice"][1].start) for c in cs], HF.contour_mean_intensities(cs, img)))
    assert means[("outer", 2, 2)] == pytest.approx((72 * 150 + 9 * 10) / 81)              # outer contour of A: the hole's pixels count
    assert means[("hole", 4, 4)] == pytest.approx((12 * 150 + 9 * 10) / 21)               # hole contour of A: hole + its 4-adjacent ring
    out = _filter_one(tmp_path, mask, img)
    exp = np.zeros_like(mask)
    exp[2:11, 2:11] = 255                                      # A with its hole filled
    exp[2:11, 14:23] = 255; exp[5:8, 17:20] = 0                # B keeps its hole
    exp[2:9, 26:33] = 255                                      # C filled
    exp[14:17, 8:11] = 255                                     # the 3x3 speck (the 2x2 one is gone)
    exp[20:24, 2:6] = 255; exp[24:28, 6:10] = 255              # E whole
    exp[30:40, 54:64] = 255                                    # F
    ring = np.zeros_like(mask, bool); ring[15, 32:37] = ring[21, 32:37] = True; ring[16:21, 31] = ring[16:21, 37] = True
    exp[ring] = 255; exp[16:21, 32:37] = 255                   # G: the hole contour's border (4-adjacent ring) + its inside
    assert np.array_equal(out, exp), np.argwhere(out != exp)[:10]
    # filterResults' shortcut: a threshold of exactly 0 filters nothing (Measurements.py:580-581)
    from PIL import Image
    HF.filter_gan_masks(str(tmp_path / "img"), str(tmp_path / "msk"), str(tmp_path / "out0"), do_watershed_and_four_connectivity=False,
                        threshold_method=lambda im: 0)
    out0 = np.array(Image.open(str(tmp_path / "out0" / "t.tif")))
    assert np.array_equal(out0 > 0, HF.draw_contours_filled(cs, mask.shape) > 0)


def test_filter_gan_masks_in_worker_processes_writes_the_same_files(tmp_path):
    """The pairs of step 5 are independent: filtered in worker processes they are the files of the plain loop (both passes of the workflow:
    without and with watershed + 4-connectivity), and a threshold function that cannot travel to a worker keeps the call inline."""
    from PIL import Image
    rng = np.random.default_rng(3)
    img_dir, msk_dir = str(tmp_path / "img"), str(tmp_path / "msk")
    os.makedirs(img_dir), os.makedirs(msk_dir)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in range(20):
        mask = np.zeros((64, 64), np.uint8)
        img = (rng.random((64, 64)) * 60).astype(np.uint8)
        for _ in range(6):
            cy, cx, r = rng.integers(8, 56), rng.integers(8, 56), rng.integers(3, 8)
            disc = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
            mask[disc] = 255
            if rng.random() < 0.6:
                img[disc] = 200
        Image.fromarray(img).save(os.path.join(img_dir, f"{i:03d}.tif"))
        Image.fromarray(mask).save(os.path.join(msk_dir, f"{i:03d}.tif"))
    for ws in (False, True):
        outs = {}
        for workers in (1, 3):
            outs[workers] = str(tmp_path / f"out_{ws}_{workers}")
            HF.filter_gan_masks(img_dir, msk_dir, outs[workers], do_watershed_and_four_connectivity=ws, workers=workers)
        assert sorted(os.listdir(outs[1])) == sorted(os.listdir(outs[3])) and len(os.listdir(outs[1])) == 20
        changed = 0
        for f in os.listdir(outs[1]):
            a, b = np.array(Image.open(os.path.join(outs[1], f))), np.array(Image.open(os.path.join(outs[3], f)))
            assert np.array_equal(a, b), f
            changed += int(not np.array_equal(a, np.array(Image.open(os.path.join(msk_dir, f)))))
        assert changed > 0          # the filter did remove particles
    out_l = str(tmp_path / "out_lambda")
    HF.filter_gan_masks(img_dir, msk_dir, out_l, do_watershed_and_four_connectivity=False, threshold_method=lambda im: 100.0, workers=3)
    assert len(os.listdir(out_l)) == 20


def test_small_contour_removal_rule():
    """len(contour) < 5 and perimeter < 8 on the CHAIN_APPROX_SIMPLE polygon (Measurements.py:176-187): vertices / perimeters of the
    shapes that can qualify (everything inside a 4 x 4 box)."""
    cases = {"pixel": ([[1]], 1, 0.0), "pair": ([[1, 1]], 2, 2.0), "line4": ([[1, 1, 1, 1]], 2, 6.0), "2x2": ([[1, 1], [1, 1]], 4, 4.0),
             "3x3": (np.ones((3, 3)), 4, 8.0), "plus": ([[0, 1, 0], [1, 1, 1], [0, 1, 0]], 4, 4 * 2 ** 0.5),
             "tromino": ([[1, 0], [1, 1]], 3, 2 + 2 ** 0.5), "2x3": (np.ones((2, 3)), 4, 6.0),
             "3x3 less a corner": ([[1, 1, 1], [1, 1, 1], [1, 1, 0]], 5, 6 + 2 ** 0.5), "diagonal pair": ([[1, 0], [0, 1]], 2, 2 * 2 ** 0.5)}
    for name, (shape, nv, perim) in cases.items():
        got = HF._small_polygon(np.asarray(shape, bool))
        assert got[0] == nv and got[1] == pytest.approx(perim), (name, got)
    m = np.zeros((12, 40), np.uint8)
    m[2, 2] = m[2:4, 6:8] = m[2, 12:16] = 255                  # dropped: pixel, 2x2, 4-pixel line (2 points, perimeter 6)
    m[6:9, 2:5] = m[6, 8:13] = 255                             # kept: 3x3 (perimeter 8), 5-pixel line (perimeter 8)
    kept = sorted(int(c["region"].sum()) for c in HF.find_contours(m))
    assert kept == [5, 9]
    # the smallness test is the `elif` of the edge test (Measurements.py:170-187): a speck with a vertex on the first / last row or
    # column is never tested, and kept when excludeEdges=False -- here a pixel in a corner, a 2x2 on the bottom row, a pair on the
    # right column; the same shapes one pixel inside are dropped
    e = np.zeros((12, 40), np.uint8)
    e[0, 0] = e[10:12, 6:8] = e[4:6, 39] = 255                 # on the border: kept (1, 4, 2 pixels)
    e[1, 3] = e[8:10, 12:14] = e[4:6, 37] = 255                # one pixel inside: dropped
    kept = sorted(int(c["region"].sum()) for c in HF.find_contours(e))
    assert kept == [1, 2, 4], kept
    out = HF.draw_contours_filled(HF.find_contours(e), e.shape)
    exp = np.zeros_like(e); exp[0, 0] = exp[10:12, 6:8] = exp[4:6, 39] = 255
    assert np.array_equal(out, exp)


def test_iou_and_roc_known_answers():
    a = np.zeros((10, 10), np.uint8); a[2:6, 2:6] = 1            # 16 pixels
    b = np.zeros((10, 10), np.uint8); b[4:8, 4:8] = 1            # 16 pixels, 4 shared
    assert SC.whole_image_iou(a, b) == pytest.approx(4 / 28)
    tpr, tnr, fpr, fnr = SC.roc(a, b)
    assert (tpr, fnr) == (pytest.approx(4 / 16), pytest.approx(12 / 16)) and tnr == pytest.approx(72 / 84) and fpr == pytest.approx(12 / 84)
    # instances: two squares in the prediction, one matching a ground-truth square exactly, one unmatched
    p = np.zeros((32, 32), np.uint8); g = np.zeros((32, 32), np.uint8)
    p[2:10, 2:10] = 1; g[2:10, 2:10] = 1
    p[20:28, 20:28] = 1
    assert SC.instance_iou(p, g, 0) == pytest.approx(0.5)        # (1.0 + 0.0) / 2
    p[15, 15] = 1                                                # a one-pixel speck: polygon area 0, dropped by min_area 9 only
    assert SC.instance_iou(p, g, 9) == pytest.approx(0.5) and SC.instance_iou(p, g, 0) < 0.5 + 1e-9


def test_iou_sweep_reproduces_reference_slot_quirk():
    """calculateIoU accumulates threshold t into slot t - 1 and reports slot / 10 (Calculate_Scores.py:247-249,259-261)."""
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:64, 0:64]
    gt = ((yy - 30) ** 2 + (xx - 30) ** 2 <= 14 ** 2).astype(np.uint8) * 255
    pred = np.clip((gt > 0) * 0.55 + rng.normal(0, 0.03, gt.shape), 0, 1).astype(np.float32)      # foreground ~0.55, background ~0
    r = SC.calculate_iou([pred], [gt])
    assert r["iou_whole"] > 0.9
    # thresholds 0.1 .. 0.5 separate the classes; the first (lowest) best true threshold is 0.1, reported by the reference as slot 0 -> 0.0
    assert r["best_threshold_whole_true"] == pytest.approx(r["best_threshold_whole"] + 0.1)


def test_data_root_scoring_command_on_a_dataset_directory(tmp_path, capsys):
    """The one-command hook for a MOUNTED dataset (missing 2): `python -m <package>.Scoring --data-root ... --predictions ...`
    pairs <id>_m.tif ground truths with <id>_raw.tif / <id>.tif predictions in the reference's Datasets/ layout, crops the SEM info bar
    rows and prints calculate_iou's figures as one JSON line."""
    import json
    from PIL import Image
    S = importlib.import_module(BASE + ".Scoring")
    img_dir, gt_dir, pred_dir = tmp_path / S.IMAGES_SUBDIR, tmp_path / S.GROUND_TRUTH_SUBDIR, tmp_path / "pred"
    for d in (img_dir, gt_dir, pred_dir):
        d.mkdir(parents=True)
    yy, xx = np.mgrid[0:96, 0:128]
    for k, ident in enumerate(("1908248", "1908250")):
        gt = (((yy - 30) ** 2 + (xx - 40 - 20 * k) ** 2 < 15 ** 2) | ((yy - 60) ** 2 + (xx - 90) ** 2 < 12 ** 2)).astype(np.uint8) * 255
        gt[80:] = 255                                   # "info bar" rows: must be cropped away, the prediction has nothing there
        prob = np.where(gt > 0, 0.9, 0.05).astype(np.float32)
        prob[80:] = 0.0
        prob[28:33, 38 + 20 * k:43 + 20 * k] = 0.05     # a small defect: IoU < 1
        Image.fromarray(gt).save(gt_dir / f"{ident}_m.tif")
        Image.fromarray(prob).save(pred_dir / f"{ident}_raw.tif")
        Image.fromarray(np.zeros((96, 128), np.uint8)).save(img_dir / f"{ident}.tif")
    res = S.main(["--data-root", str(tmp_path), "--predictions", str(pred_dir), "--crop-rows", "80", "--no-watershed"])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["images"] == 2 and line["crop_rows"] == 80 and abs(line["iou_whole"] - res["iou_whole"]) < 1e-12
    assert 0.9 < res["iou_whole"] < 1.0 and res["best_threshold_whole_true"] in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8)
    with pytest.raises(SystemExit):
        S.main(["--data-root", str(tmp_path / "nowhere"), "--predictions", str(pred_dir)])


def test_scoring_matches_the_references_own_functions():
    """Scoring.whole_image_iou / roc / the polygon-area estimate of Scoring.instances against `calculateWholeImageIoU`, `ROC` and
    `polygon_area` EXECUTED from the reference (Calculate_Scores.py:69-70,107-151; tests/golden/make_scoring_goldens.py lifts the three
    definitions out of the module's syntax tree and runs them on the committed inputs)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scoring_goldens.npz"))
    for k in range(int(z["n_pairs"])):
        a, b = z[f"pair{k}_a"], z[f"pair{k}_b"]
        assert float(SC.whole_image_iou(a, b)) == float(z[f"pair{k}_iou"]), k           # the same two integer sums: exact
        assert np.array_equal(np.array(SC.roc(a, b), np.float64), z[f"pair{k}_roc"]), k
    for i in range(int(z["n_shapes"])):
        inst = SC.instances(z[f"shape{i}"])
        assert len(inst) == 1
        # the shoelace area of the border-pixel polygon, as the reference computes it from the contour vertices, against the cell
        # count Scoring derives from pixel sets (no cv2 in the image) -- incl. a one-pixel speck and a one-pixel-wide line (area 0)
        assert inst[0][2] == pytest.approx(float(z[f"shape{i}_area"]), abs=1e-9), (i, inst[0][2], float(z[f"shape{i}_area"]))
