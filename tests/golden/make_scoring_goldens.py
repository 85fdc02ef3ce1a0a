"""Golden vectors of the publication's scoring helpers, produced by EXECUTING the reference's own function definitions:
`calculateWholeImageIoU` (:69-70), `ROC` (:107-137) and `polygon_area` (:139-151) of
`/root/reference/Archive/Other Scripts/Calculate_Scores.py`.  The module scans directories and needs cv2 / skimage at import, so the
three definitions are lifted out of its syntax tree (`ast`) and executed with numpy only -- no reference text is kept here or in the
fixture (arrays of inputs and outputs only).  Build container only (the reference is not on the GPU box):

    python tests/golden/make_scoring_goldens.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/Archive/Other Scripts/Calculate_Scores.py"
WANT = ("calculateWholeImageIoU", "ROC", "polygon_area")


def reference_functions():
    tree = ast.parse(open(REF).read())
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    assert sorted(d.name for d in defs) == sorted(WANT)
    ns = {"np": np}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)
    return ns


def border_polygon_of_rectilinear(shape_mask):
    """Vertices (x, y) of the outer border of a 4-connected, hole-free region traced through its border PIXEL CENTRES in order (what
    cv2.findContours(CHAIN_APPROX_NONE) lists for such a region): Moore neighbourhood tracing, clockwise in image coordinates."""
    m = np.pad(shape_mask.astype(bool), 1)
    ys, xs = np.nonzero(m)
    start = (ys[0], xs[np.nonzero(ys == ys[0])[0][0]])
    nb = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]
    pts, cur, d = [start], start, 6
    while True:
        for k in range(8):
            dd = (d + k) % 8
            ny, nx = cur[0] + nb[dd][0], cur[1] + nb[dd][1]
            if m[ny, nx]:
                cur, d = (ny, nx), (dd + 5) % 8
                break
        else:
            break
        if cur == start:
            break
        pts.append(cur)
    return np.array([p[1] - 1 for p in pts], float), np.array([p[0] - 1 for p in pts], float)


def main():
    f = reference_functions()
    rng = np.random.default_rng(2024)
    out = {}
    # whole-image IoU and ROC on random {0,1} pairs of several densities (incl. an all-background ground truth: the TPR / FNR guards)
    k = 0
    for h, w, pa, pb in ((17, 23, 0.3, 0.4), (32, 32, 0.05, 0.9), (9, 40, 0.5, 0.5), (24, 24, 0.7, 0.0), (12, 12, 1.0, 0.6)):
        a = (rng.random((h, w)) < pa).astype(np.uint8)
        b = (rng.random((h, w)) < pb).astype(np.uint8)
        if a.sum() + b.sum() == 0:
            a[0, 0] = 1
        out[f"pair{k}_a"], out[f"pair{k}_b"] = a, b
        out[f"pair{k}_iou"] = np.float64(f["calculateWholeImageIoU"](a, b))
        out[f"pair{k}_roc"] = np.array(f["ROC"](a, b), np.float64)
        k += 1
    out["n_pairs"] = np.int64(k)
    # polygon_area on the border polygons of rectilinear, hole-free regions (rectangles, an L, a plus, a staircase, a speck, a line)
    shapes = []
    s = np.zeros((20, 20), np.uint8); s[3:11, 4:17] = 1; shapes.append(s)
    s = np.zeros((20, 20), np.uint8); s[2:15, 2:6] = 1; s[11:15, 2:14] = 1; shapes.append(s)
    s = np.zeros((21, 21), np.uint8); s[8:13, 2:19] = 1; s[2:19, 8:13] = 1; shapes.append(s)
    s = np.zeros((16, 16), np.uint8)
    for i in range(5):
        s[2 + 2 * i:12, 2 + 2 * i:4 + 2 * i] = 1
    shapes.append(s)
    s = np.zeros((8, 8), np.uint8); s[4, 4] = 1; shapes.append(s)
    s = np.zeros((8, 12), np.uint8); s[3, 2:10] = 1; shapes.append(s)
    for i, s in enumerate(shapes):
        x, y = border_polygon_of_rectilinear(s)
        out[f"shape{i}"] = s
        out[f"shape{i}_x"], out[f"shape{i}_y"] = x, y
        out[f"shape{i}_area"] = np.float64(f["polygon_area"](x, y))
    out["n_shapes"] = np.int64(len(shapes))
    np.savez_compressed(os.path.join(HERE, "scoring_goldens.npz"), **out)
    print({k_: (float(v) if v.ndim == 0 else v.tolist()) for k_, v in out.items() if k_.endswith(("_iou", "_roc", "_area"))})


if __name__ == "__main__":
    main()
