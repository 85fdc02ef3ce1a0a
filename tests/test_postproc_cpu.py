"""Post-processing row: Otsu -> EDT -> gaussian -> peak_local_max -> marker watershed with lines -> 8-to-4
connectivity, against vectors produced by the REFERENCE's own Measurements.Measure.segment / HelperFunctions.segment
running on scikit-image 0.18.3 + scipy 1.7.1 (tests/golden/make_postproc_goldens.py).  Integer label maps: bit-exact."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
from scipy import ndimage

BASE = "automatic-sem-image-segmentation_amd"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HF = importlib.import_module(BASE + ".HelperFunctions")
CASES = range(6)


@pytest.fixture(scope="module")
def gold(golden_dir):
    if not os.path.exists(os.path.join(REPO, BASE, "libsemseg_post.so")):
        import __graft_entry__ as g
        g.build()
    return np.load(os.path.join(golden_dir, "postproc_segment.npz"))


def test_post_library_exports_declared_symbols(gold):
    header = open(os.path.join(REPO, "include", "semseg_post.h")).read()
    declared = set(re.findall(r"\b(ss_post_[a-z0-9_]+)\s*\(", header))
    lib = ctypes.CDLL(os.path.join(REPO, BASE, "libsemseg_post.so"))
    assert declared == {"ss_post_version", "ss_post_watershed", "ss_post_eight_to_four"}
    for name in declared:
        assert hasattr(lib, name)


@pytest.mark.parametrize("i", CASES)
def test_otsu_and_peaks_match_skimage(gold, i):
    img = gold[f"c{i}_image"]
    assert HF.threshold_otsu(img) == float(gold[f"c{i}_otsu"])
    peaks = HF.peak_local_max(gold[f"c{i}_distance"], int(gold[f"c{i}_min_distance"]))
    np.testing.assert_array_equal(peaks, gold[f"c{i}_peaks"])           # same peaks, same (highest-first) order


@pytest.mark.parametrize("i", CASES)
def test_watershed_bit_exact_on_reference_distance_map(gold, i):
    """The flooding itself: same distance map, same markers -> identical label support (the reference keeps labels > 0)."""
    img = gold[f"c{i}_image"]
    mask = img > gold[f"c{i}_otsu"]
    seeds = np.zeros(img.shape, np.uint8)
    seeds[tuple(gold[f"c{i}_peaks"].T)] = 1
    labels = HF.watershed(-gold[f"c{i}_distance"], ndimage.label(seeds)[0], mask=mask, watershed_line=True)
    np.testing.assert_array_equal(((labels > 0) * 255).astype(np.uint8), gold[f"c{i}_seg_ws"])
    # without lines every masked pixel connected to a marker is labelled, and lines only ever remove pixels
    plain = HF.watershed(-gold[f"c{i}_distance"], ndimage.label(seeds)[0], mask=mask, watershed_line=False)
    assert np.all((labels > 0) <= (plain > 0)) and np.all((plain > 0) <= mask)


@pytest.mark.parametrize("i", CASES)
def test_segment_end_to_end(gold, i):
    """Whole Measure.segment / HelperFunctions.segment.  The distance map is recomputed with THIS interpreter's scipy, which
    differs from the generating scipy 1.7.1 in the last ulp (3.6e-15) -- enough to move a plateau tie by a pixel; so the
    end-to-end maps are required to agree on >= 99.9 % of the pixels (threshold-only maps: exactly)."""
    img, md = gold[f"c{i}_image"], int(gold[f"c{i}_min_distance"])
    np.testing.assert_array_equal(HF.segment_measure(img, -1, False, md, darkBackground=True), gold[f"c{i}_seg_nows"])
    for got, want in ((HF.segment_measure(img, -1, True, md, darkBackground=True), gold[f"c{i}_seg_ws"]),
                      (HF.segment_measure(img, 100, True, md, darkBackground=True), gold[f"c{i}_seg_fixed_thr"]),
                      (HF.segment(img, -1, True, md, use_four_connectivity=True), gold[f"c{i}_hf_segment"])):
        assert got.dtype == np.uint8 and set(np.unique(got)) <= {0, 255}
        assert np.count_nonzero(got != want) <= 1e-3 * got.size, np.count_nonzero(got != want)


def test_constant_mask_returns_before_watershed(gold):
    np.testing.assert_array_equal(HF.segment_measure(gold["flat_image"], 3, True, 9, darkBackground=True), gold["flat_seg"])


def test_eight_to_four_c_equals_python_loop():
    rng = np.random.default_rng(3)
    for shape in ((1, 1), (2, 2), (7, 9), (40, 33)):
        img = ((rng.random(shape) > 0.55) * 255).astype(np.uint8)
        want = HF.eight_to_four_connected(img.astype(np.int32))       # non-uint8 input takes the Python loop
        got = HF.eight_to_four_connected(img.copy())
        np.testing.assert_array_equal(got, want.astype(np.uint8))


def test_postproc_under_address_sanitizer(tmp_path):
    """The host-only C code (csrc/postproc.c: priority-flood watershed, 8 -> 4 connectivity) built with AddressSanitizer + UBSan
    (`make -C csrc asan`) and driven over the watershed / connectivity fixtures of this file plus edge shapes (1-pixel-wide images,
    all-foreground, all-background, markers on the border) in a child interpreter with libasan preloaded: any out-of-bounds access,
    use-after-free or leak report fails the test."""
    import subprocess
    import sys
    csrc = os.path.join(REPO, BASE, "csrc")
    subprocess.run(["make", "-C", csrc, "asan"], check=True, capture_output=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    code = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, %r)
HF = importlib.import_module(%r + ".HelperFunctions")
g = np.load(%r)
n = 0
for i in range(6):
    lab = HF.watershed(-g[f"c{i}_distance"], g[f"c{i}_markers"] if f"c{i}_markers" in g.files else None, mask=g[f"c{i}_mask"] if f"c{i}_mask" in g.files else None, watershed_line=True) if f"c{i}_markers" in g.files else None
    out = HF.segment_measure(g[f"c{i}_image"], -1.0, True, int(g[f"c{i}_min_distance"]), darkBackground=True)
    HF.eight_to_four_connected(out.copy())
    n += 1
rng = np.random.default_rng(0)
for h, w in ((1, 1), (1, 17), (17, 1), (2, 2), (3, 64), (33, 31)):
    for fill in (0.0, 0.5, 1.0):
        m = (rng.random((h, w)) < fill).astype(np.uint8) * 255
        HF.eight_to_four_connected(m.copy())
        d = rng.random((h, w))
        mk = np.zeros((h, w), np.int32)
        mk[0, 0] = 1
        mk[-1, -1] = 2
        for line in (False, True):
            lab = HF.watershed(d, mk, mask=(m > 0) if fill else None, watershed_line=line)
            assert lab.shape == (h, w)
        n += 1
print("ASAN_RUN_OK", n)
''' % (REPO, BASE, os.path.join(REPO, "tests", "golden", "postproc_segment.npz"))
    env = dict(os.environ, LD_PRELOAD=asan, SS_POST_LIB=os.path.join(REPO, BASE, "libsemseg_post_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="halt_on_error=1:exitcode=24")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ASAN_RUN_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
