// Host-side (CPU) post-processing kernels of the inference path -- plain C, no GPU: the reference does this step on the
// CPU as well (Measurements.py:263-305 via scikit-image, HelperFunctions.py:131-152).  Built into libsemseg_post.so.
//
// ss_post_watershed: marker-controlled watershed by priority flooding, the algorithm scikit-image 0.18 publishes for
// skimage.segmentation.watershed (Soille 1990 / CellProfiler): a min-heap ordered by (pixel value, time of entry); every
// pixel is labelled when it LEAVES the heap (watershed_line mode), a pixel whose already-labelled unmasked neighbours carry
// two different labels stays 0 (the watershed line) and does not propagate.  8-connectivity, neighbour visiting order as
// scikit-image 0.18.3 produces it for a 3x3 footprint (N, E, W, S, NW, NE, SW, SE) -- the order fixes the entry times and
// with them how plateaus are split.  Ties between MARKERS of equal value (all enter at time 0) are broken by raster
// index here; scikit-image's order for that case is an artefact of its heap.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double value;
    int64_t age;
    int64_t index;
    int64_t source;
} HeapItem;

typedef struct {
    HeapItem* a;
    int64_t n, cap;
} Heap;

static int smaller(const HeapItem* x, const HeapItem* y) {
    if (x->value != y->value) return x->value < y->value;
    if (x->age != y->age) return x->age < y->age;
    return x->index < y->index;
}

static int heap_push(Heap* h, const HeapItem* it) {
    if (h->n == h->cap) {
        int64_t nc = h->cap ? h->cap * 2 : 1024;
        HeapItem* na = (HeapItem*)realloc(h->a, (size_t)nc * sizeof(HeapItem));
        if (!na) return -1;
        h->a = na;
        h->cap = nc;
    }
    int64_t c = h->n++;
    while (c > 0) {
        int64_t p = (c - 1) / 2;
        if (!smaller(it, &h->a[p])) break;
        h->a[c] = h->a[p];
        c = p;
    }
    h->a[c] = *it;
    return 0;
}

static void heap_pop(Heap* h, HeapItem* out) {
    *out = h->a[0];
    HeapItem last = h->a[--h->n];
    int64_t k = 0;
    for (;;) {
        int64_t c = 2 * k + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && smaller(&h->a[c + 1], &h->a[c])) ++c;
        if (!smaller(&h->a[c], &last)) break;
        h->a[k] = h->a[c];
        k = c;
    }
    if (h->n > 0) h->a[k] = last;
}

// image: h*w float64; markers: h*w int32 (0 = none); mask: h*w uint8 (0 = outside) or NULL; out: h*w int32 labels.
// returns 0, or -1 on allocation failure / bad arguments.
int ss_post_watershed(const double* image, const int32_t* markers, const uint8_t* mask, int h, int w, int watershed_line,
                      int32_t* out) {
    if (!image || !markers || !out || h <= 0 || w <= 0) return -1;
    const int64_t W = (int64_t)w + 2, H = (int64_t)h + 2;
    const int64_t nb[8] = {-W, 1, -1, W, -W - 1, -W + 1, W - 1, W + 1};
    double* img = (double*)calloc((size_t)(H * W), sizeof(double));
    int32_t* lab = (int32_t*)calloc((size_t)(H * W), sizeof(int32_t));
    uint8_t* msk = (uint8_t*)calloc((size_t)(H * W), 1);
    Heap hp = {0, 0, 0};
    int rc = -1;
    if (!img || !lab || !msk) goto done;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int64_t q = (int64_t)(y + 1) * W + x + 1, i = (int64_t)y * w + x;
            img[q] = image[i];
            lab[q] = markers[i];
            msk[q] = mask ? (mask[i] != 0) : 1;
        }
    for (int64_t q = 0; q < H * W; ++q)
        if (lab[q]) {
            HeapItem it = {img[q], 0, q, q};
            if (heap_push(&hp, &it)) goto done;
        }
    int64_t age = 1;
    while (hp.n > 0) {
        HeapItem e;
        heap_pop(&hp, &e);
        if (watershed_line) {
            if (lab[e.index] && e.index != e.source) continue;      // reached earlier from another neighbour
            int32_t l0 = 0;
            int differ = 0;
            for (int k = 0; k < 8; ++k) {
                const int64_t q = e.index + nb[k];
                if (!msk[q]) continue;
                if (!l0) l0 = lab[q];
                else if (lab[q] && lab[q] != l0) { differ = 1; break; }
            }
            if (differ) continue;                                   // stays 0: watershed line
            lab[e.index] = lab[e.source];
        }
        for (int k = 0; k < 8; ++k) {
            const int64_t q = e.index + nb[k];
            if (!msk[q] || lab[q]) continue;
            ++age;
            if (!watershed_line) lab[q] = lab[e.index];            // plain mode: label at push time
            HeapItem it = {img[q], age, q, e.source};
            if (heap_push(&hp, &it)) goto done;
        }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) out[(int64_t)y * w + x] = lab[(int64_t)(y + 1) * W + x + 1];
    rc = 0;
done:
    free(img);
    free(lab);
    free(msk);
    free(hp.a);
    return rc;
}

// HelperFunctions.py:131-152: break diagonal-only contacts of a binary uint8 image IN PLACE, in the reference's scan order
// (rows outer, columns inner; each 2x2 window sees the edits of the previous ones).
int ss_post_eight_to_four(uint8_t* img, int h, int w) {
    if (!img || h <= 0 || w <= 0) return -1;
    int64_t nz = 0;
    for (int64_t i = 0; i < (int64_t)h * w; ++i) nz += img[i] != 0;
    if (!(nz > 2 || nz < (int64_t)h * w - 2)) return 0;
    for (int x = 0; x < h - 1; ++x)
        for (int y = 0; y < w - 1; ++y) {
            uint8_t* a = img + (int64_t)x * w + y;        // a[0]=(x,y) a[1]=(x,y+1) a[w]=(x+1,y) a[w+1]=(x+1,y+1)
            if (a[0] == 0 && a[w + 1] == 0 && a[w] != 0 && a[1] != 0) a[w] = 0;
            else if (a[w] == 0 && a[1] == 0 && a[0] != 0 && a[w + 1] != 0) a[0] = 0;
        }
    return 0;
}

int ss_post_version(void) { return 1; }
