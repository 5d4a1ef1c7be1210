/* ORACLE (test infrastructure, never shipped or measured as the product):
 * CPU restatement of the reference's sequential priority-flood kernel
 *   /root/reference/tobac_flow/_watershed.pyx:33-164 (heap), :222-344 (watershed_raveled)
 * for the only call path the reference uses (compactness = 0, wsl = False,
 * /root/reference/tobac_flow/watershed.py:59-61,151-164).
 *
 * Parity pinned by tests/golden/watershed_ref.npz (outputs of the reference itself) and,
 * when /root/reference is present, live against oracle/_ref (the compiled .pyx).
 *
 * The heap is a binary min-heap of POINTERS into a slab of items, exactly as the reference:
 * equal keys are never swapped (strict `smaller`), so the pop order of equal-keyed items is a
 * function of the array mechanics -- reproduced here by keeping the same sift rules.
 *
 * tie_mode 0: reference semantics (value, age) with age an int32 (wraps like the reference's
 *             `new_elem.age = age` store, _watershed.pyx:157,338).
 * tie_mode 1: "idealised" total order (value, age, push sequence) -- used by tests to
 *             characterise where the reference output depends on heap-internal order.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float value; int32_t age; int64_t index; int64_t source; int64_t seq; } item_t;
typedef struct { int64_t items, space; item_t *data; item_t **ptrs; int tie_mode; } heap_t;

static inline int smaller(const heap_t *h, const item_t *a, const item_t *b) {
    /* _watershed.pyx:161-164 */
    if (a->value != b->value) return a->value < b->value;
    if (h->tie_mode == 1 && a->age == b->age) return a->seq < b->seq;
    return a->age < b->age;
}
static inline void hswap(heap_t *h, int64_t a, int64_t b) {
    item_t *t = h->ptrs[a]; h->ptrs[a] = h->ptrs[b]; h->ptrs[b] = t;
}
static int heap_init(heap_t *h, int tie_mode) {
    /* _watershed.pyx:39-49 */
    h->items = 0; h->space = 1000; h->tie_mode = tie_mode;
    h->data = (item_t *)malloc(h->space * sizeof(item_t));
    h->ptrs = (item_t **)malloc(h->space * sizeof(item_t *));
    if (!h->data || !h->ptrs) return -1;
    for (int64_t k = 0; k < h->space; k++) h->ptrs[k] = h->data + k;
    return 0;
}
static void heap_pop(heap_t *h, item_t *dest) {
    /* _watershed.pyx:67-111 */
    *dest = *h->ptrs[0];
    h->items -= 1;
    if (h->items == 0) return;
    hswap(h, 0, h->items);
    int64_t i = 0, smallest = 0;
    for (;;) {
        int64_t l = 2 * i + 1, r = 2 * i + 2;
        if (l < h->items) {
            if (smaller(h, h->ptrs[l], h->ptrs[i])) smallest = l;
            if (r < h->items && smaller(h, h->ptrs[r], h->ptrs[smallest])) smallest = r;
        } else break;
        if (smallest == i) break;
        hswap(h, i, smallest);
        i = smallest;
    }
}
static int heap_push(heap_t *h, const item_t *e) {
    /* _watershed.pyx:120-152 */
    int64_t child = h->items;
    if (h->items == h->space) {
        h->space *= 2;
        item_t *nd = (item_t *)realloc(h->data, h->space * sizeof(item_t));
        item_t **np = (item_t **)realloc(h->ptrs, h->space * sizeof(item_t *));
        if (!nd || !np) return -1;
        h->ptrs = np;
        for (int64_t k = 0; k < h->items; k++) h->ptrs[k] = nd + (h->ptrs[k] - h->data);
        for (int64_t k = h->items; k < h->space; k++) h->ptrs[k] = nd + k;
        h->data = nd;
    }
    *h->ptrs[child] = *e;
    h->items += 1;
    while (child > 0) {
        int64_t parent = (child + 1) / 2 - 1;
        if (smaller(h, h->ptrs[child], h->ptrs[parent])) { hswap(h, parent, child); child = parent; }
        else break;
    }
    return 0;
}

/* Same twelve arguments as the reference entry point minus the dead ones (strides,
 * compactness, wsl are unused on the reference's call path); all arrays flat, C-contiguous.
 * Returns 0, or -1 on allocation failure.  `output` is mutated in place. */
int oracle_watershed_raveled(const float *image, const int64_t *marker_locations, int64_t n_markers,
                             const int64_t *structure, int64_t n_neighbors,
                             const int32_t *forward_offset, const int32_t *backward_offset,
                             const int32_t *forward_offset_locations,
                             const int32_t *backward_offset_locations,
                             const int8_t *mask, int32_t *output, int tie_mode,
                             int64_t *n_pops_out)
{
    heap_t hp;
    if (heap_init(&hp, tie_mode)) return -1;
    item_t elem, ne;
    int64_t age = 1, seq = 0, pops = 0;
    for (int64_t i = 0; i < n_markers; i++) {           /* :278-284 */
        int64_t index = marker_locations[i];
        elem.value = image[index]; elem.age = 0; elem.index = index; elem.source = index;
        elem.seq = seq++;
        if (heap_push(&hp, &elem)) return -1;
    }
    while (hp.items > 0) {                              /* :286-342 */
        heap_pop(&hp, &elem); pops++;
        for (int64_t i = 0; i < n_neighbors; i++) {
            int64_t nb = structure[i] + elem.index
                       + (int64_t)forward_offset_locations[i] * forward_offset[elem.index]
                       + (int64_t)backward_offset_locations[i] * backward_offset[elem.index];
            if (!mask[nb]) continue;
            if (output[nb]) continue;
            age += 1;
            ne.value = image[nb];
            output[nb] = output[elem.index];            /* label at push time, :337 */
            ne.age = (int32_t)age;                      /* Py_ssize_t -> int32 store, :338 */
            ne.index = nb; ne.source = elem.source; ne.seq = seq++;
            if (heap_push(&hp, &ne)) return -1;
        }
    }
    free(hp.data); free(hp.ptrs);
    if (n_pops_out) *n_pops_out = pops;
    return 0;
}
