/* Hand-written entry points of the stand-in library used by tests/test_launch_cpu.py::test_bench_rehearsal_at_eight_ranks (CPU only, no GPU):
 * "device memory" is host memory, kernels do nothing, and the calls whose results bench.py reads back (sizes, tables, prediction records, the
 * communicator's shape) return well-formed stand-ins.  Everything not defined here is generated as `return 0` from the prototypes of
 * include/odise_hip.h and include/odise_hip_tools.h (tests/stub_lib.py).  TEST INFRASTRUCTURE - never loaded by the product. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

#include "odise_hip.h"
#include "odise_hip_tools.h"

struct odise_hip_ctx { int device, rank, world; double t0; };
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

int odise_hip_create(int device, odise_hip_ctx** out) {
    odise_hip_ctx* c = (odise_hip_ctx*)calloc(1, sizeof(*c));
    c->device = device; c->rank = 0; c->world = 0;
    *out = c;
    return 0;
}
int odise_hip_destroy(odise_hip_ctx* ctx) { free(ctx); return 0; }
const char* odise_hip_last_error(void) { return "stub library"; }
int odise_hip_version(void) { return 100; }
/* lazily-zeroed anonymous mappings (a 558 MB score tensor nobody touches costs no memory); buffers up to 128 MB are filled with a
 * pseudo-random pattern of small floats so that read-backs of "kernel outputs" look like data to the host-side calibration */
int odise_hip_malloc(odise_hip_ctx* ctx, size_t bytes, void** dptr) {
    (void)ctx;
    size_t n = (bytes + 4095 + 16) & ~(size_t)4095;
    char* p = (char*)mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return -4;
    *(size_t*)p = n;
    if (bytes <= ((size_t)128 << 20)) {
        float* f = (float*)(p + 16);
        uint32_t s = 12345u + (uint32_t)bytes;
        for (size_t i = 0; i < bytes / 4; ++i) { s = s * 1664525u + 1013904223u; f[i] = ((int32_t)(s >> 8) % 2001 - 1000) * 1e-3f; }
    }
    *dptr = p + 16;
    return 0;
}
int odise_hip_free(odise_hip_ctx* ctx, void* dptr) { (void)ctx; if (dptr) { char* p = (char*)dptr - 16; munmap(p, *(size_t*)p); } return 0; }
int odise_hip_memcpy_h2d(odise_hip_ctx* ctx, void* d, const void* s, size_t n) { (void)ctx; memcpy(d, s, n); return 0; }
int odise_hip_memcpy_d2h(odise_hip_ctx* ctx, void* d, const void* s, size_t n) { (void)ctx; memcpy(d, s, n); return 0; }
int odise_hip_memset(odise_hip_ctx* ctx, void* d, int v, size_t n) { (void)ctx; memset(d, v, n); return 0; }
int odise_hip_timer_start(odise_hip_ctx* ctx) { ctx->t0 = now_ms(); return 0; }
int odise_hip_timer_stop(odise_hip_ctx* ctx, float* ms) { *ms = (float)(now_ms() - ctx->t0) + 1.0f; return 0; }
int odise_hip_device_info(odise_hip_ctx* ctx, char* name, int len, int* cus, size_t* hbm) {
    (void)ctx;
    if (name) snprintf(name, len, "stub gfx950 (no device)");
    if (cus) *cus = 256;
    if (hbm) *hbm = (size_t)288 << 30;
    return 0;
}
int odise_hip_get_option(odise_hip_ctx* ctx, int option, int64_t* value) { (void)ctx; (void)option; *value = 0; return 0; }
int odise_hip_probe_read(odise_hip_ctx* ctx, float* us, int cap, int* n) { (void)ctx; (void)us; (void)cap; *n = 0; return 0; }
int odise_hip_stage_timeline_read(odise_hip_ctx* ctx, char* names, int names_cap, float* g, double* h, int cap, int* n) {
    (void)ctx; (void)g; (void)h; (void)cap;
    if (names && names_cap) names[0] = 0;
    *n = 0;
    return 0;
}
int odise_hip_launch_log_read(odise_hip_ctx* ctx, int* out6, int cap, int* n) { (void)ctx; (void)out6; (void)cap; *n = 0; return 0; }
int odise_hip_last_tile(void) { return 7; }
int odise_hip_maskgen_info(odise_hip_ctx* ctx, int* q, int* c, double* macs) { (void)ctx; if (q) *q = 100; if (c) *c = 256; if (macs) *macs = 0.0; return 0; }
int odise_hip_head_forward(odise_hip_ctx* ctx, const float* const* f, int B, int Cin, int H4, int W4, float* pm, float* me, float* mp, float* ls) {
    (void)ctx; (void)f; (void)B; (void)Cin; (void)H4; (void)W4; (void)pm; (void)me; (void)mp;
    if (ls) *ls = 100.0f;
    return 0;
}
int odise_hip_prefetch_stats(odise_hip_ctx* ctx, int* a, int* b, int* c, int* d) { (void)ctx; if (a) *a = 0; if (b) *b = 0; if (c) *c = 0; if (d) *d = 0; return 0; }
/* one model call: a well-formed record (six segments in bands) and instance table (three entries) per image */
int odise_hip_infer(odise_hip_ctx* ctx, const odise_infer_desc* d) {
    (void)ctx;
    const odise_post_desc* p = &d->post;
    for (int b = 0; b < d->B; ++b) {
        const int h = p->out_hw ? p->out_hw[2 * b] : d->img_hw[2 * b], w = p->out_hw ? p->out_hw[2 * b + 1] : d->img_hw[2 * b + 1];
        const int64_t npix = (int64_t)h * w;
        if (p->panoptic_on && p->panoptic && p->panoptic[b]) {
            int32_t* rec = p->panoptic[b];
            for (int64_t i = 0; i < npix; ++i) rec[i] = 1 + (int32_t)(i * 6 / npix);
            rec[npix] = 6;
            for (int s = 0; s < 6; ++s) { rec[npix + 1 + 3 * s] = s + 1; rec[npix + 2 + 3 * s] = s < 3; rec[npix + 3 + 3 * s] = s * 7; }
        }
        if (p->instance_on && p->inst_table && p->inst_scores) {
            const int topk = p->topk > 0 ? p->topk : 100;
            int32_t* tb = p->inst_table + (size_t)b * (1 + 2 * topk);
            memset(tb, 0, sizeof(int32_t) * (1 + 2 * topk));
            tb[0] = 3;
            for (int i = 0; i < 3; ++i) { tb[1 + i] = i; tb[1 + topk + i] = i; p->inst_scores[(size_t)b * topk + i] = 0.9f - 0.1f * i; }
        }
        if (p->semantic_on && p->sem_argmax && p->sem_argmax[b]) { for (int64_t i = 0; i < npix; ++i) p->sem_argmax[b][i] = (int32_t)(i * 3 / npix); }
    }
    return 0;
}
/* the exchange: a communicator is its (rank, world); the gather puts this rank's slice where RCCL would */
int odise_hip_comm_unique_id(void* id128) { for (int i = 0; i < ODISE_COMM_ID_BYTES; ++i) ((unsigned char*)id128)[i] = (unsigned char)(i * 7 + 3); return 0; }
int odise_hip_comm_init(odise_hip_ctx* ctx, const void* id, int rank, int world) {
    for (int i = 0; i < ODISE_COMM_ID_BYTES; ++i)
        if (((const unsigned char*)id)[i] != (unsigned char)(i * 7 + 3)) return -1;      /* the id every rank receives is rank 0's */
    ctx->rank = rank; ctx->world = world;
    return 0;
}
int odise_hip_comm_info(odise_hip_ctx* ctx, int* rank, int* world) { if (rank) *rank = ctx->rank; if (world) *world = ctx->world; return 0; }
int odise_hip_allgather_predictions(odise_hip_ctx* ctx, const int32_t* local, int64_t count, int32_t* all) {
    memmove(all + (size_t)ctx->rank * count, local, (size_t)count * 4);
    return 0;
}
int odise_hip_allgather_records(odise_hip_ctx* ctx, const int32_t* local, int n, int max_records, int64_t len, int32_t* all) {
    int32_t* mine = all + (size_t)ctx->rank * max_records * len;
    if (n > 0 && local != mine) memmove(mine, local, (size_t)n * len * 4);
    for (int64_t i = (int64_t)n * len; i < (int64_t)max_records * len; ++i) mine[i] = -1;
    return 0;
}
