/* CPU-side sanitizer run of the HOST half of libdrnmf.
 *
 * Linked against dr-nmf_amd/build_asan/libdrnmf_asan.so -- every translation unit's host code built with
 * the address + undefined-behaviour sanitizers (tools/sanitize/build_sanitized.py) -- and run on a host WITHOUT a
 * GPU: every size query, descriptor validator, layout rule (under the tuning variables that switch
 * their branches) and argument-check / error path of the C ABI, through a handle bound to no device
 * (drnmf_create_unbound).  Nothing here reaches a kernel launch: every compute entry point is called
 * with arguments its own checks must refuse.  Exit code 0 = every expectation held and no sanitizer
 * report was printed (a report aborts the process: no-recover build, halt_on_error). */
#define _POSIX_C_SOURCE 200112L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "drnmf.h"

static int g_fail = 0, g_calls = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        ++g_calls;                                                                   \
        if (!(cond)) { ++g_fail; fprintf(stderr, "%s:%d: FAILED %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)
/* a call that must be refused: negative status and a message on the handle */
#define REFUSED(h, call)                                                             \
    do {                                                                             \
        const int32_t rc_ = (call);                                                  \
        ++g_calls;                                                                   \
        if (rc_ >= 0 || drnmf_last_error(h) == NULL || drnmf_last_error(h)[0] == 0) { \
            ++g_fail;                                                                \
            fprintf(stderr, "%s:%d: %s returned %d (expected an error + message)\n", __FILE__, __LINE__, #call, (int)rc_); \
        }                                                                            \
    } while (0)

static drnmf_cell_desc_t cell_desc(int B, int T, int F, int N, int K, int untied, int f16, int div, int ah) {
    drnmf_cell_desc_t d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.T = T; d.F = F; d.N = N; d.K = K;
    d.n_D = untied ? K : 1; d.n_alph = untied ? K : 1; d.alph_len = 1; d.n_lam = 1;
    d.return_all_hidden = ah; d.operand_f16 = f16; d.divergence = div;
    return d;
}

static void size_queries(void) {
    static const int Bs[] = {1, 15, 16, 17, 32, 64, 130, 191, 192, 250, 256, 512, 1024};
    static const int Fs[] = {1, 5, 16, 21, 33, 257, 513, 514, 1025};
    static const int Ns[] = {2, 34, 200, 500, 512, 2000, 8000};
    static const int Ks[] = {1, 2, 5, 25, 50};
    size_t total = 0;
    for (size_t ib = 0; ib < sizeof(Bs) / sizeof(*Bs); ++ib)
        for (size_t jf = 0; jf < sizeof(Fs) / sizeof(*Fs); ++jf)
            for (size_t kn = 0; kn < sizeof(Ns) / sizeof(*Ns); ++kn)
                for (size_t lk = 0; lk < sizeof(Ks) / sizeof(*Ks); ++lk)
                    for (int var = 0; var < 6; ++var) {
                        /* var: 0 tied fp32, 1 untied fp32, 2 untied all-hidden (training), 3 fp16 operands,
                         * 4 KL cell all-hidden, 5 beta cell */
                        const int T = (ib + jf + kn) % 3 == 0 ? 500 : 7;
                        drnmf_cell_desc_t d = cell_desc(Bs[ib], T, Fs[jf], Ns[kn], Ks[lk], var >= 1,
                                                        var == 3, var == 4 ? DRNMF_DIV_KL : (var == 5 ? DRNMF_DIV_BETA : 0),
                                                        var == 2 || var == 4);
                        const size_t pb = drnmf_params_bytes(&d), wb = drnmf_cell_workspace_bytes(&d);
                        const size_t bb = drnmf_cell_backward_workspace_bytes(&d);
                        const int lp = drnmf_cell_launches_per_frame(&d);
                        CHECK(pb > 0 && wb > 0 && lp > 0);
                        CHECK(pb % 4 == 0);
                        total += pb + wb + bb;
                        drnmf_dense_desc_t dd;
                        memset(&dd, 0, sizeof(dd));
                        dd.B = d.B; dd.T = d.T; dd.F = d.F; dd.N = d.N > 2000 ? 2000 : d.N; dd.K = d.K;
                        dd.connect_input = var & 1; dd.activation = var % 6; dd.return_all_hidden = var == 2;
                        dd.operand_f16 = var == 3;
                        total += drnmf_dense_params_bytes(&dd) + drnmf_dense_workspace_bytes(&dd) +
                                 drnmf_dense_backward_workspace_bytes(&dd);
                    }
    CHECK(total > 0);
    /* degenerate descriptors: size queries answer 0, never crash */
    drnmf_cell_desc_t z = cell_desc(0, 0, 0, 0, 0, 0, 0, 0, 0);
    CHECK(drnmf_cell_workspace_bytes(&z) == 0);
    CHECK(drnmf_cell_workspace_bytes(NULL) == 0);
    CHECK(drnmf_cell_launches_per_frame(&z) == 0);
    CHECK(drnmf_cell_launches_per_frame(NULL) == 0);
    for (long long rows = 1; rows <= 4000000; rows *= 13) {
        CHECK(drnmf_loss_head_workspace_bytes(rows, 513, 1000) > 0);
        CHECK(drnmf_loss_forward_workspace_bytes(rows) > 0);
        CHECK(drnmf_ista_workspace_bytes(rows, 513, 2000) > 0);
        CHECK(drnmf_mu_workspace_bytes(rows, 257, 200) > 0);
        CHECK(drnmf_snmf_train_workspace_bytes(rows, 513, 1000) > 0);
    }
    CHECK(drnmf_wav_int16_workspace_bytes() > 0);
    for (int F = 1; F < 1100; F += 37) CHECK(drnmf_padded_f(F) >= F && drnmf_padded_f(F) % 4 == 0);
    for (int N = 64; N <= 4096; N *= 2) {
        CHECK(drnmf_stft_frames(160000, N, N / 4) > 0);
        CHECK(drnmf_istft_workspace_bytes(3, 100, N) > 0);
    }
    CHECK(drnmf_stft_frames(0, 512, 128) >= 0);
    CHECK(drnmf_stft_frames(100, 512, 128) >= 0);         /* shorter than one window */
    CHECK(drnmf_sdr_workspace_bytes(4, 160000, 512) > 0);
}

static void refused_calls(drnmf_handle_t h) {
    float f = 0.f;
    float* nf = NULL;
    char buf[256];
    void* aligned = (void*)(((size_t)buf + 255) & ~(size_t)255);   /* (never dereferenced) */
    (void)aligned;
    drnmf_cell_desc_t ok = cell_desc(4, 8, 65, 32, 3, 1, 0, 0, 0);
    drnmf_cell_desc_t bad = ok;
    /* descriptor validation, one broken field at a time */
    bad.B = 0;
    REFUSED(h, drnmf_cell_forward(h, &bad, &f, -1.f, &f, &f, 1.f, 0.f, 0.f, &f, &f, 1 << 20, NULL));
    bad = ok; bad.n_D = 2;
    REFUSED(h, drnmf_cell_forward(h, &bad, &f, -1.f, &f, &f, 1.f, 0.f, 0.f, &f, &f, 1 << 20, NULL));
    bad = ok; bad.alph_len = 7;
    REFUSED(h, drnmf_prepare_params(h, &bad, &f, &f, &f, &f, NULL));
    bad = ok; bad.operand_f16 = 2;
    REFUSED(h, drnmf_prepare_params(h, &bad, &f, &f, &f, &f, NULL));
    bad = ok; bad.divergence = 9;
    REFUSED(h, drnmf_cell_forward(h, &bad, &f, -1.f, &f, &f, 1.f, 0.f, 0.f, &f, &f, 1 << 20, NULL));
    bad = ok; bad.divergence = DRNMF_DIV_KL; bad.operand_f16 = 1;
    REFUSED(h, drnmf_cell_forward_ista(h, &bad, &f, -1.f, &f, &f, 1.5f, NULL, NULL, &f, &f, 1 << 20, NULL));
    bad = ok; bad.B = 1 << 20; bad.T = 1 << 20;
    REFUSED(h, drnmf_cell_forward(h, &bad, &f, -1.f, &f, &f, 1.f, 0.f, 0.f, &f, &f, 1 << 20, NULL));
    REFUSED(h, drnmf_cell_forward(h, NULL, &f, -1.f, &f, &f, 1.f, 0.f, 0.f, &f, &f, 1 << 20, NULL));
    /* NULL data pointers, short workspaces, misaligned blocks */
    REFUSED(h, drnmf_prepare_params(h, &ok, nf, nf, nf, NULL, NULL));
    REFUSED(h, drnmf_cell_forward(h, &ok, nf, -1.f, NULL, nf, 1.f, 0.f, 0.f, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_cell_forward(h, &ok, &f, -1.f, aligned, &f, 1.f, 0.f, 0.f, &f, aligned, 16, NULL));
    REFUSED(h, drnmf_cell_forward(h, &ok, &f, -1.f, (char*)aligned + 4, &f, 1.f, 0.f, 0.f, &f, (char*)aligned + 4,
                                  drnmf_cell_workspace_bytes(&ok), NULL));
    REFUSED(h, drnmf_cell_forward_stateful(h, &ok, nf, -1.f, NULL, nf, 1.f, 0.f, 0.f, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_cell_profile(h, &ok, nf, -1.f, NULL, nf, 1.f, 0.f, 0.f, nf, NULL, 0, NULL, 2, &f));
    bad = ok; bad.divergence = DRNMF_DIV_KL;
    REFUSED(h, drnmf_cell_forward(h, &bad, &f, -1.f, aligned, &f, 1.f, 0.f, 0.f, &f, aligned, 1 << 20, NULL));
    REFUSED(h, drnmf_cell_forward_ista(h, &ok, &f, -1.f, aligned, &f, 1.5f, NULL, NULL, &f, aligned, 1 << 20, NULL));
    REFUSED(h, drnmf_cell_forward_ista(h, &bad, nf, -1.f, NULL, nf, 1.5f, NULL, NULL, nf, NULL, 0, NULL));
    bad = ok; bad.return_all_hidden = 1;
    REFUSED(h, drnmf_cell_backward(h, &bad, nf, NULL, nf, 1.f, 0.f, 0.f, nf, nf, NULL, 0, NULL, 0, nf, nf, nf, nf, NULL));
    REFUSED(h, drnmf_cell_backward(h, &ok, &f, aligned, &f, 1.f, 0.f, 0.f, &f, &f, aligned, 1 << 20, aligned, 1 << 20,
                                   &f, &f, &f, &f, NULL));        /* needs the all-hidden forward */
    REFUSED(h, drnmf_cell_backward_ista(h, &bad, nf, NULL, nf, 1.5f, nf, nf, NULL, 0, NULL, 0, nf, nf, nf, nf, NULL));
    REFUSED(h, drnmf_cell_backward_profile(h, &bad, nf, NULL, nf, 1.f, 0.f, 0.f, nf, nf, NULL, 0, NULL, 0, nf, nf,
                                           nf, nf, NULL, nf));
    drnmf_dense_desc_t dd;
    memset(&dd, 0, sizeof(dd));
    dd.B = 4; dd.T = 8; dd.F = 65; dd.N = 32; dd.K = 3; dd.connect_input = 1; dd.activation = DRNMF_ACT_RELU;
    REFUSED(h, drnmf_dense_prepare_params(h, &dd, nf, nf, nf, nf, NULL, NULL));
    REFUSED(h, drnmf_dense_cell_forward(h, &dd, nf, -1.f, NULL, nf, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_dense_cell_backward(h, &dd, nf, -1.f, nf, nf, nf, nf, nf, nf, nf, nf, nf, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_dense_cell_forward_dropout(h, &dd, nf, -1.f, NULL, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_dense_cell_backward_dropout(h, &dd, nf, -1.f, nf, nf, nf, nf, nf, nf, nf, nf, nf, nf, nf, nf,
                                                 nf, NULL, 0, NULL));
    dd.activation = 77;
    REFUSED(h, drnmf_dense_prepare_params(h, &dd, &f, &f, &f, &f, aligned, NULL));
    dd.activation = DRNMF_ACT_TANH; dd.K = 0;
    REFUSED(h, drnmf_dense_cell_forward(h, &dd, &f, -1.f, aligned, &f, nf, nf, &f, aligned, 1 << 20, NULL));
    REFUSED(h, drnmf_head_forward(h, 0, 65, 16, nf, 32, 0, nf, nf, 0, nf, nf, nf, nf, NULL));
    REFUSED(h, drnmf_head_forward(h, 10, 65, 16, nf, 32, 0, nf, nf, 0, nf, nf, nf, nf, NULL));
    REFUSED(h, drnmf_loss_head_backward(h, 10, 65, 16, nf, nf, 32, 0, nf, nf, 0, nf, nf, nf, nf, nf, nf, nf, nf, nf,
                                        NULL, 0, NULL));
    REFUSED(h, drnmf_snmf_cost_head_backward(h, 10, 65, 16, nf, nf, 32, 0, nf, nf, nf, nf, nf, 0.1f, nf, nf, nf, nf,
                                             NULL, 0, NULL));
    REFUSED(h, drnmf_adam_step(h, 0, nf, nf, nf, nf, 1e-3f, .9f, .999f, 1e-8f, 1.f, NULL));
    REFUSED(h, drnmf_adam_step(h, 10, nf, nf, nf, nf, 1e-3f, .9f, .999f, 1e-8f, 1.f, NULL));
    REFUSED(h, drnmf_sumsq(h, 10, nf, nf, NULL));
    drnmf_adam_block_t blk;
    memset(&blk, 0, sizeof(blk));
    REFUSED(h, drnmf_adam_step_flat(h, 0, &blk, &f, &f, &f, &f, NULL, 1e-3f, .9f, .999f, 1e-8f, 0.f, 0, 0.f, NULL, NULL));
    REFUSED(h, drnmf_adam_step_flat(h, 1, NULL, &f, &f, &f, &f, NULL, 1e-3f, .9f, .999f, 1e-8f, 0.f, 0, 0.f, NULL, NULL));
    REFUSED(h, drnmf_adam_step_flat(h, 1, &blk, &f, &f, &f, &f, NULL, 1e-3f, .9f, .999f, 1e-8f, 1.f, 0, 0.f, NULL, NULL));
    REFUSED(h, drnmf_adam_step_flat(h, 1, &blk, &f, &f, &f, &f, NULL, 1e-3f, .9f, .999f, 1e-8f, 0.f, 5, 0.f, NULL, NULL));
    /* the counted form: step_in / step_out must be two different device words; the optimiser constants are checked */
    REFUSED(h, drnmf_adam_step_flat_counted(h, 1, &blk, &f, &f, &f, &f, NULL, 1e-3, 0., .9, .999, 1e-8f, 0.f, 0, 0.f, &f, &f, NULL, NULL));
    REFUSED(h, drnmf_adam_step_flat_counted(h, 1, &blk, &f, &f, &f, &f, NULL, 1e-3, 0., .9, .999, 1e-8f, 0.f, 0, 0.f, NULL, &f, NULL, NULL));
    { float f2 = 0.f;
      REFUSED(h, drnmf_adam_step_flat_counted(h, 1, &blk, &f, &f, &f, &f, NULL, -1., 0., .9, .999, 1e-8f, 0.f, 0, 0.f, &f, &f2, NULL, NULL));
      REFUSED(h, drnmf_adam_step_flat_counted(h, 1, &blk, &f, &f, &f, &f, NULL, 1e-3, 0., 1.5, .999, 1e-8f, 0.f, 0, 0.f, &f, &f2, NULL, NULL));
      REFUSED(h, drnmf_adam_step_flat_counted(h, 1, &blk, &f, &f, &f, &f, NULL, 1e-3, 0., .9, .999, 1e-8f, 1.f, 0, 0.f, &f, &f2, NULL, NULL)); }
    /* matrix mode of the handle */
    CHECK(drnmf_get_matrix_mode(h) == DRNMF_MATRIX_F32 && drnmf_get_matrix_mode(NULL) < 0);
    REFUSED(h, drnmf_set_matrix_mode(h, 7));
    CHECK(drnmf_set_matrix_mode(h, DRNMF_MATRIX_BF16X3) == DRNMF_OK && drnmf_get_matrix_mode(h) == DRNMF_MATRIX_BF16X3);
    CHECK(drnmf_set_matrix_mode(h, DRNMF_MATRIX_F32) == DRNMF_OK && drnmf_set_matrix_mode(NULL, 0) < 0);
    REFUSED(h, drnmf_ista_forward(h, 10, 65, 32, 3, 0, 2.f, 1.f, 8.f, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_ista_forward(h, 10, 65, 32, 3, 7, 2.f, 1.f, 8.f, &f, &f, &f, aligned, 1 << 20, NULL));
    REFUSED(h, drnmf_mu_forward(h, 10, 65, 32, 5, 2.f, .1f, nf, nf, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_snmf_train_init(h, 10, 65, 32, 2.f, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_snmf_train_step(h, 10, 65, 32, 2.f, .1f, nf, nf, NULL, 1, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_stft_mag(h, 1, 16000, 500, 125, 0, NULL, nf, NULL));
    REFUSED(h, drnmf_stft_mag(h, 1, 16000, 512, 128, 0, NULL, nf, NULL));
    REFUSED(h, drnmf_stft(h, 1, 16000, 512, 128, 0, NULL, nf, nf, nf, NULL));
    REFUSED(h, drnmf_istft_masked(h, 1, 10, 16000, 512, 128, nf, nf, nf, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_snr(h, 1, 16000, nf, nf, nf, NULL));
    REFUSED(h, drnmf_sdr_corr(h, 1, 16000, 512, nf, nf, NULL, NULL, NULL, 0, NULL));
    REFUSED(h, drnmf_sdr_project(h, 1, 16000, 512, nf, nf, NULL, NULL, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_divide_a_by_aplusb(h, 10, nf, nf, nf, NULL));
    REFUSED(h, drnmf_add(h, 10, nf, nf, nf, NULL));
    REFUSED(h, drnmf_loss_forward(h, 10, 65, 0, nf, nf, nf, nf, nf, nf, 0, 0, 0.f, nf, NULL, 0, NULL));
    REFUSED(h, drnmf_loss_forward(h, 10, 65, 3, &f, &f, &f, &f, &f, &f, 0, 0, 0.f, &f, aligned, 1 << 20, NULL));
    REFUSED(h, drnmf_wav_int16(h, 10, nf, NULL, NULL, 0, NULL));
    /* collectives without a communicator */
    REFUSED(h, drnmf_allreduce_grads(h, &f, 1, NULL));
    REFUSED(h, drnmf_broadcast_params(h, &f, 1, 0, NULL));
    int32_t r = -5, w = -5;
    CHECK(drnmf_comm_info(h, &r, &w) != 0 || (r == 0 && w == 1));
    CHECK(drnmf_comm_destroy(h) == DRNMF_OK);
    /* fault word / report ring on a handle without host-mapped memory */
    CHECK(drnmf_check_status(h) == DRNMF_OK);
    CHECK(drnmf_persist_admitted(h) == 0 && drnmf_persist_admitted(NULL) < 0);
    CHECK(drnmf_persist_admit_reason(h)[0] != 0 && drnmf_persist_admit_reason(NULL)[0] != 0);
    CHECK(drnmf_status_take_device(h, &f, NULL) == DRNMF_OK);    /* no fault word: nothing is launched */
    REFUSED(h, drnmf_status_take_device(h, NULL, NULL));
    float* ring = NULL;
    int32_t slots = 0;
    REFUSED(h, drnmf_host_report_ring(h, &ring, &slots));
    REFUSED(h, drnmf_host_report_ring(h, NULL, NULL));
}

/* Threads (include/drnmf.h, "threads"): every handle-taking entry point serialises on the handle's mutex.
 * Two threads on two handles must be independent; two threads hammering ONE handle with calls that fail (each
 * writes the handle's error string) must neither corrupt memory nor deadlock.  Only return codes are looked
 * at here -- drnmf_last_error(h) of a handle another thread is failing calls on is that thread's text. */
typedef struct { drnmf_handle_t h; int iters; int bad; } thread_arg_t;
static void* thread_body(void* p) {
    thread_arg_t* a = (thread_arg_t*)p;
    for (int i = 0; i < a->iters; ++i) {
        drnmf_cell_desc_t d = cell_desc(4 + i % 5, 7, 21, 34, 3, 1, 0, 0, 0);
        d.K = (i & 1) ? 0 : 3;                                  /* every other descriptor is invalid */
        float dummy;
        const int32_t rc = drnmf_cell_forward(a->h, &d, &dummy, -1.f, NULL, &dummy, 1.f, 0.f, 0.f, &dummy,
                                              NULL, 0, NULL);
        if (rc >= 0) a->bad++;                                  /* (NULL params / workspace: always refused) */
        int32_t rank = -1, world = -1;
        if (drnmf_comm_info(a->h, &rank, &world) != DRNMF_OK || world != 1) a->bad++;
        if (drnmf_persist_admitted(a->h) != 0) a->bad++;
        if (drnmf_check_status(a->h) != DRNMF_OK) a->bad++;
    }
    return NULL;
}
static void threads(void) {
    drnmf_handle_t h1 = NULL, h2 = NULL;
    CHECK(drnmf_create_unbound(&h1) == DRNMF_OK && drnmf_create_unbound(&h2) == DRNMF_OK);
    pthread_t t[4];
    thread_arg_t a[4] = {{h1, 2000, 0}, {h2, 2000, 0}, {h1, 2000, 0}, {h1, 2000, 0}};
    /* two handles, one thread each */
    CHECK(pthread_create(&t[0], NULL, thread_body, &a[0]) == 0);
    CHECK(pthread_create(&t[1], NULL, thread_body, &a[1]) == 0);
    pthread_join(t[0], NULL);
    pthread_join(t[1], NULL);
    /* three threads on ONE handle */
    for (int i = 0; i < 3; ++i) a[i].h = h1;
    for (int i = 0; i < 3; ++i) CHECK(pthread_create(&t[i], NULL, thread_body, &a[i]) == 0);
    for (int i = 0; i < 3; ++i) pthread_join(t[i], NULL);
    for (int i = 0; i < 4; ++i) CHECK(a[i].bad == 0);
    CHECK(drnmf_destroy(h1) == DRNMF_OK && drnmf_destroy(h2) == DRNMF_OK);
}

int main(void) {
    CHECK(drnmf_version() > 0);
    drnmf_handle_t h = NULL;
    /* no GPU on this host (or a refused device index): create fails cleanly, with a message */
    CHECK(drnmf_create(NULL, 0) == DRNMF_ERR_INVALID_ARG);
    const int32_t rc = drnmf_create(&h, 4096);
    CHECK(rc < 0 && h == NULL && drnmf_last_error(NULL)[0] != 0);
    CHECK(drnmf_create_unbound(NULL) == DRNMF_ERR_INVALID_ARG);
    CHECK(drnmf_create_unbound(&h) == DRNMF_OK && h != NULL);
    CHECK(drnmf_destroy(NULL) == DRNMF_ERR_INVALID_ARG);
    /* layouts under every tuning variable that switches a branch of them */
    static const char* const env[][2] = {
        {NULL, NULL}, {"DRNMF_GRAM", "0"}, {"DRNMF_GRAM", "1"}, {"DRNMF_RB", "1"}, {"DRNMF_RB", "2"},
        {"DRNMF_KS", "1"}, {"DRNMF_KS", "8"}, {"DRNMF_RBA", "2"}, {"DRNMF_CP_FULL", "0"},
        {"DRNMF_SPLIT", "1"}, {"DRNMF_SPLIT", "3"}, {"DRNMF_SPLIT", "4"}, {"DRNMF_SPLIT", "8"}, {"DRNMF_PERSIST", "0"}};
    for (size_t i = 0; i < sizeof(env) / sizeof(*env); ++i) {
        if (env[i][0]) setenv(env[i][0], env[i][1], 1);
        CHECK(drnmf_reload_env() == DRNMF_OK);
        size_queries();
        refused_calls(h);
        if (env[i][0]) unsetenv(env[i][0]);
    }
    CHECK(drnmf_reload_env() == DRNMF_OK);
    CHECK(drnmf_destroy(h) == DRNMF_OK);
    threads();
    printf("sanitize_host: %d checks, %d failed\n", g_calls, g_fail);
    return g_fail ? 1 : 0;
}
