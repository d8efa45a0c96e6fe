/* A plain C99 client of include/demcz.h: what a binding written in C (or any FFI that speaks the C ABI: Julia's ccall,
 * cgo, JNI ...) does.  No Python, no C++: create -> set_state -> run -> get_history -> get_state -> destroy on the
 * MvNormal target at d = 2, N = 64, K = 10, G = 40, and a few checks that need no oracle (the oracle comparison is the
 * Python tests' job): the history's last generation is the current state, M grew by N per boundary, the last appended rows
 * are the states at the last boundary, a second identical run gives identical bits, and the status codes / error strings
 * work.  Exit code 0 = all good, 77 = no GPU (demcz_create says DEMCZ_ERR_NO_DEVICE), anything else = failure.
 * Built and run by tests/test_c_abi.py (gcc -std=c99 -pedantic -Wall -Werror). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "demcz.h"

#define NCH 64
#define DIM 2
#define KW 10
#define NG 40
#define M0 64

#define CHECK(cond, msg)                                              \
    do {                                                              \
        if (!(cond)) { fprintf(stderr, "FAILED: %s\n", msg); return 1; } \
    } while (0)

static int run_once(double* chain, double* logobj, double* X, double* lp, double* Z, int64_t* M)
{
    static double Zinit[M0 * DIM], X0[NCH * DIM];
    const double mu[DIM] = {0.25, -0.5};
    const double W[DIM * DIM] = {2.0, 0.5, 0.0, 1.5};      /* column-major lower triangular inv(chol(Sigma)) */
    const double eps[DIM] = {1e-5, 1e-5};
    const int32_t offs[2] = {0, DIM};
    const int32_t idx[DIM] = {0, 1};
    demcz_config cfg;
    demcz_handle* h = NULL;
    int32_t rc;
    int i, p;
    unsigned long long s = 88172645463325252ull;       /* xorshift: deterministic start archive */
    for (i = 0; i < M0 * DIM; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        Zinit[i] = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
    }
    for (p = 0; p < DIM; ++p)
        for (i = 0; i < NCH; ++i) X0[i + NCH * p] = Zinit[(M0 - NCH + i) + M0 * p];      /* the last NCH rows, demcz.jl:15 */
    memset(&cfg, 0, sizeof cfg);
    cfg.N = NCH; cfg.d = DIM; cfg.K = KW; cfg.Mcap = M0 + NCH * (NG / KW); cfg.Gcap = NG; cfg.Nblocks = 1;
    cfg.block_offsets = offs; cfg.block_indices = idx; cfg.eps_scale = eps; cfg.seed = 2024; cfg.device_id = 0;
    cfg.target_kind = DEMCZ_TARGET_MVNORMAL; cfg.mu = mu; cfg.W = W; cfg.c0 = -1.0;
    rc = demcz_create(&h, &cfg);
    if (rc == DEMCZ_ERR_NO_DEVICE) { fprintf(stderr, "no device: %s\n", demcz_last_error(NULL)); return 77; }
    CHECK(rc == DEMCZ_OK, demcz_last_error(NULL));
    CHECK(demcz_run(h, 1, 5, 2.38, NULL) == DEMCZ_ERR_STATE, "run before set_state must be DEMCZ_ERR_STATE");
    CHECK(strlen(demcz_last_error(h)) > 0, "an error leaves a message");
    CHECK(demcz_set_state(h, X0, NULL, Zinit, M0, M0) == DEMCZ_OK, demcz_last_error(h));
    CHECK(demcz_run(h, 1, 17, 2.38, NULL) == DEMCZ_OK, demcz_last_error(h));
    CHECK(demcz_run(h, 18, NG, 2.38, NULL) == DEMCZ_OK, demcz_last_error(h));
    CHECK(demcz_run(h, NG + 1, NG + 1, 2.38, NULL) == DEMCZ_ERR_CAPACITY, "a generation outside the history window must be DEMCZ_ERR_CAPACITY");
    CHECK(demcz_get_history(h, 1, NG, chain, logobj) == DEMCZ_OK, demcz_last_error(h));
    CHECK(demcz_get_state(h, X, lp, Z, cfg.Mcap, M) == DEMCZ_OK, demcz_last_error(h));
    CHECK(demcz_destroy(h) == DEMCZ_OK, "destroy");
    return 0;
}

int main(void)
{
    static double chain[2][NCH * DIM * NG], logobj[2][NCH * NG], X[2][NCH * DIM], lp[2][NCH], Z[2][(M0 + NCH * (NG / KW)) * DIM];
    int64_t M[2] = {0, 0};
    const int64_t Mcap = M0 + NCH * (NG / KW);
    int r, i, p, changed = 0;
    if (demcz_abi_version() != DEMCZ_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    for (r = 0; r < 2; ++r) {
        const int rc = run_once(chain[r], logobj[r], X[r], lp[r], Z[r], &M[r]);
        if (rc) return rc;
    }
    CHECK(M[0] == Mcap, "M grows by NCH at every KW-th generation (demcz.jl:88-91)");
    for (p = 0; p < DIM; ++p)
        for (i = 0; i < NCH; ++i) {
            CHECK(chain[0][i + NCH * (p + DIM * (NG - 1))] == X[0][i + NCH * p], "chain[:, :, end] is Xcurrent (demcz.jl:84-86)");
            CHECK(Z[0][(Mcap - NCH + i) + Mcap * p] == chain[0][i + NCH * (p + DIM * (NG - 1))], "the last appended rows are the states at generation NG (NG % KW == 0)");
        }
    for (i = 0; i < NCH; ++i) {
        CHECK(logobj[0][i + NCH * (NG - 1)] == lp[0][i], "log_obj[:, end] is log_objcurrent (demcz.jl:85-87)");
        if (logobj[0][i + NCH * (NG - 1)] != logobj[0][i]) ++changed;
    }
    CHECK(changed > NCH / 2, "chains moved");
    CHECK(memcmp(chain[0], chain[1], sizeof chain[0]) == 0 && memcmp(Z[0], Z[1], sizeof Z[0]) == 0 && memcmp(logobj[0], logobj[1], sizeof logobj[0]) == 0,
          "the same seed gives the same bits");
    printf("abi_smoke OK: %d chains x %d generations through the C ABI, M = %lld\n", NCH, NG, (long long)M[0]);
    return 0;
}
