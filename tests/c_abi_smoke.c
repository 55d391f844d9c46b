/* Plain-C caller of the C ABI (include/lsnf_flow.h): no Python, no torch -- only the HIP runtime for device memory.
 *
 * Builds a flow whose answer is known in closed form: every actnorm b/logs = 0, W = a permutation matrix (reverse the
 * features; |det| = 1), fc_1 / fc_2 weights = 0, fc_zeros w = 0 with bias b3 (shift columns: 0.5, pre-sigmoid columns:
 * 1.0) and logs3 = 0.  Then per block:  v = reverse(z);  y = [v1, (v2 + 0.5) * sigmoid(1 + 2)];  logdet += (nz/2) * log(sigmoid(3)).
 * The program runs prepare + forward + reverse through the library and checks the numbers against that recurrence
 * evaluated on the host in double precision.  Exit code 0 = pass.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "lsnf_flow.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define LS(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "lsnf error %d: %s (%s:%d)\n", rc_, lsnf_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static float* dev_upload(const float* h, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void) {
    enum { NZ = 24, W = 10, DEPTH = 3, B = 77, HALF = NZ / 2 };
    char arch[64];
    LS(lsnf_device_arch(0, arch, sizeof arch));
    if (lsnf_abi_version() != LSNF_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    /* ---- parameters (12 tensors per block, order of lsnf_flow.h) ---- */
    const size_t sizes[12] = {NZ, NZ, NZ * NZ, HALF * W, W, W, W * W, W, W, W * NZ, NZ, NZ};
    const float* params[DEPTH * 12];
    for (int blk = 0; blk < DEPTH; ++blk)
        for (int i = 0; i < 12; ++i) {
            float* h = (float*)calloc(sizes[i], sizeof(float));
            if (i == 2) for (int k = 0; k < NZ; ++k) h[k * NZ + (NZ - 1 - k)] = 1.0f;          /* W = feature reversal */
            if (i == 10) for (int c = 0; c < NZ; ++c) h[c] = (c & 1) ? 1.0f : 0.5f;             /* b3: shift 0.5 / pre-sigmoid 1 */
            params[blk * 12 + i] = dev_upload(h, sizes[i]);
            free(h);
            if (!params[blk * 12 + i]) { fprintf(stderr, "upload failed\n"); return 2; }
        }
    size_t nplan = lsnf_plan_floats(NZ, W, DEPTH, 1), nscratch = lsnf_prepare_scratch_bytes(NZ, W, DEPTH);
    if (!nplan || !nscratch) { fprintf(stderr, "geometry rejected\n"); return 1; }
    float* plan; void* scratch;
    CHECK(hipMalloc((void**)&plan, nplan * sizeof(float)));
    CHECK(hipMalloc(&scratch, nscratch));
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    LS(lsnf_prepare(params, NZ, W, DEPTH, 1, plan, scratch, stream));

    /* ---- input and host-side expectation ---- */
    float* hz = (float*)malloc(sizeof(float) * B * NZ);
    for (int i = 0; i < B * NZ; ++i) hz[i] = (float)((i * 37 % 101) - 50) / 25.0f;
    const double sg = 1.0 / (1.0 + exp(-3.0)), lsg = log(sg);
    double* ez = (double*)malloc(sizeof(double) * B * NZ);
    double elogdet = 0.0;
    for (int i = 0; i < B * NZ; ++i) ez[i] = hz[i];
    for (int blk = 0; blk < DEPTH; ++blk) {
        for (int r = 0; r < B; ++r) {
            double v[NZ];
            for (int j = 0; j < NZ; ++j) v[j] = ez[r * NZ + (NZ - 1 - j)];
            for (int j = 0; j < HALF; ++j) ez[r * NZ + j] = v[j];
            for (int j = HALF; j < NZ; ++j) ez[r * NZ + j] = (v[j] + 0.5) * sg;
        }
        elogdet += HALF * lsg;
    }
    float *dz = dev_upload(hz, (size_t)B * NZ), *dz1, *dld, *dll, *dback, *dobj;
    double* dstats;
    CHECK(hipMalloc((void**)&dz1, sizeof(float) * B * NZ)); CHECK(hipMalloc((void**)&dback, sizeof(float) * B * NZ));
    CHECK(hipMalloc((void**)&dld, sizeof(float) * B)); CHECK(hipMalloc((void**)&dll, sizeof(float) * B)); CHECK(hipMalloc((void**)&dobj, sizeof(float) * B));
    CHECK(hipMalloc((void**)&dstats, LSNF_STATS_DOUBLES * sizeof(double))); CHECK(hipMemset(dstats, 0, LSNF_STATS_DOUBLES * sizeof(double)));

    int fails = 0;
    for (int family = 0; family < 2; ++family) {                 /* latency kernels, then throughput kernels */
        lsnf_set_small_batch_max(family == 0 ? (1 << 30) : 0);
        LS(lsnf_forward(plan, NZ, W, DEPTH, 1, 0, DEPTH, B, dz, NULL, dz1, dld, dll, NULL, NULL, NULL, dstats, stream));
        LS(lsnf_reverse(plan, NZ, W, DEPTH, 1, B, dz1, dld, dback, dobj, stream));
        CHECK(hipStreamSynchronize(stream));
        float* o = (float*)malloc(sizeof(float) * B * NZ); float ld[B], ll[B], ob[B]; double st[8];
        CHECK(hipMemcpy(o, dz1, sizeof(float) * B * NZ, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(ld, dld, sizeof ld, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ll, dll, sizeof ll, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(st, dstats, sizeof st, hipMemcpyDeviceToHost));
        double emax = 0, lmax = 0, llmax = 0, sll = 0;
        for (int r = 0; r < B; ++r) {
            double ss = 0;
            for (int j = 0; j < NZ; ++j) { double d = fabs(o[r * NZ + j] - ez[r * NZ + j]); if (d > emax) emax = d; ss += ez[r * NZ + j] * ez[r * NZ + j]; }
            double ell = -0.5 * ss + log(2 * 3.14159265358979323846) + elogdet;
            if (fabs(ld[r] - elogdet) > lmax) lmax = fabs(ld[r] - elogdet);
            if (fabs(ll[r] - ell) / fabs(ell) > llmax) llmax = fabs(ll[r] - ell) / fabs(ell);
            sll += ll[r];
        }
        CHECK(hipMemcpy(o, dback, sizeof(float) * B * NZ, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob, dobj, sizeof ob, hipMemcpyDeviceToHost));
        double rmax = 0, omax = 0;
        for (int i = 0; i < B * NZ; ++i) if (fabs(o[i] - hz[i]) > rmax) rmax = fabs(o[i] - hz[i]);
        for (int r = 0; r < B; ++r) if (fabs(ob[r]) > omax) omax = fabs(ob[r]);
        const int ok = emax < 1e-5 && lmax < 1e-5 && llmax < 1e-5 && rmax < 1e-4 && omax < 1e-4 && fabs(st[4] - sll) < 1e-6 * fabs(sll) && st[6] == B;
        printf("%s on %s: |z1 err| %.2e  |logdet err| %.2e  ll rel %.2e  round-trip %.2e / %.2e  sum ll %.6f vs %.6f -> %s\n",
               family == 0 ? "latency kernels   " : "throughput kernels", arch, emax, lmax, llmax, rmax, omax, st[4], sll, ok ? "ok" : "FAIL");
        fails += !ok;
        free(o);
    }
    /* error path: geometry outside the supported range must be refused with a message, not crash */
    if (lsnf_forward(plan, 130, W, DEPTH, 1, 0, DEPTH, B, dz, NULL, dz1, dld, dll, NULL, NULL, NULL, NULL, stream) != LSNF_E_GEOMETRY) { fprintf(stderr, "bad geometry accepted\n"); fails++; }
    return fails ? 1 : 0;
}
