/* CPU oracle (C port) for the FNN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar float64 restatement of one pass of the reference's hot loop body
 * (python/FNN_wnzh.py:296-306 of Atomu2014/deep-ctr): gather (A3, :87-96) ->
 * train(x,y) (A4/A5, :144-182) -> sequential sparse-row SGD (A6, :299-306).
 * It mirrors oracle/fnn_oracle.py line for line and is checked against it in
 * tests/test_oracle.py; bench.py times it as the `cpu_baseline` ("port", 1 core).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference holds no golden vectors (see oracle/fnn_oracle.py).
 *
 * Build: gcc -O2 -fPIC -shared -o oracle/libfnn_oracle.so oracle/fnn_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int F, K, H1, H2;      /* fields, row width (rank+1), hidden sizes */
    double lr, lambda1, lambda_fm, w0;
} oracle_cfg;

/* x[t,0]=w0; x[t,1+f*K..] = rows[ids[t,f]]   (python/FNN_wnzh.py:91-96) */
void oracle_gather(const oracle_cfg* c, const double* rows, const int32_t* ids, int B, double* x)
{
    const int xdim = 1 + c->F * c->K;
    for (int t = 0; t < B; ++t) {
        double* xt = x + (size_t)t * xdim;
        memset(xt, 0, sizeof(double) * xdim);
        xt[0] = c->w0;
        for (int f = 0; f < c->F; ++f) {
            int32_t r = ids[t * c->F + f];
            if (r < 0) continue;
            memcpy(xt + 1 + f * c->K, rows + (size_t)r * c->K, sizeof(double) * c->K);
        }
    }
}

/* predict(x): python/FNN_wnzh.py:147-163,170-171,183 (tanh/tanh/sigmoid, no masks) */
void oracle_predict(const oracle_cfg* c, const double* rows, const int32_t* ids, int B,
                    const double* w1, const double* b1, const double* w2, const double* b2,
                    const double* w3, double b3, double* p_out)
{
    const int xdim = 1 + c->F * c->K, H1 = c->H1, H2 = c->H2;
    double* x = (double*)malloc(sizeof(double) * (size_t)B * xdim);
    double* h1 = (double*)malloc(sizeof(double) * H1);
    oracle_gather(c, rows, ids, B, x);
    for (int t = 0; t < B; ++t) {
        const double* xt = x + (size_t)t * xdim;
        for (int j = 0; j < H1; ++j) h1[j] = b1[j];
        for (int i = 0; i < xdim; ++i) {
            const double xi = xt[i];
            if (xi == 0.0) continue;
            const double* wr = w1 + (size_t)i * H1;
            for (int j = 0; j < H1; ++j) h1[j] += xi * wr[j];
        }
        for (int j = 0; j < H1; ++j) h1[j] = tanh(h1[j]);
        double z3 = b3;
        for (int j = 0; j < H2; ++j) {
            double z = b2[j];
            for (int i = 0; i < H1; ++i) z += h1[i] * w2[(size_t)i * H2 + j];
            z3 += tanh(z) * w3[j];
        }
        p_out[t] = 1.0 / (1.0 + exp(-z3));
    }
    free(x); free(h1);
}

/* One hot-loop pass.  Mutates rows and the six dense tensors in place.
 * r1/r2: dropout rows ({0,1} as double), broadcast over the batch, no rescale.
 * b_size: the batch length used in the decay constant (:304).  Returns sum(xent).
 * gx_out (nullable): [B, xdim].  p_out (nullable): p_drop [B]. */
double oracle_train_step(const oracle_cfg* c, double* rows, const int32_t* ids, const double* y,
                         int B, const double* r1, const double* r2, int b_size,
                         double* w1, double* b1, double* w2, double* b2, double* w3, double* b3,
                         double* gx_out, double* p_out)
{
    const int F = c->F, K = c->K, H1 = c->H1, H2 = c->H2, xdim = 1 + F * K;
    const double lr = c->lr;
    double* x  = (double*)malloc(sizeof(double) * (size_t)B * xdim);
    double* d1 = (double*)malloc(sizeof(double) * (size_t)B * H1);   /* tanh(z1)*r1 */
    double* h1 = (double*)malloc(sizeof(double) * (size_t)B * H1);
    double* t2 = (double*)malloc(sizeof(double) * (size_t)B * H2);
    double* dl1 = (double*)malloc(sizeof(double) * (size_t)B * H1);
    double* dl2 = (double*)malloc(sizeof(double) * (size_t)B * H2);
    double* d3 = (double*)malloc(sizeof(double) * B);
    double* gx = (double*)malloc(sizeof(double) * (size_t)B * xdim);
    double* gw1 = (double*)calloc((size_t)xdim * H1, sizeof(double));
    double* gw2 = (double*)calloc((size_t)H1 * H2, sizeof(double));
    double* gb1 = (double*)calloc(H1, sizeof(double));
    double* gb2 = (double*)calloc(H2, sizeof(double));
    double* gw3 = (double*)calloc(H2, sizeof(double));
    double gb3 = 0.0, loss = 0.0;

    oracle_gather(c, rows, ids, B, x);                               /* A3 */

    for (int t = 0; t < B; ++t) {                                    /* A4: forward */
        const double* xt = x + (size_t)t * xdim;
        double* h = h1 + (size_t)t * H1;
        for (int j = 0; j < H1; ++j) h[j] = b1[j];
        for (int i = 0; i < xdim; ++i) {
            const double xi = xt[i];
            if (xi == 0.0) continue;
            const double* wr = w1 + (size_t)i * H1;
            for (int j = 0; j < H1; ++j) h[j] += xi * wr[j];
        }
        double* d = d1 + (size_t)t * H1;
        for (int j = 0; j < H1; ++j) { h[j] = tanh(h[j]); d[j] = h[j] * r1[j]; }
        double* tt = t2 + (size_t)t * H2;
        for (int j = 0; j < H2; ++j) tt[j] = b2[j];
        for (int i = 0; i < H1; ++i) {
            const double di = d[i];
            if (di == 0.0) continue;
            const double* wr = w2 + (size_t)i * H2;
            for (int j = 0; j < H2; ++j) tt[j] += di * wr[j];
        }
        double z3 = *b3;
        for (int j = 0; j < H2; ++j) { tt[j] = tanh(tt[j]); z3 += tt[j] * r2[j] * w3[j]; }
        const double p = 1.0 / (1.0 + exp(-z3));
        if (p_out) p_out[t] = p;
        loss += -y[t] * log(p) - (1.0 - y[t]) * log(1.0 - p);       /* :172 */
        d3[t] = p - y[t];
    }

    for (int t = 0; t < B; ++t) {                                    /* A5: backward */
        const double* tt = t2 + (size_t)t * H2;
        double* l2 = dl2 + (size_t)t * H2;
        gb3 += d3[t];
        for (int j = 0; j < H2; ++j) {
            gw3[j] += tt[j] * r2[j] * d3[t];
            l2[j] = d3[t] * w3[j] * r2[j] * (1.0 - tt[j] * tt[j]);
            gb2[j] += l2[j];
        }
        const double* d = d1 + (size_t)t * H1;
        const double* h = h1 + (size_t)t * H1;
        double* l1 = dl1 + (size_t)t * H1;
        for (int i = 0; i < H1; ++i) {
            const double* wr = w2 + (size_t)i * H2;
            double* gr = gw2 + (size_t)i * H2;
            double s = 0.0;
            for (int j = 0; j < H2; ++j) { s += l2[j] * wr[j]; gr[j] += d[i] * l2[j]; }
            l1[i] = s * r1[i] * (1.0 - h[i] * h[i]);
            gb1[i] += l1[i];
        }
        const double* xt = x + (size_t)t * xdim;
        double* gxt = gx + (size_t)t * xdim;
        for (int i = 0; i < xdim; ++i) {
            const double* wr = w1 + (size_t)i * H1;
            double* gr = gw1 + (size_t)i * H1;
            const double xi = xt[i];
            double s = 0.0;
            for (int j = 0; j < H1; ++j) { s += l1[j] * wr[j]; gr[j] += xi * l1[j]; }
            gxt[i] = s;
        }
    }
    for (int j = 0; j < H2; ++j) gw3[j] += 2.0 * c->lambda1 * w3[j];   /* :173 */
    gb3 += 2.0 * c->lambda1 * (*b3);

    for (size_t i = 0; i < (size_t)xdim * H1; ++i) w1[i] -= lr * gw1[i];   /* :179-182 */
    for (int j = 0; j < H1; ++j) b1[j] -= lr * gb1[j];
    for (size_t i = 0; i < (size_t)H1 * H2; ++i) w2[i] -= lr * gw2[i];
    for (int j = 0; j < H2; ++j) { b2[j] -= lr * gb2[j]; w3[j] -= lr * gw3[j]; }
    *b3 -= lr * gb3;

    const double cdec = 1.0 - 2.0 * c->lambda_fm * lr / (double)b_size;   /* A6, :299-306 */
    for (int t = 0; t < B; ++t) {
        const double* gxt = gx + (size_t)t * xdim;
        for (int f = 0; f < F; ++f) {
            int32_t r = ids[t * F + f];
            if (r < 0) continue;
            double* row = rows + (size_t)r * K;
            for (int l = 0; l < K; ++l) row[l] = row[l] * cdec - lr * gxt[1 + f * K + l] * 1;
        }
    }
    if (gx_out) memcpy(gx_out, gx, sizeof(double) * (size_t)B * xdim);
    free(x); free(d1); free(h1); free(t2); free(dl1); free(dl2); free(d3); free(gx);
    free(gw1); free(gw2); free(gb1); free(gb2); free(gw3);
    return loss;
}
